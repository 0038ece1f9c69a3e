// Micro-benchmark: issue cost of the VALU / LDS instructions the EDT line passes are made of, on
// gfx950, as a function of waves per SIMD.  Prints cycles per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kIters = 2000;
constexpr int kUnroll = 16;  // instructions per loop body, 8 independent chains x 2

#define BODY8(INS) \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c) : "vcc");

#define DEFK(NAME, INS) \
  __global__ void NAME(uint32_t* out, uint32_t seed, long long* cyc) { \
    uint32_t a[8]; for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * (i + 1); \
    uint32_t b = seed * 3 + threadIdx.x, c = seed + 7; \
    long long t0 = clock64(); \
    for (int it = 0; it < kIters; it++) { BODY8(INS) } \
    long long t1 = clock64(); \
    uint32_t s = 0; for (int i = 0; i < 8; i++) s += a[i]; \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s; \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0; }

#define I_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define I_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define I_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define I_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define I_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define I_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define I_MIN(i) "v_min_i32 %" #i ", %" #i ", %8\n"
#define I_MIN3(i) "v_min3_i32 %" #i ", %" #i ", %8, %9\n"
#define I_FFBH(i) "v_ffbh_u32 %" #i ", %" #i "\n"
#define I_FFBL(i) "v_ffbl_b32 %" #i ", %" #i "\n"
#define I_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define I_CMPCND(i) "v_cmp_lt_u32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define I_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define I_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define I_RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define I_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define I_CVTF(i) "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define I_CVTI(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define I_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define I_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %9\n"
#define I_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 5\n"
#define I_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define I_PKADD16(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define I_PKMIN16(i) "v_pk_min_i16 %" #i ", %" #i ", %8\n"
#define I_PKMAD16(i) "v_pk_mad_u16 %" #i ", %" #i ", %8, %9\n"
#define I_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define I_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_BPERM(i) "ds_bpermute_b32 %" #i ", %8, %" #i "\ns_waitcnt lgkmcnt(0)\n"
#define I_BPERM_NOWAIT(i) "ds_bpermute_b32 %" #i ", %8, %" #i "\n"
#define I_SADD(i) "s_add_u32 s20, s20, 1\n"

DEFK(k_add, I_ADD) DEFK(k_and, I_AND) DEFK(k_lshl, I_LSHL) DEFK(k_mad24, I_MAD24) DEFK(k_mul24, I_MUL24)
DEFK(k_mullo, I_MULLO) DEFK(k_mulhi, I_MULHI) DEFK(k_min, I_MIN) DEFK(k_min3, I_MIN3) DEFK(k_ffbh, I_FFBH)
DEFK(k_ffbl, I_FFBL) DEFK(k_bcnt, I_BCNT) DEFK(k_cmpcnd, I_CMPCND) DEFK(k_cnd, I_CND) DEFK(k_fma32, I_FMA32)
DEFK(k_rcp, I_RCP) DEFK(k_rsq, I_RSQ) DEFK(k_sqrt, I_SQRT) DEFK(k_cvtf, I_CVTF) DEFK(k_cvti, I_CVTI)
DEFK(k_add3, I_ADD3) DEFK(k_lshladd, I_LSHLADD) DEFK(k_bfe, I_BFE) DEFK(k_perm, I_PERM)
DEFK(k_pkadd16, I_PKADD16) DEFK(k_pkmin16, I_PKMIN16) DEFK(k_pkmad16, I_PKMAD16) DEFK(k_mov, I_MOV) DEFK(k_dpp, I_DPP)
DEFK(k_bperm, I_BPERM_NOWAIT)

// 64-bit / double forms: 4 chains of register pairs
#define BODY4D(INS) \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(0) INS(1) INS(2) INS(3) INS(0) INS(1) INS(2) INS(3) INS(0) INS(1) INS(2) INS(3) \
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b), "v"(c), "v"(bi) : "vcc");
#define DEFKD(NAME, INS) \
  __global__ void NAME(uint32_t* out, uint32_t seed, long long* cyc) { \
    double a[4]; for (int i = 0; i < 4; i++) a[i] = 1.0 + seed + threadIdx.x * (i + 1); \
    double b = 1.0000001, c = 1e-9 * seed; uint32_t bi = seed + threadIdx.x; \
    long long t0 = clock64(); \
    for (int it = 0; it < kIters; it++) { BODY4D(INS) } \
    long long t1 = clock64(); \
    double s = 0; for (int i = 0; i < 4; i++) s += a[i]; \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)__double_as_longlong(s); \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0; }
#define D_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %4, %5\n"
#define D_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %4\n"
#define D_ADD64(i) "v_add_f64 %" #i ", %" #i ", %5\n"
#define D_MAD64(i) "v_mad_u64_u32 %" #i ", vcc, %6, %6, %" #i "\n"
#define D_LSHL64(i) "v_lshlrev_b64 %" #i ", 1, %" #i "\n"
#define D_RSQ64(i) "v_rsq_f64 %" #i ", %" #i "\n"
#define D_SQRT64(i) "v_sqrt_f64 %" #i ", %" #i "\n"
#define D_RCP64(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define D_CVT64(i) "v_cvt_f64_i32 %" #i ", %6\n"
#define D_CVT32(i) "v_cvt_f32_f64 %6, %" #i "\n"
#define D_PKFMA32(i) "v_pk_fma_f32 %" #i ", %" #i ", %4, %5\n"
#define D_CMP64(i) "v_cmp_lt_f64 vcc, %" #i ", %4\n"
#define D_CMPU64(i) "v_cmp_lt_u64 vcc, %" #i ", %4\n"
DEFKD(k_fma64, D_FMA64) DEFKD(k_mul64, D_MUL64) DEFKD(k_add64, D_ADD64) DEFKD(k_mad64, D_MAD64) DEFKD(k_lshl64, D_LSHL64)
DEFKD(k_rsq64, D_RSQ64) DEFKD(k_sqrt64, D_SQRT64) DEFKD(k_rcp64, D_RCP64) DEFKD(k_cvt64, D_CVT64)
DEFKD(k_pkfma32, D_PKFMA32) DEFKD(k_cmp64, D_CMP64) DEFKD(k_cmpu64, D_CMPU64)

// LDS: conflict-free ds_read_b32 / ds_write_b32 streams
__global__ void k_ldsread(uint32_t* out, uint32_t seed, long long* cyc) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i + seed;
  __syncthreads();
  uint32_t s = 0; const uint32_t* p = lds + (threadIdx.x & 63);
  long long t0 = clock64();
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int k = 0; k < 16; k++) s += p[k * 64];
    asm volatile("" : "+v"(s));
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_ldsread_stride8(uint32_t* out, uint32_t seed, long long* cyc) {
  // F[row][8] layout: lane = (band, line) reads row r of its line; 8 lanes with the same band share a row
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i + seed;
  __syncthreads();
  uint32_t s = 0; const int lane = threadIdx.x & 63; const uint32_t* p = lds + (lane >> 3) * 32 * 8 + (lane & 7);
  long long t0 = clock64();
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int k = 0; k < 16; k++) s += p[k * 8];
    asm volatile("" : "+v"(s));
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*Kern)(uint32_t*, uint32_t, long long*);
struct Entry { const char* name; Kern k; int per_iter; size_t lds; };

int main() {
  uint32_t* out; long long* cyc;
  const int cus = 256;
  CHECK(hipMalloc(&out, sizeof(uint32_t) * cus * 8 * 1024));
  CHECK(hipMalloc(&cyc, sizeof(long long) * cus * 8));
  std::vector<Entry> es = {
    {"v_add_u32", k_add, 16, 0}, {"v_and_b32", k_and, 16, 0}, {"v_lshlrev_b32", k_lshl, 16, 0},
    {"v_mad_u32_u24", k_mad24, 16, 0}, {"v_mul_u32_u24", k_mul24, 16, 0}, {"v_mul_lo_u32", k_mullo, 16, 0},
    {"v_mul_hi_u32", k_mulhi, 16, 0}, {"v_min_i32", k_min, 16, 0}, {"v_min3_i32", k_min3, 16, 0},
    {"v_ffbh_u32", k_ffbh, 16, 0}, {"v_ffbl_b32", k_ffbl, 16, 0}, {"v_bcnt_u32_b32", k_bcnt, 16, 0},
    {"v_cmp+v_cndmask (pair)", k_cmpcnd, 16, 0}, {"v_cndmask_b32", k_cnd, 16, 0}, {"v_fma_f32", k_fma32, 16, 0},
    {"v_rcp_f32", k_rcp, 16, 0}, {"v_rsq_f32", k_rsq, 16, 0}, {"v_sqrt_f32", k_sqrt, 16, 0},
    {"v_cvt_f32_i32", k_cvtf, 16, 0}, {"v_cvt_i32_f32", k_cvti, 16, 0}, {"v_add3_u32", k_add3, 16, 0},
    {"v_lshl_add_u32", k_lshladd, 16, 0}, {"v_bfe_u32", k_bfe, 16, 0}, {"v_perm_b32", k_perm, 16, 0},
    {"v_pk_add_u16", k_pkadd16, 16, 0}, {"v_pk_min_i16", k_pkmin16, 16, 0}, {"v_pk_mad_u16", k_pkmad16, 16, 0},
    {"v_mov_b32", k_mov, 16, 0}, {"v_mov_b32_dpp row_shr", k_dpp, 16, 0}, {"ds_bpermute_b32", k_bperm, 16, 0},
    {"v_fma_f64", k_fma64, 16, 0}, {"v_mul_f64", k_mul64, 16, 0}, {"v_add_f64", k_add64, 16, 0},
    {"v_mad_u64_u32", k_mad64, 16, 0}, {"v_lshlrev_b64", k_lshl64, 16, 0}, {"v_rsq_f64", k_rsq64, 16, 0},
    {"v_sqrt_f64", k_sqrt64, 16, 0}, {"v_rcp_f64", k_rcp64, 16, 0}, {"v_cvt_f64_i32", k_cvt64, 16, 0},
    {"v_pk_fma_f32", k_pkfma32, 16, 0}, {"v_cmp_lt_f64", k_cmp64, 16, 0}, {"v_cmp_lt_u64", k_cmpu64, 16, 0},
    {"ds_read_b32 conflict-free", k_ldsread, 16, 16384}, {"ds_read_b32 F[row][8] pattern", k_ldsread_stride8, 16, 32768},
  };
  printf("%-32s", "cycles per wave-instr per SIMD");
  const int wps[] = {1, 2, 4, 8};
  for (int w : wps) printf("  %dw/SIMD(tick/ns)", w);
  printf("\n");
  std::vector<long long> h(cus * 8);
  hipEvent_t ev0, ev1; CHECK(hipEventCreate(&ev0)); CHECK(hipEventCreate(&ev1));
  for (auto& e : es) {
    printf("%-32s", e.name);
    for (int w : wps) {
      // one block per CU with 4*w waves -> w waves per SIMD
      const int threads = 256 * w;
      const int blocks = (threads > 1024) ? cus * 2 : cus;
      const int tpb = (threads > 1024) ? 1024 : threads;
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(tpb), e.lds, 0, out, 1u, cyc);
      CHECK(hipEventRecord(ev0, 0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(tpb), e.lds, 0, out, 2u, cyc);
      CHECK(hipEventRecord(ev1, 0));
      CHECK(hipDeviceSynchronize());
      float ms = 0; CHECK(hipEventElapsedTime(&ms, ev0, ev1));
      CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
      double sum = 0; for (int i = 0; i < blocks; i++) sum += h[i];
      const double cycles = sum / blocks;  // cycles for kIters * per_iter instructions per wave, w waves per SIMD
      printf("  %6.2f/%5.2fns", cycles / (double(kIters) * e.per_iter * w), ms * 1e6 / (double(kIters) * e.per_iter * w));
    }
    printf("\n");
  }
  return 0;
}
