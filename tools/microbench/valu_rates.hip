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

#define J_SUB(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
DEFK(k2_sub, J_SUB)
#define J_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n"
DEFK(k2_or, J_OR)
#define J_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
DEFK(k2_xor, J_XOR)
#define J_MINF(i) "v_min_f32 %" #i ", %" #i ", %8\n"
DEFK(k2_minf, J_MINF)
#define J_MAXF(i) "v_max_f32 %" #i ", %" #i ", %8\n"
DEFK(k2_maxf, J_MAXF)
#define J_ADDF(i) "v_add_f32 %" #i ", %" #i ", %8\n"
DEFK(k2_addf, J_ADDF)
#define J_MULF(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
DEFK(k2_mulf, J_MULF)
#define J_SUBF(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
DEFK(k2_subf, J_SUBF)
#define J_MINU(i) "v_min_u32 %" #i ", %" #i ", %8\n"
DEFK(k2_minu, J_MINU)
#define J_MAXI(i) "v_max_i32 %" #i ", %" #i ", %8\n"
DEFK(k2_maxi, J_MAXI)
#define J_LSHR(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
DEFK(k2_lshr, J_LSHR)
#define J_ASHR(i) "v_ashrrev_i32 %" #i ", 1, %" #i "\n"
DEFK(k2_ashr, J_ASHR)
#define J_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
DEFK(k2_andor, J_ANDOR)
#define J_BFI(i) "v_bfi_b32 %" #i ", %" #i ", %8, %9\n"
DEFK(k2_bfi, J_BFI)
#define J_NOT(i) "v_not_b32 %" #i ", %" #i "\n"
DEFK(k2_not, J_NOT)
#define J_MADI24(i) "v_mad_i32_i24 %" #i ", %" #i ", %8, %9\n"
DEFK(k2_madi24, J_MADI24)
#define J_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
DEFK(k2_fmac, J_FMAC)
#define J_CVTFU(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
DEFK(k2_cvtfu, J_CVTFU)
#define J_MIN3F(i) "v_min3_f32 %" #i ", %" #i ", %8, %9\n"
DEFK(k2_min3f, J_MIN3F)
#define J_MED3F(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
DEFK(k2_med3f, J_MED3F)
#define J_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 3\n"
DEFK(k2_alignbit, J_ALIGNBIT)
#define J_CMPF(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
DEFK(k2_cmpf, J_CMPF)
#define J_CMPU(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n"
DEFK(k2_cmpu, J_CMPU)
#define J_CMPI(i) "v_cmp_lt_i32 vcc, %" #i ", %8\n"
DEFK(k2_cmpi, J_CMPI)
#define J_CMPX(i) "v_cmp_lt_u32 s[20:21], %" #i ", %8\n"
DEFK(k2_cmpx, J_CMPX)
#define J_CNDSG(i) "v_cndmask_b32 %" #i ", %" #i ", %9, s[20:21]\n"
DEFK(k2_cndsg, J_CNDSG)
#define J_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 2, %9\n"
DEFK(k2_lshlor, J_LSHLOR)
#define J_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
DEFK(k2_or3, J_OR3)
#define J_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
DEFK(k2_xad, J_XAD)
#define J_ADDLSHL(i) "v_add_lshl_u32 %" #i ", %" #i ", %8, 1\n"
DEFK(k2_addlshl, J_ADDLSHL)
#define J_SUBREV(i) "v_subrev_u32 %" #i ", %8, %" #i "\n"
DEFK(k2_subrev, J_SUBREV)
#define J_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x80\n"
DEFK(k2_bitop3, J_BITOP3)
#define J_MAXF64(i) "v_max_f32 %" #i ", %" #i ", %" #i "\n"
DEFK(k2_maxf64, J_MAXF64)
#define J_RNDNE(i) "v_rndne_f32 %" #i ", %" #i "\n"
DEFK(k2_rndne, J_RNDNE)
#define J_FLOOR(i) "v_floor_f32 %" #i ", %" #i "\n"
DEFK(k2_floor, J_FLOOR)
#define J_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n"
DEFK(k2_mbcnt, J_MBCNT)
#define J_READLANE(i) "v_readfirstlane_b32 s20, %" #i "\n"
DEFK(k2_readlane, J_READLANE)
#define J_MUL_LEGACY(i) "v_mul_legacy_f32 %" #i ", %" #i ", %8\n"
DEFK(k2_mul_legacy, J_MUL_LEGACY)
#define J_LDEXP(i) "v_ldexp_f32 %" #i ", %" #i ", %8\n"
DEFK(k2_ldexp, J_LDEXP)
#define J_SAD(i) "v_sad_u32 %" #i ", %" #i ", %8, %9\n"
DEFK(k2_sad, J_SAD)
#define J_CVTPK(i) "v_cvt_pk_u16_u32 %" #i ", %" #i ", %8\n"
DEFK(k2_cvtpk, J_CVTPK)
#define J_PKMUL(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n"
DEFK(k2_pkmul, J_PKMUL)
#define J_PKMAX(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n"
DEFK(k2_pkmax, J_PKMAX)
#define J_PKLSHL(i) "v_pk_lshlrev_b16 %" #i ", 1, %" #i "\n"
DEFK(k2_pklshl, J_PKLSHL)
#define J_PKSUB(i) "v_pk_sub_i16 %" #i ", %" #i ", %8\n"
DEFK(k2_pksub, J_PKSUB)
#define J_DOT2(i) "v_dot2_u32_u16 %" #i ", %" #i ", %8, %9\n"
DEFK(k2_dot2, J_DOT2)
#define J_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
DEFK(k2_dot4, J_DOT4)
#define J_MADU16(i) "v_mad_u32_u16 %" #i ", %" #i ", %8, %9\n"
DEFK(k2_madu16, J_MADU16)
#define J_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
DEFK(k2_addco, J_ADDCO)

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
    {"v_sub_u32", k2_sub, 16, 0},
    {"v_or_b32", k2_or, 16, 0},
    {"v_xor_b32", k2_xor, 16, 0},
    {"v_min_f32", k2_minf, 16, 0},
    {"v_max_f32", k2_maxf, 16, 0},
    {"v_add_f32", k2_addf, 16, 0},
    {"v_mul_f32", k2_mulf, 16, 0},
    {"v_sub_f32", k2_subf, 16, 0},
    {"v_min_u32", k2_minu, 16, 0},
    {"v_max_i32", k2_maxi, 16, 0},
    {"v_lshrrev_b32", k2_lshr, 16, 0},
    {"v_ashrrev_i32", k2_ashr, 16, 0},
    {"v_and_or_b32", k2_andor, 16, 0},
    {"v_bfi_b32", k2_bfi, 16, 0},
    {"v_not_b32", k2_not, 16, 0},
    {"v_mad_i32_i24", k2_madi24, 16, 0},
    {"v_fmac_f32", k2_fmac, 16, 0},
    {"v_cvt_f32_u32", k2_cvtfu, 16, 0},
    {"v_min3_f32", k2_min3f, 16, 0},
    {"v_med3_f32", k2_med3f, 16, 0},
    {"v_alignbit_b32", k2_alignbit, 16, 0},
    {"v_cmp_lt_f32", k2_cmpf, 16, 0},
    {"v_cmp_lt_u32", k2_cmpu, 16, 0},
    {"v_cmp_lt_i32", k2_cmpi, 16, 0},
    {"v_cmp_lt_u32", k2_cmpx, 16, 0},
    {"v_cndmask_b32", k2_cndsg, 16, 0},
    {"v_lshl_or_b32", k2_lshlor, 16, 0},
    {"v_or3_b32", k2_or3, 16, 0},
    {"v_xad_u32", k2_xad, 16, 0},
    {"v_add_lshl_u32", k2_addlshl, 16, 0},
    {"v_subrev_u32", k2_subrev, 16, 0},
    {"v_bitop3_b32", k2_bitop3, 16, 0},
    {"v_max_f32", k2_maxf64, 16, 0},
    {"v_rndne_f32", k2_rndne, 16, 0},
    {"v_floor_f32", k2_floor, 16, 0},
    {"v_mbcnt_lo_u32_b32", k2_mbcnt, 16, 0},
    {"v_readfirstlane_b32", k2_readlane, 16, 0},
    {"v_mul_legacy_f32", k2_mul_legacy, 16, 0},
    {"v_ldexp_f32", k2_ldexp, 16, 0},
    {"v_sad_u32", k2_sad, 16, 0},
    {"v_cvt_pk_u16_u32", k2_cvtpk, 16, 0},
    {"v_pk_mul_lo_u16", k2_pkmul, 16, 0},
    {"v_pk_max_i16", k2_pkmax, 16, 0},
    {"v_pk_lshlrev_b16", k2_pklshl, 16, 0},
    {"v_pk_sub_i16", k2_pksub, 16, 0},
    {"v_dot2_u32_u16", k2_dot2, 16, 0},
    {"v_dot4_u32_u8", k2_dot4, 16, 0},
    {"v_mad_u32_u16", k2_madu16, 16, 0},
    {"v_add_co_u32", k2_addco, 16, 0},
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
