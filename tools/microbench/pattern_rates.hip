// Micro-benchmark: cost of the instruction PATTERNS the sweep passes are made of (dependent chains, exec-mask regions,
// scalar work between vector work, LDS round trips, taken branches) on gfx950, as a function of waves per SIMD.
// Prints ns per pattern and SIMD (time / (iterations x waves per SIMD)).
// Build: hipcc --offload-arch=gfx950 -O3 -o pattern_rates pattern_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int kIters = 4000;

#define KERNEL(NAME, BODY)                                                                         \
  __global__ void NAME(uint32_t* out, uint32_t seed) {                                             \
    __shared__ uint32_t lds[4096];                                                                 \
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = threadIdx.x & 1, d = (threadIdx.x * 4) & 0x3ffc; \
    lds[threadIdx.x] = a;                                                                          \
    uint64_t wide = a;                                                                             \
    for (int it = 0; it < kIters; it++) { BODY }                                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (uint32_t)wide + lds[threadIdx.x]; }

#define ADD12 "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n" \
              "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n" \
              "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
#define MUL12 "v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %0, %0, %1\n" \
              "v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %0, %0, %1\n" \
              "v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %0, %0, %1\n"

// 12 dependent fast vector adds
KERNEL(p_chain_add, asm volatile(ADD12 : "+v"(a) : "v"(b));)
// 12 dependent 24-bit multiplies
KERNEL(p_chain_mul, asm volatile(MUL12 : "+v"(a) : "v"(b));)
// the same adds inside an exec-mask region that every lane enters
KERNEL(p_region_all, asm volatile("v_cmp_ne_u32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 12\n" ADD12
                                  "s_or_b64 exec, exec, s[20:21]\n" : "+v"(a) : "v"(b) : "vcc", "scc", "s20", "s21");)
// ... that every second lane enters
KERNEL(p_region_half, asm volatile("v_cmp_eq_u32 vcc, 1, %2\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 12\n" ADD12
                                   "s_or_b64 exec, exec, s[20:21]\n" : "+v"(a) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21");)
// ... that no lane enters (the branch is taken)
KERNEL(p_region_none, asm volatile("v_cmp_eq_u32 vcc, 7, %2\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 12\n" ADD12
                                   "s_or_b64 exec, exec, s[20:21]\n" : "+v"(a) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21");)
// three small regions of 4 adds each (a sweep-1 row has three)
#define REGION4 "v_cmp_ne_u32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 4\n" \
                "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n s_or_b64 exec, exec, s[20:21]\n"
KERNEL(p_three_regions, asm volatile(REGION4 REGION4 REGION4 : "+v"(a) : "v"(b) : "vcc", "scc", "s20", "s21");)
// 12 adds with 6 scalar adds in between
KERNEL(p_valu_salu, asm volatile("v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1\n"
                                 "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                                 "s_add_u32 s20, s20, 1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_u32 %0, %0, %1\n"
                                 "v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1\n v_add_u32 %0, %0, %1\n" : "+v"(a) : "v"(b) : "scc", "s20");)
// vector compare -> scalar use of vcc -> vector use of the scalar result (the dependency of every region test)
KERNEL(p_cmp_to_scalar, asm volatile("v_cmp_ne_u32 vcc, %0, %1\n s_and_b64 s[20:21], vcc, exec\n v_cndmask_b32 %0, %0, %1, s[20:21]\n"
                                     "v_cmp_ne_u32 vcc, %0, %1\n s_and_b64 s[20:21], vcc, exec\n v_cndmask_b32 %0, %0, %1, s[20:21]\n"
                                     "v_cmp_ne_u32 vcc, %0, %1\n s_and_b64 s[20:21], vcc, exec\n v_cndmask_b32 %0, %0, %1, s[20:21]\n"
                                     "v_cmp_ne_u32 vcc, %0, %1\n s_and_b64 s[20:21], vcc, exec\n v_cndmask_b32 %0, %0, %1, s[20:21]\n"
                                     : "+v"(a) : "v"(b) : "vcc", "scc", "s20", "s21");)
// LDS round trip: read, wait, 10 dependent adds
KERNEL(p_lds_roundtrip, asm volatile("ds_read_b32 %0, %2\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                                     "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                                     "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n" : "+v"(a) : "v"(b), "v"(d));)
// LDS write + read of the same address, wait, 10 adds
KERNEL(p_lds_write_read, asm volatile("ds_write_b32 %2, %0\n ds_read_b32 %0, %2\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                                      "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                                      "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n" : "+v"(a) : "v"(b), "v"(d));)
// the pop test: two 64-bit multiply-adds and a 64-bit compare, result used by a select
KERNEL(p_pop_test, asm volatile("v_mad_i64_i32 %2, s[20:21], %0, %1, 0\n v_mad_i64_i32 %2, s[20:21], %1, %0, %2\n v_cmp_gt_i64 vcc, 0, %2\n v_cndmask_b32 %0, %0, %1, vcc\n"
                                "v_mad_i64_i32 %2, s[20:21], %0, %1, 0\n v_mad_i64_i32 %2, s[20:21], %1, %0, %2\n v_cmp_gt_i64 vcc, 0, %2\n v_cndmask_b32 %0, %0, %1, vcc\n"
                                "v_mad_i64_i32 %2, s[20:21], %0, %1, 0\n v_mad_i64_i32 %2, s[20:21], %1, %0, %2\n v_cmp_gt_i64 vcc, 0, %2\n v_cndmask_b32 %0, %0, %1, vcc\n"
                                : "+v"(a), "+v"(b), "+v"(wide) : : "vcc", "scc", "s20", "s21");)
// a divergent loop that runs once: mask bookkeeping of `do { 8 adds } while (false for everyone)`
KERNEL(p_loop_once, asm volatile("s_mov_b64 s[22:23], 0\n"
                                 "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                                 "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                                 "v_cmp_ne_u32 vcc, %0, %0\n s_or_b64 s[22:23], vcc, s[22:23]\n s_mov_b64 s[20:21], exec\n s_andn2_b64 exec, exec, s[22:23]\n"
                                 "s_cbranch_execz 0\n s_mov_b64 exec, s[20:21]\n" : "+v"(a) : "v"(b) : "vcc", "scc", "s20", "s21", "s22", "s23");)

struct Entry { const char* name; void (*k)(uint32_t*, uint32_t); int valu; };

int main()
{
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t* out; CHECK(hipMalloc(&out, sizeof(uint32_t) * cus * 2 * 1024));
  const Entry es[] = {
    {"12 dependent v_add_u32", p_chain_add, 12}, {"12 dependent v_mul_u32_u24", p_chain_mul, 12},
    {"region, all lanes: cmp+saveexec+branch+12 adds+or", p_region_all, 13},
    {"region, every second lane", p_region_half, 13}, {"region, no lane (branch taken)", p_region_none, 1},
    {"three regions of 4 adds", p_three_regions, 15}, {"12 adds + 6 scalar adds", p_valu_salu, 12},
    {"4 x (cmp -> s_and -> cndmask)", p_cmp_to_scalar, 8},
    {"ds_read + wait + 10 adds", p_lds_roundtrip, 10}, {"ds_write + ds_read + wait + 10 adds", p_lds_write_read, 10},
    {"3 x (2 mad_i64_i32 + cmp_i64 + cndmask)", p_pop_test, 12}, {"loop run once around 8 adds", p_loop_once, 9},
  };
  printf("%-52s", "ns per pattern and SIMD");
  const int wps[] = {1, 2, 4, 8};
  for (int w : wps) printf("  %dw/SIMD", w);
  printf("\n");
  hipEvent_t ev0, ev1; CHECK(hipEventCreate(&ev0)); CHECK(hipEventCreate(&ev1));
  for (const Entry& e : es) {
    printf("%-52s", e.name); fflush(stdout);
    for (int w : wps) {
      const int threads = 256 * w;
      const int blocks = (threads > 1024) ? cus * 2 : cus;
      const int tpb = (threads > 1024) ? 1024 : threads;
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(tpb), 0, 0, out, 1u);
      CHECK(hipEventRecord(ev0, 0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(tpb), 0, 0, out, 2u);
      CHECK(hipEventRecord(ev1, 0));
      CHECK(hipDeviceSynchronize());
      float ms = 0; CHECK(hipEventElapsedTime(&ms, ev0, ev1));
      printf("  %8.2f", ms * 1e6 / (double(kIters) * w)); fflush(stdout);
    }
    printf("\n");
  }
  return 0;
}
