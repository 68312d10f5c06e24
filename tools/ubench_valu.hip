// Development microbenchmark: issue rate of single VALU instructions on gfx950
// (inline asm, independent destination registers, 16 waves per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a = seed * (threadIdx.x + 1), b = seed ^ threadIdx.x;
  uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
  uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0)
      { REP16(asm volatile("v_and_b32 %0, %8, %9\n v_and_b32 %1, %8, %9\n v_and_b32 %2, %8, %9\n v_and_b32 %3, %8, %9\n"
                         "v_and_b32 %4, %8, %9\n v_and_b32 %5, %8, %9\n v_and_b32 %6, %8, %9\n v_and_b32 %7, %8, %9\n"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a), "v"(b));) }
    if (MODE == 1)
      { REP16(asm volatile("v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n"
                         "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));) }
    if (MODE == 2)
      { REP16(asm volatile("v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n"
                         "v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));) }
    if (MODE == 3)
      { REP16(asm volatile("v_mbcnt_lo_u32_b32 %0, %8, %0\n v_mbcnt_lo_u32_b32 %1, %8, %1\n v_mbcnt_lo_u32_b32 %2, %8, %2\n v_mbcnt_lo_u32_b32 %3, %8, %3\n"
                         "v_mbcnt_lo_u32_b32 %4, %8, %4\n v_mbcnt_lo_u32_b32 %5, %8, %5\n v_mbcnt_lo_u32_b32 %6, %8, %6\n v_mbcnt_lo_u32_b32 %7, %8, %7\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));) }
    if (MODE == 4)
      { REP16(asm volatile("v_and_b32 %0, %8, %9\n v_bcnt_u32_b32 %1, %0, %1\n v_and_b32 %2, %8, %9\n v_bcnt_u32_b32 %3, %2, %3\n"
                         "v_and_b32 %4, %8, %9\n v_bcnt_u32_b32 %5, %4, %5\n v_and_b32 %6, %8, %9\n v_bcnt_u32_b32 %7, %6, %7\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));) }
    if (MODE == 5)
      { REP16(asm volatile("v_sad_u32 %0, %8, %9, %0\n v_sad_u32 %1, %8, %9, %1\n v_sad_u32 %2, %8, %9, %2\n v_sad_u32 %3, %8, %9, %3\n"
                         "v_sad_u32 %4, %8, %9, %4\n v_sad_u32 %5, %8, %9, %5\n v_sad_u32 %6, %8, %9, %6\n v_sad_u32 %7, %8, %9, %7\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));) }
    if (MODE == 6)  // 8 ANDs into temporaries, then 8 popcounts into 8 accumulators: no back-to-back dependence
      { REP16(asm volatile("v_and_b32 %8, %16, %17\n v_and_b32 %9, %16, %17\n v_and_b32 %10, %16, %17\n v_and_b32 %11, %16, %17\n"
                         "v_and_b32 %12, %16, %17\n v_and_b32 %13, %16, %17\n v_and_b32 %14, %16, %17\n v_and_b32 %15, %16, %17\n"
                         "v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %9, %1\n v_bcnt_u32_b32 %2, %10, %2\n v_bcnt_u32_b32 %3, %11, %3\n"
                         "v_bcnt_u32_b32 %4, %12, %4\n v_bcnt_u32_b32 %5, %13, %5\n v_bcnt_u32_b32 %6, %14, %6\n v_bcnt_u32_b32 %7, %15, %7\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7),
                           "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                         : "v"(a), "v"(b));) }
    if (MODE == 7)  // the kernel's present pattern: and -> popcount back to back, four in a row on one accumulator
      { REP16(asm volatile("v_and_b32 %8, %9, %10\n v_bcnt_u32_b32 %0, %8, %0\n v_and_b32 %8, %9, %10\n v_bcnt_u32_b32 %0, %8, %0\n"
                         "v_and_b32 %8, %9, %10\n v_bcnt_u32_b32 %0, %8, %0\n v_and_b32 %8, %9, %10\n v_bcnt_u32_b32 %0, %8, %0\n"
                         "v_and_b32 %8, %9, %10\n v_bcnt_u32_b32 %1, %8, %1\n v_and_b32 %8, %9, %10\n v_bcnt_u32_b32 %1, %8, %1\n"
                         "v_and_b32 %8, %9, %10\n v_bcnt_u32_b32 %1, %8, %1\n v_and_b32 %8, %9, %10\n v_bcnt_u32_b32 %1, %8, %1\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "=&v"(t0)
                         : "v"(a), "v"(b));) }
    if (MODE == 8)  // interleaved but two instructions apart: and(i+1) between and(i) and popcount(i)
      { REP16(asm volatile("v_and_b32 %8, %16, %17\n v_and_b32 %9, %16, %17\n v_bcnt_u32_b32 %0, %8, %0\n v_and_b32 %10, %16, %17\n v_bcnt_u32_b32 %1, %9, %1\n"
                         "v_and_b32 %11, %16, %17\n v_bcnt_u32_b32 %2, %10, %2\n v_and_b32 %12, %16, %17\n v_bcnt_u32_b32 %3, %11, %3\n"
                         "v_and_b32 %13, %16, %17\n v_bcnt_u32_b32 %4, %12, %4\n v_and_b32 %14, %16, %17\n v_bcnt_u32_b32 %5, %13, %5\n"
                         "v_and_b32 %15, %16, %17\n v_bcnt_u32_b32 %6, %14, %6\n v_bcnt_u32_b32 %7, %15, %7\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7),
                           "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                         : "v"(a), "v"(b));) }
  }
  out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7;
}
template <int MODE> void run(const char* name) {
  uint32_t* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000;
  for (int wpc : {4, 8, 16, 32}) {
    int blocks = 256 * wpc / 4;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 12345u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const int per_iter = MODE >= 6 ? 256 : 128;                         // instructions per loop iteration
    double winstr = (double)blocks * 4 * iters * per_iter;
    double per_simd_ns = ms * 1e6 / ((double)wpc / 4 * iters * per_iter);   // ns per wave-instruction per SIMD
    printf("%-10s waves/CU=%2d: %.3f ms  %.2f ns per wave-instr per SIMD  (%.2f cycles @2.4GHz, %.2f @2.0GHz)\n", name, wpc, ms,
           per_simd_ns, per_simd_ns * 2.4, per_simd_ns * 2.0);
    (void)winstr;
  }
  (void)hipFree(d);
}
int main() {
  run<0>("v_and");
  run<2>("v_add");
  run<1>("v_bcnt");
  run<3>("v_mbcnt_lo");
  run<5>("v_sad_u32");
  run<4>("and+bcnt");
  run<6>("8and,8bcnt");
  run<7>("chain4");
  run<8>("skewed");
  return 0;
}
