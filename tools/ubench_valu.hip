// Development microbenchmark: issue rate of v_and_b32 / v_bcnt_u32_b32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a[16], acc[16];
  for (int i = 0; i < 16; ++i) { a[i] = seed * (threadIdx.x + 1 + i); acc[i] = i; }
  uint32_t b = seed ^ threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (MODE == 0) acc[i] = (acc[i] & a[i]) + 0;            // v_and only (dependent per acc, 16 chains)
        if (MODE == 1) acc[i] += __popc(a[i]);                  // v_bcnt accumulate
        if (MODE == 2) acc[i] += __popc(a[i] & b);              // and + bcnt
        if (MODE == 3) acc[i] = acc[i] * 3 + a[i];              // v_mad_u32_u24-ish / mul-add
      }
      b = b * 1664525u + 1013904223u;
      if (MODE != 2) { for (int i = 0; i < 16; ++i) a[i] ^= b; }
    }
  }
  uint32_t s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int ops_per_inner) {
  uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wpc : {4, 8, 16}) {   // waves per CU via blocks of 256 threads: blocks = 256 CUs * wpc/4
    int blocks = 256 * wpc / 4;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters * 4 * 16 * ops_per_inner;
    printf("%s waves/CU=%d: %.3f ms, %.2f wave-instr/cycle/CU @2.4GHz (counted ops only)\n", name, wpc, ms,
           winstr / (ms * 1e-3) / 256 / 2.4e9);
  }
  hipFree(d);
}
int main() {
  run<0>("and", 1);
  run<1>("bcnt", 1);
  run<2>("and+bcnt", 2);
  run<3>("mad", 1);
  return 0;
}
