// mfma_probe.hip -- development probe (not part of the library).
//
// 1. Exactness and operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with FP4
//    (e2m1) operands built from sample BITS by masking: a 32-bit word of 32
//    sample bits is split into four words of eight nibbles each,
//    (w >> j) & 0x11111111 ... expressed without the shift for j = 0, 1, 2 as
//    w & (0x11111111 << j), whose nibbles are 0b0001 / 0b0010 / 0b0100 =
//    0.5 / 1.0 / 2.0 in e2m1; the block scales (E8M0) bring every product of
//    two set bits to exactly 1.0.  The contraction is a sum over samples, so any
//    assignment of samples to K positions is fine as long as A and B use the same.
// 2. Issue rate of that instruction and of v_mfma_i32_32x32x32_i8.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                              \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
      std::exit(1);                                                           \
    }                                                                         \
  } while (0)

__device__ __forceinline__ v8i frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  v8i f = {(int)a, (int)b, (int)c, (int)d, 0, 0, 0, 0};
  return f;
}

// C[32][32] = popcount(A[r] & B[c]) over `words` 32-bit words per row
// (words a multiple of 8); one wavefront.
__global__ void count_fp4(const uint32_t* A, const uint32_t* B, uint32_t words, uint32_t* C) {
  const uint32_t l = threadIdx.x, r = l & 31, h = l >> 5;
  v16f acc = {0};
  for (uint32_t w0 = 0; w0 < words; w0 += 8) {
    const uint32_t* pa = A + r * words + w0 + 4 * h;
    const uint32_t* pb = B + r * words + w0 + 4 * h;
    const uint32_t a0 = pa[0], a1 = pa[1], a2 = pa[2], a3 = pa[3];
    const uint32_t b0 = pb[0], b1 = pb[1], b2 = pb[2], b3 = pb[3];
    const uint32_t m0 = 0x11111111u, m1 = 0x22222222u, m2 = 0x44444444u;
    // nibble 0b0001 = 0.5: scale 2 on each side (E8M0 128)
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
        frag(a0 & m0, a1 & m0, a2 & m0, a3 & m0), frag(b0 & m0, b1 & m0, b2 & m0, b3 & m0), acc, 4, 4,
        0, 0x80808080, 0, 0x80808080);
    // nibble 0b0010 = 1.0
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
        frag(a0 & m1, a1 & m1, a2 & m1, a3 & m1), frag(b0 & m1, b1 & m1, b2 & m1, b3 & m1), acc, 4, 4,
        0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    // nibble 0b0100 = 2.0: scale 1/2 on each side (E8M0 126)
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
        frag(a0 & m2, a1 & m2, a2 & m2, a3 & m2), frag(b0 & m2, b1 & m2, b2 & m2, b3 & m2), acc, 4, 4,
        0, 0x7e7e7e7e, 0, 0x7e7e7e7e);
    // bit 3 of a nibble is the sign: shift it down to 0b0001
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
        frag((a0 >> 3) & m0, (a1 >> 3) & m0, (a2 >> 3) & m0, (a3 >> 3) & m0),
        frag((b0 >> 3) & m0, (b1 >> 3) & m0, (b2 >> 3) & m0, (b3 >> 3) & m0), acc, 4, 4, 0, 0x80808080,
        0, 0x80808080);
  }
  // C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  for (int reg = 0; reg < 16; ++reg) {
    const uint32_t row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
    C[row * 32 + r] = (uint32_t)acc[reg];
  }
}

// The same with the int8 form: every sample bit becomes a 0/1 byte.
__global__ void count_i8(const uint32_t* A, const uint32_t* B, uint32_t words, uint32_t* C) {
  const uint32_t l = threadIdx.x, r = l & 31, h = l >> 5;
  v16i acc = {0};
  for (uint32_t w0 = 0; w0 < words; ++w0) {
    // lane half h takes bits 16 h .. 16 h + 15 of the word: four nibbles -> four words of bytes
    const uint32_t a = A[r * words + w0] >> (16 * h), b = B[r * words + w0] >> (16 * h);
    v4i fa, fb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[i] = (int)((((a >> (4 * i)) & 15u) * 0x204081u) & 0x01010101u);
      fb[i] = (int)((((b >> (4 * i)) & 15u) * 0x204081u) & 0x01010101u);
    }
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc, 0, 0, 0);
  }
  for (int reg = 0; reg < 16; ++reg) {
    const uint32_t row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
    C[row * 32 + r] = (uint32_t)acc[reg];
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, uint32_t seed) {
  const uint32_t l = threadIdx.x;
  uint32_t x = seed * 2654435761u + l * 40503u + blockIdx.x;
  v8i fa = {(int)(x & 0x22222222u), (int)((x * 3) & 0x22222222u), (int)((x * 5) & 0x22222222u),
            (int)((x * 7) & 0x22222222u), 0, 0, 0, 0};
  v8i fb = {(int)((x * 11) & 0x22222222u), (int)((x * 13) & 0x22222222u), (int)((x * 17) & 0x22222222u),
            (int)((x * 19) & 0x22222222u), 0, 0, 0, 0};
  if constexpr (MODE == 0 || MODE == 2) {
    v16f a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    for (int i = 0; i < iters; ++i) {
      if constexpr (MODE == 0) {
        a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, a0, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, a1, 4, 4, 0, 0x80808080, 0, 0x7e7e7e7e);
        a2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb, fa, a2, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        a3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb, fa, a3, 4, 4, 0, 0x7e7e7e7e, 0, 0x80808080);
      } else {  // scales 0, 0: the unscaled instruction
        a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, a0, 4, 4, 0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, a1, 4, 4, 0, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb, fa, a2, 4, 4, 0, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb, fa, a3, 4, 4, 0, 0, 0, 0);
      }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    out[blockIdx.x * 256 + l] = s;
  } else {
    v16i a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    v4i ia = {fa[0] & 0x01010101, fa[1] & 0x01010101, fa[2] & 0x01010101, fa[3] & 0x01010101};
    v4i ib = {fb[0] & 0x01010101, fb[1] & 0x01010101, fb[2] & 0x01010101, fb[3] & 0x01010101};
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia, ib, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ib, ia, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ib, ia, a3, 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    out[blockIdx.x * 256 + l] = (float)s;
  }
}

int loop_main();
int loop2x2_main();
int main() {
  const uint32_t words = 64;  // 2048 samples
  std::vector<uint32_t> A(32 * words), B(32 * words), want(32 * 32), have(32 * 32);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&] {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return (uint32_t)(s >> 16);
  };
  for (auto& x : A) x = rnd() & rnd();
  for (auto& x : B) x = rnd() | (rnd() & rnd());
  for (int r = 0; r < 32; ++r)
    for (int c = 0; c < 32; ++c) {
      uint32_t t = 0;
      for (uint32_t w = 0; w < words; ++w) t += __builtin_popcount(A[r * words + w] & B[c * words + w]);
      want[r * 32 + c] = t;
    }
  uint32_t *dA, *dB, *dC;
  CHECK(hipMalloc(&dA, A.size() * 4));
  CHECK(hipMalloc(&dB, B.size() * 4));
  CHECK(hipMalloc(&dC, want.size() * 4));
  CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  for (int mode = 0; mode < 2; ++mode) {
    CHECK(hipMemset(dC, 0xff, want.size() * 4));
    if (mode == 0) hipLaunchKernelGGL(count_fp4, dim3(1), dim3(64), 0, 0, dA, dB, words, dC);
    else hipLaunchKernelGGL(count_i8, dim3(1), dim3(64), 0, 0, dA, dB, words, dC);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(have.data(), dC, have.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 32 * 32; ++i) bad += have[i] != want[i];
    std::printf("%s: %d of 1024 cells differ (cell[1][2]: have %u want %u)\n", mode == 0 ? "fp4" : "i8",
                bad, have[1 * 32 + 2], want[1 * 32 + 2]);
  }
  // issue rate: 1024 workgroups of 4 waves (one per SIMD, four workgroups per CU)
  float* dout;
  CHECK(hipMalloc(&dout, 4096 * 256 * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int blocks : {256, 512, 1024}) {
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, dout, iters, 1u);
        if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, dout, iters, 1u);
        if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(blocks), dim3(256), 0, 0, dout, iters, 1u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 1) {
          const double mfma = (double)blocks * 4 * iters * 4;
          const double per_simd_ns = ms * 1e6 / (mfma / 1024.0);
          std::printf("%-10s blocks %4d: %.3f ms, %.2f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)\n",
                      mode == 0 ? "fp4 scaled" : mode == 1 ? "i8" : "fp4 plain", blocks, ms, per_simd_ns,
                      per_simd_ns * 2.4);
        }
      }
    }
  }
  return loop_main();
}

// ---- inner-loop probe: the operand build + MFMA stream of the engine's tile loop,
// fed from LDS, without global loads.  VAR 0: as the engine; 1: no operand build
// (raw words as operands); 2: operands built once outside the loop (no LDS reads).
typedef int pv8i __attribute__((ext_vector_type(8)));
template <int J>
__device__ __forceinline__ pv8i p_operand(const uint4 w) {
  pv8i f;
  if constexpr (J < 3) {
    constexpr uint32_t m = 0x11111111u << J;
    f[0] = (int)(w.x & m); f[1] = (int)(w.y & m); f[2] = (int)(w.z & m); f[3] = (int)(w.w & m);
  } else {
    f[0] = (int)((w.x >> 3) & 0x11111111u); f[1] = (int)((w.y >> 3) & 0x11111111u);
    f[2] = (int)((w.z >> 3) & 0x11111111u); f[3] = (int)((w.w >> 3) & 0x11111111u);
  }
  f[4] = f[5] = f[6] = f[7] = 0;
  return f;
}
__device__ __forceinline__ pv8i p_raw(const uint4 w) {
  pv8i f = {(int)w.x, (int)w.y, (int)w.z, (int)w.w, 0, 0, 0, 0};
  return f;
}
template <int VAR, int JM>
__global__ __launch_bounds__(256) void loop_kernel(float* out, int steps, unsigned long long* stamps) {
  constexpr int STRIDE = 36;
  __shared__ __attribute__((aligned(16))) uint32_t smem[2 * 2 * 128 * STRIDE];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lr = lane & 31, h4 = (lane >> 5) * 4;
  for (uint32_t i = tid; i < 2 * 2 * 128 * STRIDE; i += 256) smem[i] = i * 2654435761u;
  __syncthreads();
  v16f acc[JM];
  for (int j = 0; j < JM; ++j)
    for (int i = 0; i < 16; ++i) acc[j][i] = 0;
  const uint32_t* sA = smem;
  const uint32_t* sB = smem + 128 * STRIDE;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < steps; ++s) {
#pragma unroll 1
    for (int rd = 0; rd < 4; ++rd) {
      uint4 wa, wb[JM];
      if (VAR != 2 || (s == 0 && rd == 0)) {
        wa = *reinterpret_cast<const uint4*>(&sA[(wv * 32 + lr) * STRIDE + rd * 8 + h4]);
#pragma unroll
        for (int j = 0; j < JM; ++j) wb[j] = *reinterpret_cast<const uint4*>(&sB[(32 * j + lr) * STRIDE + rd * 8 + h4]);
      }
      __builtin_amdgcn_sched_barrier(0);
#define BLK(J, SC)                                                                                  \
  {                                                                                                 \
    const pv8i a = VAR == 0 ? p_operand<J>(wa) : p_raw(wa);                                         \
    _Pragma("unroll") for (int j = 0; j < JM; ++j) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4( \
        a, VAR == 0 ? p_operand<J>(wb[j]) : p_raw(wb[j]), acc[j], 4, 4, 0, SC, 0, SC);              \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  }
      BLK(0, 0x80808080)
      BLK(1, 0x7f7f7f7f)
      BLK(2, 0x7e7e7e7e)
      BLK(3, 0x80808080)
#undef BLK
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    stamps[(blockIdx.x * 4 + wv) * 2] = c1 - c0;
    stamps[(blockIdx.x * 4 + wv) * 2 + 1] = r1 - r0;
  }
  float sum = 0;
  for (int j = 0; j < JM; ++j)
    for (int i = 0; i < 16; ++i) sum += acc[j][i];
  out[blockIdx.x * 256 + tid] = sum;
}

template <int VAR, int JM>
void run_loop(float* dout, int blocks, const char* name) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int steps = 4000;
  static unsigned long long* dst = nullptr;
  if (!dst) CHECK(hipMalloc(&dst, 4096 * 4 * 2 * 8));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((loop_kernel<VAR, JM>), dim3(blocks), dim3(256), 0, 0, dout, steps, dst);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 1) {
      const double mfma_per_simd = (double)blocks * 4 * steps * 16 * JM / 1024.0;
      std::vector<unsigned long long> st((size_t)blocks * 8);
      CHECK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> clk, cyc;
      for (int w = 0; w < blocks * 4; ++w) {
        clk.push_back((double)st[2 * w] / (double)st[2 * w + 1] * 0.1);          // GHz (realtime ticks at 100 MHz)
        cyc.push_back((double)st[2 * w] / ((double)steps * 16 * JM));            // wave cycles per own MFMA
      }
      std::sort(clk.begin(), clk.end());
      std::sort(cyc.begin(), cyc.end());
      std::printf("loop %-28s JM %d blocks %4d: %.3f ms, %.2f ns per MFMA per SIMD; clock %.2f GHz, %.1f wave cycles per own MFMA\n",
                  name, JM, blocks, ms, ms * 1e6 / mfma_per_simd, clk[clk.size() / 2], cyc[cyc.size() / 2]);
    }
  }
}

int loop_main() {
  float* dout;
  CHECK(hipMalloc(&dout, 4096 * 256 * 4));
  for (int blocks : {256, 512}) {
    run_loop<0, 4>(dout, blocks, "engine form");
    run_loop<1, 4>(dout, blocks, "no operand build");
    run_loop<2, 4>(dout, blocks, "no LDS reads, raw operands");
    run_loop<0, 1>(dout, blocks, "engine form");
    run_loop<0, 2>(dout, blocks, "engine form");
  }
  return loop2x2_main();
}

// ---- pipelined operand build: the operands of block j + 1 are built while the
// MFMAs of block j issue (2 x 2 blocks per wave), scheduled explicitly.
template <int J>
__device__ __forceinline__ int p_scale() { return (J == 1) ? 0x7f7f7f7f : (J == 2) ? 0x7e7e7e7e : (int)0x80808080u; }

struct Masks { uint32_t m0, m1, m2; };
// the masks in scalar registers the compiler cannot see through: v_and_b32 with an SGPR
// operand is a 4-byte instruction, with a 32-bit literal an 8-byte one
__device__ __forceinline__ Masks opaque_masks() {
  Masks m;
  asm volatile("s_mov_b32 %0, 0x11111111\n\ts_mov_b32 %1, 0x22222222\n\ts_mov_b32 %2, 0x44444444"
               : "=s"(m.m0), "=s"(m.m1), "=s"(m.m2));
  return m;
}
template <int J>
__device__ __forceinline__ pv8i p_operand_m(const uint4 w, const Masks& mk) {
  pv8i f;
  if constexpr (J < 3) {
    const uint32_t m = J == 0 ? mk.m0 : J == 1 ? mk.m1 : mk.m2;
    f[0] = (int)(w.x & m); f[1] = (int)(w.y & m); f[2] = (int)(w.z & m); f[3] = (int)(w.w & m);
  } else {
    f[0] = (int)((w.x >> 3) & mk.m0); f[1] = (int)((w.y >> 3) & mk.m0);
    f[2] = (int)((w.z >> 3) & mk.m0); f[3] = (int)((w.w >> 3) & mk.m0);
  }
  f[4] = f[5] = f[6] = f[7] = 0;
  return f;
}
static __device__ Masks g_mk_dummy;
template <int J, bool OPAQUE = false>
__device__ __forceinline__ void build2x2(const uint4 (&wa)[2], const uint4 (&wb)[2], pv8i (&a)[2], pv8i (&b)[2],
                                         const Masks& mk = Masks{0x11111111u, 0x22222222u, 0x44444444u}) {
  if constexpr (OPAQUE) {
    a[0] = p_operand_m<J>(wa[0], mk); a[1] = p_operand_m<J>(wa[1], mk);
    b[0] = p_operand_m<J>(wb[0], mk); b[1] = p_operand_m<J>(wb[1], mk);
  } else {
    a[0] = p_operand<J>(wa[0]); a[1] = p_operand<J>(wa[1]);
    b[0] = p_operand<J>(wb[0]); b[1] = p_operand<J>(wb[1]);
  }
}
template <int J>
__device__ __forceinline__ void mfma2x2(const pv8i (&a)[2], const pv8i (&b)[2], v16f (&acc)[2][2]) {
  constexpr int SC = (J == 1) ? 0x7f7f7f7f : (J == 2) ? 0x7e7e7e7e : (int)0x80808080u;
  acc[0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b[0], acc[0][0], 4, 4, 0, SC, 0, SC);
  acc[1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[1], b[0], acc[1][0], 4, 4, 0, SC, 0, SC);
  acc[0][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[0], b[1], acc[0][1], 4, 4, 0, SC, 0, SC);
  acc[1][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[1], b[1], acc[1][1], 4, 4, 0, SC, 0, SC);
}

// MODE 0: 2 x 2, build-then-issue per block with fences (the engine's form);
// MODE 1: software pipelined: build(j+1) interleaved with mfma(j) by sched_group_barrier.
// KW: words per slab row (32: two workgroups per CU fit; 16: three or four); MINB: workgroups per CU asked for
template <int MODE, int KW = 32, int MINB = 1>
__global__ __launch_bounds__(256, MINB) void loop2x2_kernel(float* out, int steps, unsigned long long* stamps) {
  constexpr int STRIDE = KW + 4;
  __shared__ __attribute__((aligned(16))) uint32_t smem[2 * 2 * 128 * STRIDE];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lr = lane & 31, h4 = (lane >> 5) * 4;
  for (uint32_t i = tid; i < 2 * 2 * 128 * STRIDE; i += 256) smem[i] = i * 2654435761u;
  __syncthreads();
  v16f acc[2][2];
  for (int a = 0; a < 2; ++a)
    for (int j = 0; j < 2; ++j)
      for (int i = 0; i < 16; ++i) acc[a][j][i] = 0;
  const uint32_t* pa = smem + ((wv >> 1) * 64 + lr) * STRIDE + h4;
  const uint32_t* pb = smem + 128 * STRIDE + ((wv & 1) * 64 + lr) * STRIDE + h4;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  uint4 wa[2], wb[2];
  for (int i = 0; i < 2; ++i) {
    wa[i] = *reinterpret_cast<const uint4*>(pa + 32 * i * STRIDE);
    wb[i] = *reinterpret_cast<const uint4*>(pb + 32 * i * STRIDE);
  }
  pv8i oa[2], ob[2];
  const Masks mk = opaque_masks();
  constexpr bool OPQ = MODE >= 2;
  if (MODE == 1 || MODE == 3) build2x2<0, OPQ>(wa, wb, oa, ob, mk);
  for (int s = 0; s < steps; ++s) {
#pragma unroll 1
    for (int rd = 0; rd < 4; ++rd) {
      const int nx = ((rd + 1) & (KW / 8 - 1)) * 8;
      uint4 na[2], nb[2];
      for (int i = 0; i < 2; ++i) {
        na[i] = *reinterpret_cast<const uint4*>(pa + 32 * i * STRIDE + nx);
        nb[i] = *reinterpret_cast<const uint4*>(pb + 32 * i * STRIDE + nx);
      }
      if (MODE == 0 || MODE == 2) {
        __builtin_amdgcn_sched_barrier(0);
        { pv8i a[2], b[2]; build2x2<0, OPQ>(wa, wb, a, b, mk); mfma2x2<0>(a, b, acc); __builtin_amdgcn_sched_barrier(0); }
        { pv8i a[2], b[2]; build2x2<1, OPQ>(wa, wb, a, b, mk); mfma2x2<1>(a, b, acc); __builtin_amdgcn_sched_barrier(0); }
        { pv8i a[2], b[2]; build2x2<2, OPQ>(wa, wb, a, b, mk); mfma2x2<2>(a, b, acc); __builtin_amdgcn_sched_barrier(0); }
        { pv8i a[2], b[2]; build2x2<3, OPQ>(wa, wb, a, b, mk); mfma2x2<3>(a, b, acc); __builtin_amdgcn_sched_barrier(0); }
      } else {
        // block j: issue its 4 MFMAs (operands built one block earlier) with the 16-32 vector
        // instructions of block j + 1's build spread between them
        pv8i pa1[2], pb1[2];
#define PIPE(J, JN, SRCA, SRCB)                                                  \
        mfma2x2<J>(oa, ob, acc);                                                 \
        build2x2<JN, OPQ>(SRCA, SRCB, pa1, pb1, mk);                             \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                          \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* 1 MFMA */        \
          __builtin_amdgcn_sched_group_barrier(0x002, 8, 0); /* up to 8 VALU */  \
        }                                                                        \
        oa[0] = pa1[0]; oa[1] = pa1[1]; ob[0] = pb1[0]; ob[1] = pb1[1];
        PIPE(0, 1, wa, wb)
        PIPE(1, 2, wa, wb)
        PIPE(2, 3, wa, wb)
        PIPE(3, 0, na, nb)
#undef PIPE
      }
      for (int i = 0; i < 2; ++i) { wa[i] = na[i]; wb[i] = nb[i]; }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    stamps[(blockIdx.x * 4 + wv) * 2] = c1 - c0;
    stamps[(blockIdx.x * 4 + wv) * 2 + 1] = r1 - r0;
  }
  float sum = 0;
  for (int a = 0; a < 2; ++a)
    for (int j = 0; j < 2; ++j)
      for (int i = 0; i < 16; ++i) sum += acc[a][j][i];
  out[blockIdx.x * 256 + tid] = sum;
}

template <int MODE, int KW = 32, int MINB = 1>
void run_loop2x2(float* dout, int blocks, const char* name) {
  const int steps = 4000;
  static unsigned long long* dst = nullptr;
  if (!dst) CHECK(hipMalloc(&dst, 4096 * 4 * 2 * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((loop2x2_kernel<MODE, KW, MINB>), dim3(blocks), dim3(256), 0, 0, dout, steps, dst);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep == 1) {
      std::vector<unsigned long long> st((size_t)blocks * 8);
      CHECK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> clk, cyc;
      for (int w = 0; w < blocks * 4; ++w) {
        clk.push_back((double)st[2 * w] / (double)st[2 * w + 1] * 0.1);
        cyc.push_back((double)st[2 * w] / ((double)steps * 16 * 4));
      }
      std::sort(clk.begin(), clk.end());
      std::sort(cyc.begin(), cyc.end());
      const double mfma_per_simd = (double)blocks * 4 * steps * 16 * 4 / 1024.0;
      std::printf("loop2x2 %-24s blocks %4d: %.3f ms, %.2f ns per MFMA per SIMD; clock %.2f GHz, %.1f wave cycles per own MFMA\n",
                  name, blocks, ms, ms * 1e6 / mfma_per_simd, clk[clk.size() / 2], cyc[cyc.size() / 2]);
    }
  }
}

int loop2x2_main() {
  float* dout;
  CHECK(hipMalloc(&dout, 4096 * 256 * 4));
  for (int blocks : {256, 512}) {
    run_loop2x2<0>(dout, blocks, "build-then-issue");
    run_loop2x2<1>(dout, blocks, "pipelined build");
    run_loop2x2<2>(dout, blocks, "build-then-issue, sgpr masks");
    run_loop2x2<3>(dout, blocks, "pipelined, sgpr masks");
  }
  // three and four waves per SIMD (16-word slab rows: 40 KB of LDS per workgroup)
  run_loop2x2<0, 16, 3>(dout, 768, "build-then-issue, 3/SIMD");
  run_loop2x2<1, 16, 3>(dout, 768, "pipelined build, 3/SIMD");
  run_loop2x2<0, 16, 4>(dout, 1024, "build-then-issue, 4/SIMD");
  return 0;
}
