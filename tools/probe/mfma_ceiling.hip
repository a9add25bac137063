// What this board sustains on bare bf16 MFMAs ("measure the ceiling you claim").
// Register-only loops -- no LDS, no global memory inside the timed part -- of v_mfma_f32_32x32x16_bf16 (the shape every conv
// kernel of the library uses) and v_mfma_f32_16x16x32_bf16 on RANDOM bf16 operands (and, for contrast, on zeros: the chip
// lowers its clock with the switching activity of the operands), all 256 CUs, one or two waves per SIMD, launched back to
// back for >= 5 s per variant.  Reported per variant: TFLOP/s by HIP events over the whole sustained run and over its last
// second, and the in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz, median over workgroups of the last launch).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/mfma_ceiling tools/probe/mfma_ceiling.hip
// Run beside `rocm-smi` sampling: tools/mfma_ceiling.sh
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t& s) {
  s = s * 1664525u + 1013904223u;
  return s;
}
// two bf16 values in (-2, 2) with random mantissas and signs (exponents 126..127: no denormals, no overflow of the f32 sums
// within 10^6 MFMAs)
__device__ __forceinline__ uint32_t rnd_pair(uint32_t& s, int zero) {
  if (zero) return 0u;
  const uint32_t r0 = lcg(s) >> 8, r1 = lcg(s) >> 8;
  auto one = [](uint32_t r) { return ((r & 1u) << 15) | ((126u + ((r >> 1) & 1u)) << 7) | ((r >> 2) & 0x7Fu); };
  return one(r0) | (one(r1) << 16);
}

struct Stamp {
  unsigned long long t0, t1, r0, r1;
};

template <int SHAPE>   // 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(512) void mfma_loop(int iters, int zero, float* sink, Stamp* stamps) {
  uint32_t seed = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  u32x4 a[4], b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[j][e] = rnd_pair(seed, zero);
      b[j][e] = rnd_pair(seed, zero);
    }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float out = 0.f;
  if constexpr (SHAPE == 0) {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(j + u) & 3]), __builtin_bit_cast(bf16x8, b[j]),
                                                           acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) out += acc[j][i];
  } else {
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j)     // 32 MFMAs of half the FLOPs each per trip
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[(j + u) & 3]), __builtin_bit_cast(bf16x8, b[j & 3]),
                                                           acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) out += acc[j][i];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) stamps[blockIdx.x] = Stamp{t0, t1, r0, r1};
  if (out == 123.456f) sink[0] = out;     // keeps the accumulators alive
}

static double run_variant(const char* name, int shape, int threads, int zero, double seconds, float* sink, Stamp* dstamps, int blocks) {
  const int iters = 20000;        // 320k MFMA-equivalents of 32768 FLOP per wave and launch: ~4-6 ms per launch
  const double flop_launch = (double)blocks * (threads / 64) * iters * 16.0 * 32768.0;
  hipEvent_t e0, e1, em;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&em));
  auto launch = [&]() {
    if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(threads), 0, 0, iters, zero, sink, dstamps);
    else hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(threads), 0, 0, iters, zero, sink, dstamps);
  };
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipDeviceSynchronize());
  // size the run from one timed launch, then go: everything is enqueued at once so the GPU never idles between launches
  CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms1 = 0.f; CHECK(hipEventElapsedTime(&ms1, e0, e1));
  const int n = std::max(8, (int)(seconds * 1e3 / ms1));
  const int ntail = std::max(1, (int)(1e3 / ms1));
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < n; ++i) {
    if (i == n - ntail) CHECK(hipEventRecord(em));
    launch();
  }
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f, mst = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipEventElapsedTime(&mst, em, e1));
  std::vector<Stamp> st(blocks);
  CHECK(hipMemcpy(st.data(), dstamps, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost));
  std::vector<double> clk;
  for (auto& s : st)
    if (s.r1 > s.r0) clk.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1);   // GHz: s_memrealtime ticks at 100 MHz
  std::sort(clk.begin(), clk.end());
  const double tf_all = flop_launch * n / (ms * 1e-3) / 1e12, tf_tail = flop_launch * ntail / (mst * 1e-3) / 1e12;
  const double cyc_per_mfma = clk.empty() ? 0.0 : 0.0;
  (void)cyc_per_mfma;
  printf("%-34s %4d launches %6.2f s  sustained %7.1f TFLOP/s  last second %7.1f TFLOP/s  in-kernel clock median %.3f GHz (min %.3f max %.3f)  "
         "-> %.2f cycles per 32x32x16-equivalent MFMA per SIMD\n",
         name, n, ms * 1e-3, tf_all, tf_tail, clk.empty() ? 0.0 : clk[clk.size() / 2], clk.empty() ? 0.0 : clk.front(),
         clk.empty() ? 0.0 : clk.back(),
         clk.empty() ? 0.0 : (clk[clk.size() / 2] * 1e9) / (tf_tail * 1e12 / 32768.0 / (256.0 * 4.0)));
  fflush(stdout);
  return tf_tail;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 6.0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clockRate %.0f MHz; %.1f s per variant\n", prop.gcnArchName, cus, prop.clockRate / 1e3, seconds);
  float* sink;
  Stamp* stamps;
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMalloc(&stamps, sizeof(Stamp) * cus * 2));
  double best = 0.0;
  best = std::max(best, run_variant("32x32x16 random, 1 wave/SIMD", 0, 256, 0, seconds, sink, stamps, cus));
  best = std::max(best, run_variant("32x32x16 random, 2 waves/SIMD", 0, 512, 0, seconds, sink, stamps, cus));
  run_variant("16x16x32 random, 1 wave/SIMD", 1, 256, 0, seconds, sink, stamps, cus);
  run_variant("16x16x32 random, 2 waves/SIMD", 1, 512, 0, seconds, sink, stamps, cus);
  run_variant("32x32x16 zeros, 1 wave/SIMD", 0, 256, 1, seconds, sink, stamps, cus);
  printf("SUSTAINED_PEAK_32x32x16_TFLOPS %.1f\n", best);
  return 0;
}
