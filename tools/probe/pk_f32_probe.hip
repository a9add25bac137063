// Packed-f32 VALU ops beside MFMAs.  Found through the masked epilogue of the sliding-halo conv kernel: 32 elements of
// `v += bit ? (slope - 1) * v : 0`, which hipcc's SLP vectoriser turns into v_pk_mul_f32 / v_pk_add_f32, took 3.6k cycles
// while the SIMD's other wave issued back-to-back MFMAs, 0.7k without the arithmetic (in-kernel stamps, tools/ts_conv.py).
// The probe: waves 4-7 of a block time a stream of 256 independent v_pk_mul_f32 (or 512 v_mul_f32 doing the same work)
// while waves 0-3 (one per SIMD) either idle or issue v_mfma_f32_32x32x16_bf16 back to back.
//   hipcc --offload-arch=gfx950 -O3 -o pk_f32_probe tools/probe/pk_f32_probe.hip && ./pk_f32_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int PACKED>
__global__ __launch_bounds__(512) void probe(float* sink, unsigned long long* cyc, int with_mfma) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __shared__ int done;
  if (threadIdx.x == 0) done = 0;
  __syncthreads();
  if (wave < 4) {
    if (!with_mfma) return;
    f32x16 acc = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
    // keep the matrix pipe busy until the timed waves are through
    while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4)
      for (int it = 0; it < 16; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (acc[0] == 12345.f) sink[0] = acc[1];
    return;
  }
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)(lane + i), (float)(lane - i)};
  const f32x2 m = {1.0001f, 0.9999f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int rep = 0; rep < 32; ++rep)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (PACKED) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(m));
      else {
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i][0]) : "v"(v[i][0]), "v"(m[0]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i][1]) : "v"(v[i][1]), "v"(m[1]));
      }
    }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) {
    if (wave == 4) cyc[blockIdx.x] = t1 - t0;
    __hip_atomic_fetch_add(&done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  if (s == 12345.f) sink[1] = s;
}

template <int PACKED>
static void run(float* d_sink, unsigned long long* d_cyc, int with_mfma) {
  const int blocks = 256;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe<PACKED>, dim3(blocks), dim3(512), 0, 0, d_sink, d_cyc, with_mfma);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double c = (double)h[blocks / 2];
  printf("%-28s %-34s %7.0f cycles for 512 multiplies = %5.1f per instruction\n", PACKED ? "256 x v_pk_mul_f32" : "512 x v_mul_f32",
         with_mfma ? "beside back-to-back MFMAs" : "alone on the SIMD", c, c / (PACKED ? 256.0 : 512.0));
}

int main() {
  float* d_sink; unsigned long long* d_cyc;
  if (hipMalloc(&d_sink, 64) != hipSuccess || hipMalloc(&d_cyc, 256 * 8) != hipSuccess) return 1;
  run<0>(d_sink, d_cyc, 0);
  run<1>(d_sink, d_cyc, 0);
  run<0>(d_sink, d_cyc, 1);
  run<1>(d_sink, d_cyc, 1);
  return 0;
}
