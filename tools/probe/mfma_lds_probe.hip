// What bounds the MFMA phase of the sliding-halo kernels?  One wave per SIMD (waves 4-7 of the block idle at a
// barrier) runs the kernels' unrolled step sequence -- per 6 MFMAs (v_mfma_f32_32x32x16_bf16, two accumulators)
// 7 ds_read_b128 issued PF steps ahead and 4 counted s_waitcnt -- in variants that drop one ingredient at a time.
// Prints cycles per MFMA (s_memtime over the whole sequence, median block).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lds_probe tools/probe/mfma_lds_probe.hip && ./mfma_lds_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: reads + counted waits + MFMAs (the kernel's loop); 1: MFMAs only; 2: reads + MFMAs, one wait at the end;
// 3: reads + waits, no MFMAs; 4: as 0 but every wait is lgkmcnt(0)
template <int MODE, int RING>
struct Seq {
  static constexpr int NS = 72, PF = RING - 2, PB = 13312;
  static constexpr int nloads(int st) { return st >= NS ? 0 : ((st & 3) < 3 ? 2 : 1); }
  static constexpr int younger(int st) { int n = 0; for (int t = st + 1; t <= st + PF; ++t) n += nloads(t); return n; }
  template <int ST>
  static __device__ __forceinline__ void load(u32x4 (&wfr)[RING], u32x4 (&xfr)[RING], int xa, int wa) {
    if constexpr (MODE != 1) {
      constexpr int SL = ST % RING, pl = ST & 3, grp_ = ST >> 2;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xfr[SL]) : "v"(xa), "n"(pl * PB + (grp_ % 9) * 64));
      if constexpr (pl < 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wfr[SL]) : "v"(wa), "n"((pl * 18 + grp_) << 10));
    }
  }
  template <int ST>
  static __device__ __forceinline__ void step(f32x16 (&acc)[2], u32x4 (&wfr)[RING], u32x4 (&xfr)[RING], int xa, int wa) {
    if constexpr (ST < NS) {
      if constexpr (ST + PF < NS) load<ST + PF>(wfr, xfr, xa, wa);
      if constexpr (MODE == 0 || MODE == 3) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(younger(ST)));
      if constexpr (MODE == 4) asm volatile("s_waitcnt lgkmcnt(0)");
      __builtin_amdgcn_sched_barrier(0);
      constexpr int pl = ST & 3;
      if constexpr (MODE != 3) {
        if constexpr (pl <= 2)
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfr[ST % RING]), __builtin_bit_cast(bf16x8, xfr[ST % RING]), acc[0], 0, 0, 0);
        if constexpr (pl >= 1)
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfr[(ST - 1) % RING]), __builtin_bit_cast(bf16x8, xfr[ST % RING]), acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      step<ST + 1>(acc, wfr, xfr, xa, wa);
    }
  }
  template <int ST>
  static __device__ __forceinline__ void prologue(u32x4 (&wfr)[RING], u32x4 (&xfr)[RING], int xa, int wa) {
    if constexpr (ST < PF) { load<ST>(wfr, xfr, xa, wa); prologue<ST + 1>(wfr, xfr, xa, wa); }
  }
  static __device__ __forceinline__ void run(f32x16 (&acc)[2], int xa, int wa) {
    u32x4 wfr[RING], xfr[RING];
    if constexpr (MODE == 1) {
      for (int i = 0; i < RING; ++i) { wfr[i] = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}; xfr[i] = wfr[i]; }
    }
    prologue<0>(wfr, xfr, xa, wa);
    step<0>(acc, wfr, xfr, xa, wa);
    if constexpr (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)");
  }
};

template <int MODE, int RING>
__global__ __launch_bounds__(512) void probe(float* sink, unsigned long long* cyc, int reps, int rnd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 512) {   // random-looking bf16 pairs of magnitude ~1 (operand toggling as in a real layer)
    uint32_t h = (uint32_t)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    reinterpret_cast<uint32_t*>(smem)[i] = rnd ? ((h & 0x807F807Fu) | 0x3F003F00u) : 0x3C003C00u;
  }
  __syncthreads();
  if (wave >= 4) return;
  // X: 32 consecutive 64-byte rows per half-wave, 16-byte slots swizzled as in the kernel; W: lane * 16 in a 1-KiB fragment
  const int r = lane & 31, hh = lane >> 5;
  const int row = wave * 34 + r;
  const int xa = row * 64 + (((hh) ^ ((row >> 2) & 3)) << 4);
  const int wa = 106496 + lane * 16;
  f32x16 acc[2] = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < reps; ++it) Seq<MODE, RING>::run(acc, xa, wa);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0 && wave == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc[0][0] + acc[1][0] == 12345.f) sink[0] = acc[0][1];
}

template <int MODE, int RING>
static void run(const char* what, float* d_sink, unsigned long long* d_cyc, int reps = 64, int rnd = 0) {
  const int blocks = 256;
  auto k = probe<MODE, RING>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 160 * 1024, 0, d_sink, d_cyc, reps, rnd);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double per_seq = (double)h[blocks / 2] / reps;
  printf("%-58s ring %d reps %5d %s: %7.0f cycles per 108-MFMA sequence = %5.1f per MFMA slot\n", what, RING, reps, rnd ? "random" : "const ", per_seq, per_seq / 108.0);
}

int main() {
  float* d_sink; unsigned long long* d_cyc;
  if (hipMalloc(&d_sink, 64) != hipSuccess || hipMalloc(&d_cyc, 256 * 8) != hipSuccess) return 1;
  run<1, 6>("MFMAs only", d_sink, d_cyc);
  run<0, 6>("reads + counted waits + MFMAs (the kernel's loop)", d_sink, d_cyc);
  run<0, 8>("reads + counted waits + MFMAs (the kernel's loop)", d_sink, d_cyc);
  run<2, 6>("reads + MFMAs, no waits inside", d_sink, d_cyc);
  run<4, 6>("reads + lgkmcnt(0) waits + MFMAs", d_sink, d_cyc);
  run<3, 6>("reads + counted waits, no MFMAs", d_sink, d_cyc);
  run<0, 6>("the kernel's loop, long run", d_sink, d_cyc, 4000, 0);
  run<0, 6>("the kernel's loop, long run", d_sink, d_cyc, 4000, 1);
  run<1, 6>("MFMAs only, long run", d_sink, d_cyc, 4000, 1);
  return 0;
}
