// Does a VALU write to the first data register of a buffer_store_dwordx4, issued right after the store, reach memory?
// Found through the sliding-halo conv kernel's epilogue on gfx950: the compiler reuses v0..v3 of a 16-byte row store
// for the next M tile's LeakyReLU one instruction after the store, and rows of the output carried that NEXT value
// in lanes 12-15 of each 16-lane row, dword 0, while another wave of the SIMD was issuing MFMAs.
// The probe: waves 0-3 of a 512-thread block run MFMAs; waves 4-7 store a known pattern with `pad` independent
// s_nop cycles between the store and a VALU overwrite of its data registers.  Host counts wrong dwords per pad.
// mfma 2 = waves 0-3 stream loads instead (vector-memory back-pressure).
//   hipcc --offload-arch=gfx950 -O3 -o store_war_probe tools/probe/store_war_probe.hip && ./store_war_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int PAD, int SOFF>   // SOFF 0: buffer store, literal soffset; 1: SGPR soffset (LLVM's hazard recogniser pads form 0 only); 2: global_store
__global__ __launch_bounds__(512) void probe(uint32_t* __restrict__ out, float* __restrict__ sink, int iters, int with_mfma) {
  const size_t rows = (size_t)gridDim.x * 4 * iters;   // the buffer holds rows x 64 lanes x 16 bytes
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {
    if (!with_mfma) return;
    if (with_mfma == 2) {   // memory pressure instead: these waves keep the CU's vector-memory queue full
      const u32x4* src = reinterpret_cast<const u32x4*>(out);
      u32x4 t = {0u, 0u, 0u, 0u};
      for (int it = 0; it < iters * 16; ++it) t ^= __builtin_nontemporal_load(src + (((size_t)(blockIdx.x * 4 + wave) * iters * 16 + it) % rows) * 64 + lane);   // < rows * 64 16-byte records
      if ((t[0] ^ t[1] ^ t[2] ^ t[3]) == 0x12345678u) sink[1] = 1.f;
      return;
    }
    f32x16 acc = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
    for (int it = 0; it < iters * 24; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (acc[0] == 12345.f) sink[0] = acc[1];
    return;
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7FFFFFFF, 0x00020000);
  const uint32_t w = (uint32_t)(blockIdx.x * 4 + (wave - 4));
  for (int it = 0; it < iters; ++it) {
    const uint32_t sbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)((w * (uint32_t)iters + (uint32_t)it) * 1024u));   // bytes, wave-uniform
    const uint32_t base = (SOFF ? 0u : sbase) + (uint32_t)lane * 16u;
    const uint32_t tag = 0x10000000u | ((uint32_t)it << 8) | (uint32_t)lane;
    char* gptr = reinterpret_cast<char*>(out) + sbase + (uint32_t)lane * 16u;   // SOFF 2: global_store_dwordx4
    uint32_t v0 = tag, v1 = tag + 0x01000000u, v2 = tag + 0x02000000u, v3 = tag + 0x03000000u;
    // store, PAD cycles of s_nop, then overwrite all four data registers with a poison value by plain VALU moves
    asm volatile(
        "v_mov_b32 v0, %1\n\tv_mov_b32 v1, %2\n\tv_mov_b32 v2, %3\n\tv_mov_b32 v3, %4\n\t"
        "s_nop 4\n\t"
        ".if %8 == 2\n\tglobal_store_dwordx4 %9, v[0:3], off\n\t.elseif %8 == 1\n\tbuffer_store_dwordx4 v[0:3], %0, %5, %7 offen\n\t.else\n\tbuffer_store_dwordx4 v[0:3], %0, %5, 0 offen\n\t.endif\n\t"
        ".if %6 > 0\n\ts_nop %6 - 1\n\t.endif\n\t"
        "v_mov_b32 v0, 0xdeadbeef\n\tv_mov_b32 v1, 0xdeadbeef\n\tv_mov_b32 v2, 0xdeadbeef\n\tv_mov_b32 v3, 0xdeadbeef\n\t"
        :: "v"(base), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(rs), "n"(PAD), "s"(sbase), "n"(SOFF), "v"(gptr)
        : "v0", "v1", "v2", "v3", "memory");
  }
}

template <int PAD, int SOFF>
static long run(uint32_t* d_out, float* d_sink, uint32_t* h, int blocks, int iters, int with_mfma) {
  const size_t n = (size_t)blocks * 4 * iters * 64 * 4;
  hipMemset(d_out, 0, n * 4);
  hipLaunchKernelGGL((probe<PAD, SOFF>), dim3(blocks), dim3(512), 0, 0, d_out, d_sink, iters, with_mfma);
  hipDeviceSynchronize();
  hipMemcpy(h, d_out, n * 4, hipMemcpyDeviceToHost);
  long bad = 0, bad_lane[64] = {0}, bad_dw[4] = {0};
  for (size_t i = 0; i < n; i += 4) {
    const size_t rec = i / 4;
    const uint32_t lane = rec % 64, it = (rec / 64) % iters;
    const uint32_t tag = 0x10000000u | (it << 8) | lane;
    for (int k = 0; k < 4; ++k)
      if (h[i + k] != tag + 0x01000000u * k) { ++bad; ++bad_lane[lane]; ++bad_dw[k]; }
  }
  printf("pad %d  form %d  co-runner %d: %ld wrong dwords of %zu  (dword 0..3: %ld %ld %ld %ld)", PAD, SOFF, with_mfma, bad, n, bad_dw[0], bad_dw[1],
         bad_dw[2], bad_dw[3]);
  if (bad) {
    printf("  lanes:");
    for (int l = 0; l < 64; ++l) if (bad_lane[l]) printf(" %d", l);
  }
  printf("\n");
  return bad;
}

int main() {
  const int blocks = 256, iters = 64;
  const size_t n = (size_t)blocks * 4 * iters * 64 * 4;
  uint32_t* d_out; float* d_sink;
  if (hipMalloc(&d_out, n * 4) != hipSuccess || hipMalloc(&d_sink, 64) != hipSuccess) return 1;
  uint32_t* h = (uint32_t*)malloc(n * 4);
  for (int m = 0; m < 3; ++m) {
    run<0, 0>(d_out, d_sink, h, blocks, iters, m);
    run<1, 0>(d_out, d_sink, h, blocks, iters, m);
    run<2, 0>(d_out, d_sink, h, blocks, iters, m);
    run<0, 1>(d_out, d_sink, h, blocks, iters, m);
    run<1, 1>(d_out, d_sink, h, blocks, iters, m);
    run<2, 1>(d_out, d_sink, h, blocks, iters, m);
    run<0, 2>(d_out, d_sink, h, blocks, iters, m);
    run<1, 2>(d_out, d_sink, h, blocks, iters, m);
    run<2, 2>(d_out, d_sink, h, blocks, iters, m);
    run<3, 2>(d_out, d_sink, h, blocks, iters, m);
  }
  return 0;
}
