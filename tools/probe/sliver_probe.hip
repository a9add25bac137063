// Calibration of rocprofv3's FETCH_SIZE for SLIVER reads on gfx950: the streamed conv kernel (conv_fwd4) fetches each
// 128-byte NDHWC channel row of a 64-channel tensor as four 32-byte slivers, one per K-chunk pass.  MI355X_MICROARCH.md
// calibrates FETCH_SIZE only for wide coalesced reads (reports 1/2: x2).  This probe reads, from a buffer far larger than
// L2 + Infinity Cache, (a) whole rows with 16 B per lane (the documented case), (b) ONE 32-byte sliver of every row,
// (c) all four slivers in four passes inside one launch -- each with a known byte count, so the counter's unit for the
// conv kernel's access shape can be read off.
//   hipcc --offload-arch=gfx950 -O3 -o sliver_probe tools/probe/sliver_probe.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o s -- ./sliver_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// mode 0: every 16-byte piece of every row; mode 1: slots 0,1 (first 32 B) of every row; mode 2: four passes, sliver p
__global__ void probe_kernel(const char* __restrict__ x, uint32_t* __restrict__ out, long rows, int mode) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x, gsz = (long)gridDim.x * blockDim.x;
  if (mode == 0) {
    for (long i = gid; i < rows * 8; i += gsz) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(x + i * 16);
      acc ^= v;
    }
  } else {
    const int passes = mode == 2 ? 4 : 1;
    for (int p = 0; p < passes; ++p)
      for (long i = gid; i < rows * 2; i += gsz) {      // two lanes per row: 32 contiguous bytes of a 128-byte row
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + (i >> 1) * 128 + p * 32 + (i & 1) * 16);
        acc ^= v;
      }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1u;   // keeps the loads alive
}

int main() {
  const long rows = 1L << 25;            // 32 Mi rows x 128 B = 4 GiB
  char* x;
  uint32_t* out;
  if (hipMalloc(&x, rows * 128) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  hipMemset(x, 1, rows * 128);
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      hipLaunchKernelGGL(probe_kernel, dim3(256 * 8), dim3(256), 0, 0, x, out, rows, mode);
      hipDeviceSynchronize();
    }
  printf("rows %ld: mode0 reads %ld B, mode1 %ld B (of %ld B of lines), mode2 %ld B\n", rows, rows * 128, rows * 32, rows * 128,
         rows * 128);
  return 0;
}
