// Prints which XCD (XCC_ID hardware register) and CU each workgroup of a 1-D / 2-D grid lands on.
// build: hipcc --offload-arch=gfx950 -O2 -o xcc_map tools/probe/xcc_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const unsigned b = blockIdx.x + blockIdx.y * gridDim.x;
    out[2 * b] = xcc; out[2 * b + 1] = hwid;
  }
  // keep the block alive a little so that all 256 are resident together
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}
int main() {
  for (int cfg = 0; cfg < 2; ++cfg) {
    dim3 grid = cfg == 0 ? dim3(256, 1) : dim3(128, 2);
    const int nb = grid.x * grid.y;
    unsigned* d; hipMalloc(&d, nb * 8);
    hipLaunchKernelGGL(k, grid, dim3(512), 160 * 512, 0, d);
    std::vector<unsigned> h(2 * nb);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    printf("grid %dx%d: xcc of linear block 0..31:", grid.x, grid.y);
    for (int b = 0; b < 32; ++b) printf(" %u", h[2 * b] & 0xf);
    int ok = 0;
    for (int b = 0; b < nb; ++b) ok += ((h[2 * b] & 0xf) == (unsigned)(b % 8));
    printf("\n  blocks with xcc == linear %% 8: %d of %d\n", ok, nb);
    hipFree(d);
  }
  return 0;
}
