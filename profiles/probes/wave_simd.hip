// Which SIMD does wave i of a workgroup land on?  (HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13])
// Build: hipcc -O2 --offload-arch=gfx950 wave_simd.hip -o wave_simd ; run: ./wave_simd
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = hw;
}
int main() {
  for (int nt : {256, 512, 768}) {
    unsigned* d; hipMalloc(&d, 4096 * 4);
    hipLaunchKernelGGL(k, dim3(4), dim3(nt), 0, 0, d);
    unsigned h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("threads %d:\n", nt);
    for (int b = 0; b < 4; ++b) {
      printf("  wg %d:", b);
      for (int w = 0; w < nt / 64; ++w) { unsigned v = h[b * (nt / 64) + w]; printf(" w%d->simd%u(cu%u,slot%u)", w, (v >> 4) & 3, (v >> 8) & 15, v & 15); }
      printf("\n");
    }
    hipFree(d);
  }
  return 0;
}
