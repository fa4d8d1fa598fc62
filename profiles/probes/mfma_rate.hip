// MFMA issue-rate probe: N_IT x 48 v_mfma_f32_32x32x16_bf16 per wave, WAVES waves per workgroup, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int n_it, unsigned long long* clk, float seed) {
  f16v acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  bf8 A, B;
  for (int j = 0; j < 8; ++j) { A[j] = (__bf16)(seed * (threadIdx.x * 8 + j) * 0.37f); B[j] = (__bf16)(seed * (threadIdx.x * 3 + j) * 0.11f + 0.5f); }
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < n_it; ++it) {
#pragma unroll
    for (int m = 0; m < 48; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[m % NACC], 0, 0, 0);
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 256 * 512 * 4 * 4); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Cfg { int nacc; float seed; int waves; int n_it; };
  std::vector<Cfg> cfgs;
  for (float seed : {0.f, 1.f}) for (int waves : {4, 8}) for (int n_it : {25, 250, 2500}) cfgs.push_back({4, seed, waves, n_it});
  for (int nacc : {1, 2, 3}) cfgs.push_back({nacc, 1.f, 4, 250});      // dependent accumulator chains
  for (const Cfg& c : cfgs) {
    const int nacc = c.nacc, waves = c.waves, n_it = c.n_it; const float seed = c.seed;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (nacc == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, out, n_it, clk, seed);
      else if (nacc == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 0, 0, out, n_it, clk, seed);
      else if (nacc == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(64 * waves), 0, 0, out, n_it, clk, seed);
      else hipLaunchKernelGGL(k<4>, dim3(256), dim3(64 * waves), 0, 0, out, n_it, clk, seed);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      double flop = 256.0 * waves * n_it * 48 * 32768.0;
      if (rep == 2) printf("nacc %d seed %.0f waves/CU %d n_it %5d: %.1f us, %.2f PFLOP/s, shader clk %.2f GHz (clock64/wall100MHz), cycles per MFMA per SIMD %.1f\n",
             nacc, seed, waves, n_it, ms * 1e3, flop / (ms * 1e-3) / 1e15, (double)h[0] / ((double)h[1] / 100e6) / 1e9,
             (double)h[0] / (n_it * 48.0 * (waves / 4)));
    }
  }
  return 0;
}
