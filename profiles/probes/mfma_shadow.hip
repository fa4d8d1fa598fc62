// How much other work hides behind a matrix instruction of a wave that is ALONE on its SIMD?
// N_IT x 36 v_mfma_f32_32x32x16_bf16 per wave (9 accumulators in rotation, as the strip scatter), 4 waves per workgroup,
// one workgroup per CU, with NV instructions of one kind dealt out behind every MFMA (sched_group_barrier):
//   kind 0: v_fma_f32          kind 1: v_cvt_pk_bf16_f32 (+ the shift / subtract of the 3-way split)
//   kind 2: ds_read_b128 (one per MFMA, NV ignored)      kind 3: buffer/global dword loads (one per MFMA)
// Prints cycles per MFMA as the wave sees them (clock64) -- 32 = everything hid.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int KIND, int NV>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int n_it, unsigned long long* clk, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed * i;
  __syncthreads();
  f16v acc[9];
  for (int a = 0; a < 9; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  bf8 A, B;
  for (int j = 0; j < 8; ++j) { A[j] = (__bf16)(seed * (threadIdx.x * 8 + j) * 0.37f); B[j] = (__bf16)(seed * (threadIdx.x * 3 + j) * 0.11f + 0.5f); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed * (threadIdx.x + i);
  f4 l = {0.f, 0.f, 0.f, 0.f};
  float g = 0.f;
  const float* gp = in + threadIdx.x + blockIdx.x * 256;
  unsigned long long c0 = clock64();
  for (int it = 0; it < n_it; ++it) {
#pragma unroll
    for (int m = 0; m < 36; ++m) {
      acc[m % 9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[m % 9], 0, 0, 0);
      if (KIND == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[(m * NV + q) % 16] = fmaf(v[(m * NV + q) % 16], 1.0001f, 0.5f);
      } else if (KIND == 1) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {          // one step of the split: cvt, widen, subtract
          float& x = v[(m * NV + q) % 16];
          const __bf16 a = (__bf16)x;
          x = x - (float)a + 1.0f;
        }
      } else if (KIND == 2) {
        l += *reinterpret_cast<const f4*>(&lds[((threadIdx.x * 4 + m * 64 + it * 16) & 4092)]);
      } else if (KIND == 3) {
        g += gp[(size_t)((m + it * 36) & 1023) * 65536];
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (KIND == 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      if (KIND == 1) __builtin_amdgcn_sched_group_barrier(0x002, 3 * NV, 0);
      if (KIND == 2) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); }
      if (KIND == 3) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
    }
  }
  unsigned long long c1 = clock64();
  float s = l[0] + l[1] + l[2] + l[3] + g;
  for (int a = 0; a < 9; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

template <int KIND, int NV>
void run(float* out, const float* in, unsigned long long* clk, const char* what) {
  const int n_it = 300;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  unsigned long long h = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NV>), dim3(256), dim3(256), 0, 0, out, in, n_it, clk, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  }
  printf("%-28s NV %2d: %7.1f us, %6.1f cycles per MFMA\n", what, NV, ms * 1e3, (double)h / (n_it * 36.0));
}

int main() {
  float *out, *in; unsigned long long* clk;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&in, (size_t)1024 * 65536 * 4 + 65536 * 4); hipMalloc(&clk, 16);
  hipMemset(in, 0, (size_t)1024 * 65536 * 4 + 65536 * 4);
  run<0, 0>(out, in, clk, "MFMA only");
  run<0, 2>(out, in, clk, "v_fma_f32"); run<0, 4>(out, in, clk, "v_fma_f32"); run<0, 6>(out, in, clk, "v_fma_f32");
  run<0, 8>(out, in, clk, "v_fma_f32"); run<0, 12>(out, in, clk, "v_fma_f32");
  run<1, 1>(out, in, clk, "split step (cvt,shl,sub)"); run<1, 2>(out, in, clk, "split step (cvt,shl,sub)");
  run<1, 3>(out, in, clk, "split step (cvt,shl,sub)"); run<1, 4>(out, in, clk, "split step (cvt,shl,sub)");
  run<2, 0>(out, in, clk, "ds_read_b128 + 4 v_add");
  run<3, 0>(out, in, clk, "global dword load + add");
  return 0;
}
