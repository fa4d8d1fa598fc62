// Role-specialised waves: does the VALU work of a SECOND wave on the same SIMD hide behind the first wave's matrix
// instructions?  512 threads = 8 waves = 2 per SIMD, one workgroup per CU.  Waves 0..3: 36 v_mfma_f32_32x32x16_bf16 per
// iteration (9 accumulators), optionally one INDEPENDENT ds_read_b128 behind every MFMA (consumed 36 MFMAs later) and two
// VALU ops.  Waves 4..7: NV split steps (cvt, shl, sub) per iteration (+ optionally 3 ds_write_b128), i.e. the staging
// work of one k-step.  BAR: one s_barrier per iteration (the hand-off of a k-step).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int LDSR, int NV, int BAR, int MFMA_ON>
__global__ __launch_bounds__(512) void k(float* out, int n_it, unsigned long long* clk, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = seed * i;
  __syncthreads();
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float s = 0.f;
  unsigned long long c0 = clock64(), c1 = c0;
  if (wid < 4) {
    f16v acc[9];
    for (int a = 0; a < 9; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    bf8 A, B;
    for (int j = 0; j < 8; ++j) { A[j] = (__bf16)(seed * (lane * 8 + j) * 0.37f); B[j] = (__bf16)(seed * (lane * 3 + j) * 0.11f + 0.5f); }
    f4 fr[12];
    for (int i = 0; i < 12; ++i) fr[i] = f4{0.f, 0.f, 0.f, 0.f};
    f4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < n_it; ++it) {
#pragma unroll
      for (int m = 0; m < 36; ++m) {
        if (MFMA_ON) acc[m % 9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[m % 9], 0, 0, 0);
        if (LDSR && (m % 3) == 0) {               // 12 independent reads per iteration (9 LUT + 3 fragments of a k-step)
          sum += fr[m / 3];                       // consume what was read one iteration ago
          fr[m / 3] = *reinterpret_cast<const f4*>(&lds[(wid * 4096 + lane * 4 + (m / 3) * 256 + (it & 3) * 64) & 16380]);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (LDSR && (m % 3) == 0) { __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
      }
      if (BAR) __syncthreads();
    }
    c1 = clock64();
    s = sum[0] + sum[1] + sum[2] + sum[3];
    for (int a = 0; a < 9; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
  } else {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed * (lane + i);
    for (int it = 0; it < n_it; ++it) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        float& x = v[q % 16];
        const __bf16 a = (__bf16)x;
        x = x - (float)a + 1.0f;
      }
      if (NV) {
        *reinterpret_cast<f4*>(&lds[((wid - 4) * 4096 + 2048 + lane * 4 + (it & 7) * 256) & 16380]) = f4{v[0], v[1], v[2], v[3]};
      }
      if (BAR) __syncthreads();
    }
    c1 = clock64();
    for (int i = 0; i < 16; ++i) s += v[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
  if (threadIdx.x == 256 && blockIdx.x == 0) clk[1] = c1 - c0;
}

template <int LDSR, int NV, int BAR, int MFMA_ON>
void run(float* out, unsigned long long* clk, const char* what) {
  const int n_it = 300;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  unsigned long long h[2] = {0, 0};
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<LDSR, NV, BAR, MFMA_ON>), dim3(256), dim3(512), 0, 0, out, n_it, clk, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  }
  printf("%-46s: %7.1f us  MFMA wave %6.1f cycles per MFMA   VALU wave %7.1f cycles per iteration\n", what, ms * 1e3,
         (double)h[0] / (n_it * 36.0), (double)h[1] / n_it);
}

int main() {
  float* out; unsigned long long* clk;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&clk, 16);
  run<0, 0, 0, 1>(out, clk, "MFMA waves alone (VALU waves idle)");
  run<0, 40, 0, 0>(out, clk, "VALU waves alone, 40 split steps");
  run<0, 40, 0, 1>(out, clk, "MFMA + 40 split steps per 36 MFMA, no barrier");
  run<0, 80, 0, 1>(out, clk, "MFMA + 80 split steps, no barrier");
  run<0, 160, 0, 1>(out, clk, "MFMA + 160 split steps, no barrier");
  run<0, 40, 1, 1>(out, clk, "MFMA + 40 split steps, barrier per iteration");
  run<1, 0, 0, 1>(out, clk, "MFMA + 12 independent ds_read_b128, no VALU wave");
  run<1, 40, 0, 1>(out, clk, "MFMA + 12 ds_read_b128 + 40 split steps");
  run<1, 40, 1, 1>(out, clk, "MFMA + 12 ds_read_b128 + 40 split + barrier");
  run<1, 80, 1, 1>(out, clk, "MFMA + 12 ds_read_b128 + 80 split + barrier");
  return 0;
}
