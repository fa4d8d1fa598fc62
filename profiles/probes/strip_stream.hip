// What does the memory system deliver for the scatter's access pattern, with nothing else in the kernel?
// x [183,400 x 128] fp32 (94 MB).  A: the strip pattern (grid 64 x 4 workgroups, each streams 128 B of every row of its row
// range: 16-byte loads, 8 lanes per row segment, 4 waves on 4 row quarters, DEPTH k-steps of 16 rows in flight per wave).
// B: the same bytes as whole rows (256 workgroups, each wave streams 512-B rows of its range).  C: flat contiguous read.
// Build: hipcc -O3 --offload-arch=gfx950 -o strip_stream strip_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(256) void k_strip(const float* __restrict__ x, float* __restrict__ out, int n_rows, int D) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wid, nq = gridDim.x * 4;
  const int ksteps = (n_rows + 15) / 16;
  const int k0 = (int)((long)q * ksteps / nq), k1 = (int)((long)(q + 1) * ksteps / nq);
  const float* base = x + (size_t)k0 * 16 * D + blockIdx.y * 32;
  const long rows = (long)(k1 - k0) * 16 < (long)n_rows - (long)k0 * 16 ? (long)(k1 - k0) * 16 : (long)n_rows - (long)k0 * 16;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0,
                                                                       rows > 0 ? (int)((rows * D - blockIdx.y * 32) * 4) : 0, 0x00020000);
  const unsigned rb = D * 4, vo0 = (lane >> 3) * rb + (lane & 7) * 16;
  f4 ring[DEPTH][2], acc = {0, 0, 0, 0};
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    ring[d][0] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo0 + d * 16 * rb, 0, 0));
    ring[d][1] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo0 + d * 16 * rb, 8 * rb, 0));
  }
  for (int k = 0; k < k1 - k0; k += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      acc += ring[d][0] + ring[d][1];
      ring[d][0] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo0 + (k + d + DEPTH) * 16 * rb, 0, 0));
      ring[d][1] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo0 + (k + d + DEPTH) * 16 * rb, 8 * rb, 0));
    }
  }
  out[(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int DEPTH>
__global__ __launch_bounds__(256) void k_rows(const float* __restrict__ x, float* __restrict__ out, int n_rows, int D) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wid, nq = gridDim.x * 4;
  const int steps = (n_rows + 3) / 4;                          // 4 whole rows (2 KB) per wave instruction pair
  const int s0 = (int)((long)q * steps / nq), s1 = (int)((long)(q + 1) * steps / nq);
  const float* base = x + (size_t)s0 * 4 * D;
  const long rows = (long)(s1 - s0) * 4 < (long)n_rows - (long)s0 * 4 ? (long)(s1 - s0) * 4 : (long)n_rows - (long)s0 * 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, rows > 0 ? (int)(rows * D * 4) : 0, 0x00020000);
  const unsigned vo0 = lane * 16;
  f4 ring[DEPTH][2], acc = {0, 0, 0, 0};
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    ring[d][0] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo0 + d * 2048, 0, 0));
    ring[d][1] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo0 + d * 2048, 1024, 0));
  }
  for (int k = 0; k < s1 - s0; k += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      acc += ring[d][0] + ring[d][1];
      ring[d][0] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo0 + (k + d + DEPTH) * 2048, 0, 0));
      ring[d][1] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo0 + (k + d + DEPTH) * 2048, 1024, 0));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <class F> float time_us(F f, int it) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) f();
  (void)hipEventRecord(a);
  for (int i = 0; i < it; ++i) f();
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / it;
}

int main() {
  const int n = 183400, D = 128;
  float *x, *out, *y;
  (void)hipMalloc(&x, (size_t)n * D * 4); (void)hipMalloc(&y, (size_t)512 << 20); (void)hipMalloc(&out, 1 << 20);
  (void)hipMemset(x, 1, (size_t)n * D * 4);
  const double mb = (double)n * D * 4 / 1e6;
  auto flush = [&] { (void)hipMemsetAsync(y, 0, (size_t)512 << 20, 0); };       // evict x from the Infinity Cache between runs
  for (int cold = 0; cold < 2; ++cold) {
    auto run = [&](const char* nm, auto kern, dim3 g) {
      float us;
      if (cold) {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        float tot = 0;
        for (int i = 0; i < 10; ++i) {
          flush(); (void)hipEventRecord(a); hipLaunchKernelGGL(kern, g, dim3(256), 0, 0, x, out, n, D); (void)hipEventRecord(b);
          (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); tot += ms;
        }
        us = tot * 1e3f / 10;
      } else {
        us = time_us([&] { hipLaunchKernelGGL(kern, g, dim3(256), 0, 0, x, out, n, D); }, 50);
      }
      printf("%-52s %s: %6.1f us  %5.2f TB/s\n", nm, cold ? "after a 512 MB memset" : "back to back          ", us, mb / us);
    };
    run("strip 64x4 workgroups, 2 k-steps in flight per wave", k_strip<2>, dim3(64, 4));
    run("strip 64x4 workgroups, 4 k-steps in flight per wave", k_strip<4>, dim3(64, 4));
    run("strip 64x4 workgroups, 8 k-steps in flight per wave", k_strip<8>, dim3(64, 4));
    run("strip 128x4 workgroups, 4 k-steps in flight", k_strip<4>, dim3(128, 4));
    run("strip 256x4 workgroups, 4 k-steps in flight", k_strip<4>, dim3(256, 4));
    run("whole rows, 256 workgroups, 4 x 2 KB in flight", k_rows<4>, dim3(256));
    run("whole rows, 1024 workgroups, 4 x 2 KB in flight", k_rows<4>, dim3(1024));
  }
  return 0;
}
