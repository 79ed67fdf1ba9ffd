// Micro-benchmark: 64 fp32 MFMAs per "stage" followed by a workgroup barrier (4 waves = 4 SIMDs),
// optionally with a 16 KiB LDS write phase, at 1..4 workgroups per CU -- isolates what the per-stage
// barrier of the LDS-staged sparse-conv kernel costs.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int PER_STAGE>   // MODE 0: no barrier; 1: barrier; 2: barrier + LDS write + LDS B reads
__global__ void __launch_bounds__(256) k(float *out, const float4 *src, int stages) {
  __shared__ float4 lds[2][1024];
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int tid = threadIdx.x, lane = tid & 63;
  float a = 1.f + tid * 1e-3f;
  float4 w = src[tid];
  for (int s = 0; s < stages; ++s) {
    if (MODE == 2) {
      for (int q = 0; q < 4; ++q) lds[s & 1][q * 256 + tid] = w;
    }
    if (MODE >= 1) __syncthreads();
#pragma unroll
    for (int r = 0; r < PER_STAGE / 16; ++r) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float4 b = MODE == 2 ? lds[s & 1][(r * 4 + c) % 16 * 64 + lane] : w;
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.x, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.y, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.z, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.w, acc[c], 0, 0, 0);
      }
    }
  }
  float t = 0;
  for (int i = 0; i < 4; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + tid] = t;
}

template <int MODE, int PS>
void run(const char *name, int blocks, float *d, float4 *src) {
  const int stages = 2000 * 64 / PS;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE, PS><<<blocks, 256>>>(d, src, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE, PS><<<blocks, 256>>>(d, src, stages);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double mfma_per_simd = (double)blocks / 256.0 * stages * PS;
  printf("%-34s PS=%3d blocks/CU=%d  %.3f ms  %.1f cyc/MFMA/SIMD @2.4GHz\n", name, PS, blocks / 256, ms,
         ms * 1e-3 * 2.4e9 / mfma_per_simd);
}

int main() {
  float *d; float4 *src;
  (void)hipMalloc(&d, 1024 * 256 * 4); (void)hipMalloc(&src, 256 * 16); (void)hipMemset(src, 0, 256 * 16);
  for (int blocks : {256, 512, 1024}) {
    run<0, 64>("no barrier", blocks, d, src);
    run<1, 64>("barrier per stage", blocks, d, src);
    run<2, 64>("barrier + 16K LDS write + B reads", blocks, d, src);
    run<1, 128>("barrier per stage", blocks, d, src);
    run<2, 128>("barrier + 16K LDS write + B reads", blocks, d, src);
    run<1, 256>("barrier per stage", blocks, d, src);
  }
  return 0;
}
