// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 under different accumulator dependency
// patterns and waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_f32.hip -o mfma_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a0, float b0) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
    if (PATTERN == 0) {          // round-robin over 4 accumulators (dependency distance 4)
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    } else if (PATTERN == 1) {   // 4 back-to-back on the same accumulator, then the next
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    } else if (PATTERN == 2) {   // single accumulator chain
#pragma unroll
      for (int r = 0; r < 64; ++r) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
    } else {                     // distance 2
#pragma unroll
      for (int r = 0; r < 32; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int P>
void run(const char *name, int blocks, float *d) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<P><<<blocks, 256>>>(d, 10, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<P><<<blocks, 256>>>(d, iters, 1.f, 1.f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // each block = 4 waves (1 per SIMD); blocks/256 CUs = waves per SIMD
  double mfma_per_simd = (double)blocks / 256.0 * iters * 64;
  double cyc = ms * 1e-3 * 2.4e9 / mfma_per_simd;
  double tf = (double)blocks * 4 * iters * 64 * 2048.0 / (ms * 1e-3) / 1e12;
  printf("%-28s blocks=%4d (%.0f waves/SIMD)  %.3f ms  %.1f cyc/MFMA/SIMD @2.4GHz  %.1f TF\n", name, blocks,
         blocks / 256.0, ms, cyc, tf);
}

int main() {
  float *d; hipMalloc(&d, 2048 * 256 * 4);
  for (int blocks : {256, 512, 1024}) {
    run<0>("rr over 4 accumulators", blocks, d);
    run<1>("4 back-to-back per acc", blocks, d);
    run<3>("rr over 2 accumulators", blocks, d);
    run<2>("single chain", blocks, d);
  }
  return 0;
}
