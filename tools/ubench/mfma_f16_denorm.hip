// Probe: does v_mfma_f32_16x16x32_f16 keep f16 subnormal inputs, and what is its issue interval?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void probe(float *out) {
  const int lane = threadIdx.x;
  h8 a, b;
  for (int t = 0; t < 8; ++t) {
    a[t] = (_Float16)0.0f;
    b[t] = (_Float16)0.0f;
  }
  // A[m][k]: only k = 0 (lanes 0..15, t = 0) non-zero: subnormal 2^-20 ; B[k=0][n] = 1024
  if (lane < 16) {
    a[0] = __builtin_bit_cast(_Float16, (unsigned short)0x0010);   // 16 * 2^-24 = 2^-20 (subnormal)
    b[0] = (_Float16)1024.0f;
  }
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  out[lane] = c[0];
  // conversion of a subnormal-range f32 to f16 and back
  float x = 3.0e-6f * (lane + 1);
  _Float16 hx = (_Float16)x;
  out[64 + lane] = (float)hx;
}

__global__ void rate(float *out, int iters, long long *cyc) {
  h8 a, b;
  for (int t = 0; t < 8; ++t) { a[t] = (_Float16)(threadIdx.x * 0.001f + t); b[t] = (_Float16)(t * 0.5f); }
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float *d; long long *dc;
  hipMalloc(&d, 4096); hipMalloc(&dc, 8);
  probe<<<1, 64>>>(d);
  float h[128];
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("mfma subnormal A (2^-20) x 1024 = %g (expect %g if kept, 0 if flushed)\n", h[0], 1024.0 / 1048576.0);
  printf("f32->f16->f32 of 3e-6: %g ; of 6e-6: %g (0 if flushed)\n", h[64], h[65]);
  for (int w = 1; w <= 2; ++w) {
    rate<<<1, 64 * w>>>(d, 10000, dc);
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("waves/SIMD-ish %d: %.2f s_memtime ticks per MFMA (4 independent accumulators)\n", w, (double)c / 40000.0);
  }
  return 0;
}
