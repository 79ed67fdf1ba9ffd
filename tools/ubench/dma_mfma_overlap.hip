// Micro-benchmark: can LDS-DMA issue (buffer_load_dwordx4 ... lds) of one wavefront overlap the MFMAs of ANOTHER wavefront
// on the same SIMD?  An 8-wavefront workgroup per CU: wavefronts 0..3 (one per SIMD) are PRODUCERS -- per round they issue
// NP 1 KiB LDS-DMA pieces from an L2-resident buffer and wait for them; wavefronts 4..7 are CONSUMERS -- per round 48
// independent v_mfma_f32_16x16x32_f16 (what one sub-stage of the sparse convolution issues).  Three runs: producers only,
// consumers only, both.  If "both" ~ max(producers, consumers) the two pipes overlap across wavefronts and a
// producer / consumer split of the convolution can hide the DMA issue; if "both" ~ sum, it cannot.
// A fourth run has every wavefront do BOTH per round (the current kernels' shape) with 8 wavefronts per CU.
// build: hipcc --offload-arch=gfx950 -O3 dma_mfma_overlap.hip -o dma_mfma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

template <int MODE, int NP>   // MODE 1 producers only, 2 consumers only, 3 both (specialised), 4 every wavefront does both in turn
__global__ void __launch_bounds__(512, 2) k(float *out, const float *src, int rounds) {
  __shared__ float4 lds[8 * 1024];                               // 16 KiB per wavefront
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave < 4;
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(1.f + lane * 1e-3f); b[i] = (_Float16)(0.5f + i); }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), (short)0, 0x7FFFFFFF, 0x00020000);
  float4 *const mine = lds + wave * 1024;
  const unsigned base = ((unsigned)blockIdx.x * 8u + (unsigned)wave) * 16384u & 0x3FFFFFu;   // 4 MiB window: L2-resident
  for (int r = 0; r < rounds; ++r) {
    const bool do_p = MODE == 4 || (producer && (MODE & 1));
    const bool do_c = MODE == 4 || (!producer && (MODE & 2));
    if (do_p) {
#pragma unroll
      for (int j = 0; j < NP; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(mine + 64 * j), 16, (unsigned)lane * 16u + 1024u * j,
                                                 (base + (unsigned)(r & 15) * 262144u) & 0x3FFFFFu, 0, 0);
      if (MODE != 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (do_c) {
#pragma unroll
      for (int m = 0; m < 48; ++m) acc[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 15], 0, 0, 0);
    }
    if (MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float t = mine[lane].x;
  for (int i = 0; i < 16; ++i) t += acc[i][0];
  out[blockIdx.x * 512 + tid] = t;
}

template <int MODE, int NP>
void run(const char *name, float *d, float *src) {
  const int rounds = 4000, blocks = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE, NP><<<blocks, 512>>>(d, src, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE, NP><<<blocks, 512>>>(d, src, rounds);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s NP=%2d  %.3f ms  %.0f cycles per round @2.1GHz\n", name, NP, ms, ms * 1e-3 * 2.1e9 / rounds);
}

int main() {
  float *d, *src;
  (void)hipMalloc(&d, 256 * 512 * 4); (void)hipMalloc(&src, 8 << 20); (void)hipMemset(src, 0, 8 << 20);
  run<1, 16>("producers only (4 wavefronts x 16 pieces)", d, src);
  run<2, 16>("consumers only (4 wavefronts x 48 MFMAs)", d, src);
  run<3, 16>("both, specialised (4 producers + 4 consumers)", d, src);
  run<1, 32>("producers only (4 wavefronts x 32 pieces)", d, src);
  run<3, 32>("both, specialised (32 pieces per 48 MFMAs per SIMD)", d, src);
  run<4, 16>("every wavefront: 16 pieces then 48 MFMAs (8 wavefronts)", d, src);
  run<4, 8>("every wavefront: 8 pieces then 48 MFMAs (8 wavefronts)", d, src);
  return 0;
}
