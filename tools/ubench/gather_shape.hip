// Micro-benchmark: what does the SHAPE of a row gather cost on the vector-memory path (TA / L1) of gfx950?
//
// The sparse convolution gathers 16 input rows x 128 bytes per wavefront and sub-stage.  k_spconv_h3 loads them in
// MFMA-fragment shape (lane l: row l & 15, 16-byte piece l >> 4): four consecutive lanes touch four DIFFERENT rows,
// i.e. every lane is its own L1 access.  The alternatives below let four (or eight) consecutive lanes read one
// contiguous 64 (128) byte run of ONE row, which then needs a lane transpose (through LDS) before the MFMA.
//
//   pattern 0  fragment shape                    lane l: row l & 15, bytes 16 (l >> 4) and + 64
//   pattern 1  quad-coalesced                    lane l: row l >> 2, bytes 32 (l & 3) and + 16
//   pattern 2  eight lanes per row               lane l: row (l >> 3) [+ 8], bytes 16 (l & 7)
//   pattern 3  pattern 1 straight into LDS       buffer_load_dwordx4 ... lds (no VGPR destination)
//   pattern 4  pattern 2 straight into LDS
//   pattern 5 / 6  contiguous 1 KiB blocks, the same sequence in every workgroup (weight-like), to VGPRs / to LDS
//
// Also checks what an out-of-window lane of `buffer_load ... lds` leaves in LDS (zeros are what the kernel needs).
// hipcc --offload-arch=gfx950 -O3 gather_shape.hip -o gather_shape && ./gather_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

constexpr unsigned kNoRow = 0x00FFFFFFu;

// occupancy_pct of the rows exist; the others read beyond the buffer window (zeros, no memory access)
template <int PAT>
__global__ void __launch_bounds__(256, 4)
k_gather(const float *feat, int n_rows, int row_bytes, int local_span, int occupancy_pct, int iters, float *out) {
  __shared__ float4 abuf[4][2][128];   // pattern 3: per wave, two 2 KiB sub-stage images
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const unsigned gw = blockIdx.x * 4 + wave;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(feat), (short)0, 0x7FFFF000, 0x00020000);
  const int chunks = row_bytes / 128;
  float acc = 0.f;
  const unsigned base_row = hash32(gw * 2654435761u) % (unsigned)n_rows;
  for (int it = 0; it < iters; ++it) {
    // which of the 16 rows of this step does the lane load?
    int sel0, sel1;
    unsigned byte0, byte1;
    if (PAT == 0)      { sel0 = sel1 = lane & 15; byte0 = 16u * (lane >> 4); byte1 = byte0 + 64u; }
    else if (PAT == 2 || PAT == 4) { sel0 = lane >> 3; sel1 = sel0 + 8; byte0 = byte1 = 16u * (lane & 7); }
    else               { sel0 = sel1 = lane >> 2; byte0 = 32u * (lane & 3); byte1 = byte0 + 16u; }
    unsigned r0, r1;
    {
      const unsigned h0 = hash32((gw * 1315423911u) ^ (unsigned)(it * 16 + sel0) * 2246822519u);
      const unsigned h1 = hash32((gw * 1315423911u) ^ (unsigned)(it * 16 + sel1) * 2246822519u);
      r0 = local_span ? (base_row + (h0 >> 8) % (unsigned)local_span) % (unsigned)n_rows : (h0 >> 8) % (unsigned)n_rows;
      r1 = local_span ? (base_row + (h1 >> 8) % (unsigned)local_span) % (unsigned)n_rows : (h1 >> 8) % (unsigned)n_rows;
      if ((int)(h0 & 127u) * 100 >= occupancy_pct * 128) r0 = kNoRow;
      if ((int)(h1 & 127u) * 100 >= occupancy_pct * 128) r1 = kNoRow;
    }
    for (int cc = 0; cc < chunks; ++cc) {
      unsigned v0 = __umul24(r0, (unsigned)row_bytes) + byte0, v1 = __umul24(r1, (unsigned)row_bytes) + byte1;
      if (PAT >= 5) {   // weight-like: every workgroup walks the same contiguous 1 KiB blocks
        v0 = (unsigned)((it * chunks + cc) * 2 % 2048) * 1024u + 16u * lane - (unsigned)cc * 128u;
        v1 = v0 + 1024u;
      }
      if (PAT == 3 || PAT == 4 || PAT == 6) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)&abuf[wave][0][0], 16, v0, (unsigned)cc * 128u, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)&abuf[wave][0][64], 16, v1, (unsigned)cc * 128u, 0, 0);
      } else {
        const float4 a = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, v0, (unsigned)cc * 128u, 0));
        const float4 b = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, v1, (unsigned)cc * 128u, 0));
        acc += a.x + b.y;
      }
    }
  }
  if (PAT == 3 || PAT == 4 || PAT == 6) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc = abuf[wave][0][lane].x + abuf[wave][0][64 + lane].y;
  }
  out[blockIdx.x * 256 + tid] = acc;
}

// one wave: lanes with an odd index read beyond the window; LDS pre-filled with 7.0
__global__ void k_oob(const float *feat, float *out) {
  __shared__ float4 buf[64];
  const int lane = threadIdx.x;
  buf[lane] = make_float4(7.f, 7.f, 7.f, 7.f);
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(feat), (short)0, 0x7FFFF000, 0x00020000);
  const unsigned v = (lane & 1) ? __umul24(kNoRow, 256u) + (16u * lane & 127u) : 16u * lane;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)&buf[0], 16, v, 0u, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float4 r = buf[lane];
  out[4 * lane + 0] = r.x; out[4 * lane + 1] = r.y; out[4 * lane + 2] = r.z; out[4 * lane + 3] = r.w;
}

template <int PAT>
static double run(const float *feat, int n_rows, int row_bytes, int span, int occ, int iters, float *out, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k_gather<PAT><<<blocks, 256>>>(feat, n_rows, row_bytes, span, occ, iters, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) k_gather<PAT><<<blocks, 256>>>(feat, n_rows, row_bytes, span, occ, iters, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / 5;
}

int main() {
  const int blocks = 2048, iters = 64;
  float *out;
  CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  {  // OOB check
    float *f; CK(hipMalloc(&f, 1 << 20));
    std::vector<float> h(1 << 18);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1.f + (float)(i % 97);
    CK(hipMemcpy(f, h.data(), 1 << 20, hipMemcpyHostToDevice));
    k_oob<<<1, 64>>>(f, out);
    CK(hipDeviceSynchronize());
    float r[256];
    CK(hipMemcpy(r, out, sizeof r, hipMemcpyDeviceToHost));
    int zeros = 0, kept = 0, good = 0;
    for (int l = 0; l < 64; ++l) {
      if (l & 1) { zeros += (r[4 * l] == 0.f && r[4 * l + 3] == 0.f); kept += (r[4 * l] == 7.f); }
      else good += (r[4 * l] == h[4 * l] && r[4 * l + 3] == h[4 * l + 3]);
    }
    printf("buffer_load ... lds, out-of-window lanes: %d of 32 wrote zeros, %d of 32 left LDS untouched; in-window lanes correct: %d of 32\n", zeros, kept, good);
    CK(hipFree(f));
  }
  struct Case { const char *name; int n_rows, row_bytes, span, occ; };
  const Case cases[] = {
      {"103k rows x 256 B (26 MB), random, 52 % occupied", 103396, 256, 0, 52},
      {"103k rows x 256 B, local span 4096, 52 %", 103396, 256, 4096, 52},
      {"103k rows x 256 B, local span 4096, 100 %", 103396, 256, 4096, 100},
      {"8k rows x 512 B (4 MB), random, 52 %", 8192, 512, 0, 52},
      {"2.2k rows x 1024 B (2.2 MB), random, 52 %", 2200, 1024, 0, 52},
  };
  for (const Case &c : cases) {
    float *f;
    const size_t bytes = (size_t)c.n_rows * c.row_bytes;
    CK(hipMalloc(&f, bytes + 4096));
    CK(hipMemset(f, 0, bytes + 4096));
    const double loads = (double)blocks * 4 * iters * (c.row_bytes / 128) * 2;   // wave-level load instructions
    const double t0 = run<0>(f, c.n_rows, c.row_bytes, c.span, c.occ, iters, out, blocks);
    const double t1 = run<1>(f, c.n_rows, c.row_bytes, c.span, c.occ, iters, out, blocks);
    const double t2 = run<2>(f, c.n_rows, c.row_bytes, c.span, c.occ, iters, out, blocks);
    const double t3 = run<3>(f, c.n_rows, c.row_bytes, c.span, c.occ, iters, out, blocks);
    const double t4 = run<4>(f, c.n_rows, c.row_bytes, c.span, c.occ, iters, out, blocks);
    const double t5 = run<5>(f, c.n_rows, c.row_bytes, c.span, c.occ, iters, out, blocks);
    const double t6 = run<6>(f, c.n_rows, c.row_bytes, c.span, c.occ, iters, out, blocks);
    printf("   8-lane->LDS %7.1f us (%.1f ns) | contiguous 1 KiB blocks (weight-like) %7.1f us (%.1f ns) | the same ->LDS %7.1f us (%.1f ns)\n",
           t4, t4 * 1e3 * 256 / loads, t5, t5 * 1e3 * 256 / loads, t6, t6 * 1e3 * 256 / loads);
    printf("%-52s fragment %7.1f us | quad %7.1f us | 8-lane %7.1f us | quad->LDS %7.1f us   (%.0f k wave-loads; ns per load per CU: %.1f / %.1f / %.1f / %.1f)\n",
           c.name, t0, t1, t2, t3, loads / 1e3, t0 * 1e3 * 256 / loads, t1 * 1e3 * 256 / loads, t2 * 1e3 * 256 / loads,
           t3 * 1e3 * 256 / loads);
    CK(hipFree(f));
  }
  return 0;
}
