// Weight gradient of the sparse convolution (last item: the training backward of lib/trainer.py:495-569).
// The input gradient needs no kernel of its own: it is imf_spconv_fwd over the opposite kernel map with the transposed
// weights (imfnet_amd/autograd.py).  The weight gradient is, per kernel offset k, the [cin x cout] outer-product sum
//     dW[k] = sum over pairs (i, o) of offset k of  in[i]^T . grad_out[o]
// -- a reduction over rows.  One workgroup owns a (k, 32 x 32 block of dW, chunk of 4096 rulebook slots) cell: it stages
// 64 gathered input rows and 64 gradient rows in LDS per step and accumulates its block on the fp32 vector pipe; the chunk
// partials are then summed in ascending order by k_wgrad_reduce (deterministic, no float atomics).
#include "common.h"

namespace imf {

constexpr int kWgTile = 32;          // dW block edge
constexpr int kWgRows = 64;          // rows staged per step
constexpr int kWgChunk = 4096;       // slots per workgroup

__global__ void __launch_bounds__(256)
k_wgrad_partial(const float *__restrict__ in, int cin, const float *__restrict__ grad, int cout,
                const int32_t *__restrict__ tile_rows, const int32_t *__restrict__ nbr, long long n_slots, long long n_out,
                int kvol, float *__restrict__ partial /* [chunks][kvol][cin][cout] */) {
  __shared__ float A[kWgRows][kWgTile + 1];
  __shared__ float G[kWgRows][kWgTile + 1];
  const int k = blockIdx.x;
  const int nci = cin / kWgTile > 0 ? (cin + kWgTile - 1) / kWgTile : 1;
  const int cib = blockIdx.y % nci, cob = blockIdx.y / nci;
  const long long chunk = blockIdx.z;
  const int tid = threadIdx.x;
  const int ta = tid >> 4, tb = tid & 15;                 // thread owns dW rows 2ta, 2ta+1 and columns 2tb, 2tb+1 of the block
  float acc00 = 0.f, acc01 = 0.f, acc10 = 0.f, acc11 = 0.f;
  const long long s_begin = chunk * kWgChunk, s_end = s_begin + kWgChunk < n_slots ? s_begin + kWgChunk : n_slots;
  for (long long s0 = s_begin; s0 < s_end; s0 += kWgRows) {
    // stage: 64 rows x 32 channels of the gathered input and of the output gradient (8 floats per thread each)
    for (int e = tid; e < kWgRows * kWgTile; e += 256) {
      const int r = e >> 5, c = e & 31;
      const long long slot = s0 + r;
      float a = 0.f, g = 0.f;
      if (slot < s_end) {
        const int orow = tile_rows ? tile_rows[slot] : (slot < n_out ? (int)slot : -1);
        const int irow = nbr ? nbr[(long long)k * n_slots + slot] : orow;
        if (orow >= 0 && irow >= 0) {
          const int ci = cib * kWgTile + c, co = cob * kWgTile + c;
          if (ci < cin) a = in[(long long)irow * cin + ci];
          if (co < cout) g = grad[(long long)orow * cout + co];
        }
      }
      A[r][c] = a;
      G[r][c] = g;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < kWgRows; ++r) {
      const float a0 = A[r][2 * ta], a1 = A[r][2 * ta + 1], g0 = G[r][2 * tb], g1 = G[r][2 * tb + 1];
      acc00 += a0 * g0; acc01 += a0 * g1; acc10 += a1 * g0; acc11 += a1 * g1;
    }
    __syncthreads();
  }
  float *dst = partial + ((long long)chunk * kvol + k) * cin * cout;
  const int ci0 = cib * kWgTile + 2 * ta, co0 = cob * kWgTile + 2 * tb;
  if (ci0 < cin && co0 < cout) dst[(long long)ci0 * cout + co0] = acc00;
  if (ci0 < cin && co0 + 1 < cout) dst[(long long)ci0 * cout + co0 + 1] = acc01;
  if (ci0 + 1 < cin && co0 < cout) dst[(long long)(ci0 + 1) * cout + co0] = acc10;
  if (ci0 + 1 < cin && co0 + 1 < cout) dst[(long long)(ci0 + 1) * cout + co0 + 1] = acc11;
}

__global__ void __launch_bounds__(256)
k_wgrad_reduce(const float *__restrict__ partial, long long n_elems, int chunks, float *__restrict__ dw) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_elems) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += partial[(long long)c * n_elems + i];
  dw[i] = s;
}

}  // namespace imf

using namespace imf;

extern "C" {

size_t imf_spconv_wgrad_workspace_bytes(int64_t n_slots, int kvol, int cin, int cout) {
  const int64_t chunks = div_up(n_slots, kWgChunk);
  return (size_t)chunks * kvol * cin * cout * sizeof(float);
}

int imf_spconv_wgrad(const float *in, int cin, const float *grad_out, int cout, const int32_t *tile_rows, const int32_t *nbr,
                     int64_t n_slots, int64_t n_out, int kvol, float *dw, void *workspace, size_t workspace_bytes,
                     void *stream) {
  IMF_REQUIRE(in && grad_out && dw && workspace, "imf_spconv_wgrad: null pointer");
  IMF_REQUIRE(cin > 0 && cout > 0 && kvol >= 1 && kvol <= IMF_MAX_KVOL, "imf_spconv_wgrad: cin=%d cout=%d kvol=%d", cin, cout, kvol);
  IMF_REQUIRE(nbr || kvol == 1, "imf_spconv_wgrad: nbr may be NULL only when kvol == 1");
  IMF_REQUIRE(n_slots > 0 && n_out > 0, "imf_spconv_wgrad: n_slots / n_out");
  IMF_REQUIRE(workspace_bytes >= imf_spconv_wgrad_workspace_bytes(n_slots, kvol, cin, cout), "imf_spconv_wgrad: workspace too small");
  const int64_t chunks = div_up(n_slots, kWgChunk);
  IMF_REQUIRE(chunks <= 65535, "imf_spconv_wgrad: too many rows");
  const int nci = (cin + kWgTile - 1) / kWgTile, nco = (cout + kWgTile - 1) / kWgTile;
  hipStream_t st = (hipStream_t)stream;
  k_wgrad_partial<<<dim3((unsigned)kvol, (unsigned)(nci * nco), (unsigned)chunks), 256, 0, st>>>(
      in, cin, grad_out, cout, tile_rows, nbr, n_slots, n_out, kvol, (float *)workspace);
  const long long n_elems = (long long)kvol * cin * cout;
  k_wgrad_reduce<<<(unsigned)div_up(n_elems, 256), 256, 0, st>>>((const float *)workspace, n_elems, (int)chunks, dw);
  IMF_CHECK_LAUNCH("k_wgrad");
  return IMF_OK;
}

}  // extern "C"
