// Pieces shared by the sparse-convolution kernels (spconv.hip, spconv_h3.hip).
#pragma once
#include "common.h"

namespace imf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvParams {
  const float *in_a, *in_b;
  int c_a, c_b;
  const float *w_packed;
  int kvol, cout;
  const int32_t *tile_rows, *nbr;
  const uint32_t *tile_mask;
  long long n_slots, n_out;
  const float *scale, *shift, *residual;
  int relu, l2norm;
  float *out;
  float *partial;   // split-K partial sums [S][n_slots][cout] (S = gridDim.z > 1)
  int *tickets;     // optional arrival counters [n_tiles][n_slabs] (zero on entry, left zero): the last
                    // partition to arrive reduces the tile in-kernel instead of a second launch
  int ablate;       // debugging only (env IMF_ABLATE): bit0 no MFMA, bit1 no LDS add, bit2 no A gather, bit3 no B load
  // Tail balancing (variant 6, gridDim.z == 1): tiles >= tail_begin are split tail_split ways over their
  // active offsets so that the last, partial round of workgroups per CU is made of small pieces; their
  // partial sums live in `partial` as [tail_split][n_slots - 64 tail_begin][cout].  0 = off.
  int tail_begin, tail_split;
  // variant 6: the split-f16 weight image is stored scaled by a power of two (so that the lo halves stay
  // normal f16 numbers); *w_unscale = 2^-s is multiplied back into the fp32 accumulators (exact).  NULL = 1.
  const float *w_unscale;
  // Capacity mode (whole-forward graphs): n_slots / n_out are capacities, the actual row count lives on the
  // device.  Tiles beyond the actual slots exit at once; with dyn_split_kvol != 0 the number of kernel-offset
  // partitions is the automatic rule evaluated on the ACTUAL rows (gridDim.z covers the largest it can return),
  // so the sums are formed exactly as by an exact-size launch.
  const int32_t *n_out_dev;
  int dyn_split_kvol;       // third argument of the split rule (active offsets per tile); 0 = gridDim.z is the split
  int slots_extra;          // slots the rulebook lays out beyond roundup64(rows): 0, or 512 for transposed maps
  int split_min_blocks, split_target;
  int no_xcd_swizzle;       // A/B switch (env IMF_H3_NO_XCD): plain blockIdx.x -> tile order
  int w_xcd;                // k_spconv_w: slab = f(XCD) workgroup order (env IMF_W_XCD)
  int geglu;                // epilogue of the fusion block's first feed-forward GEMM (variant 6, 64-column slabs, unsplit): the
                            // packed columns of slab y are [32 values | 32 gates] of hidden units 32 y .. 32 y + 31; the output
                            // is [n_out, cout / 2]: out = (v + shift_v) * gelu(g + shift_g), exact-erf GELU
                            // (model/attention_fusion.py:20-23 GEGLU)
  int32_t *err;             // flag word (optional): 16 = the rule wanted more partitions than the launch covers
                            // (capacity mode); 32 = an output value left the f16 range (|y| >= 65504 or NaN): the
                            // next split-f16 convolution would turn it into inf -- see IMF_FLAG_RANGE
};

constexpr float kF16Max = 65504.f;
// true when y cannot be carried by the split-f16 operands of the next convolution (also for NaN)
__device__ __forceinline__ bool out_of_f16_range(float y) { return !(fabsf(y) < kF16Max); }

// The automatic split-K rule (imf_spconv_auto_split), shared by host and device.
__host__ __device__ inline int auto_split_rule(long long n_slots, int cout, int kvol, int min_blocks, int target) {
  if (kvol <= 1 || kvol >= 28) return 1;
  const long long blocks = (n_slots / IMF_TILE_ROWS) * (cout / (16 * ((cout % 64 == 0) ? 4 : 2)));
  if (blocks >= min_blocks || blocks <= 0) return 1;
  long long s = (target + blocks - 1) / blocks;
  if (s > 8) s = 8;
  if (s > kvol / 2) s = kvol / 2;
  return s < 1 ? 1 : (int)s;
}

// actual rows / slots of a launch (capacity mode reads them from the device)
__device__ __forceinline__ long long conv_rows(const ConvParams &p) {
  if (!p.n_out_dev) return p.n_out;
  const long long n = *p.n_out_dev;
  return n < p.n_out ? n : p.n_out;
}
__device__ __forceinline__ long long conv_slots(const ConvParams &p, long long rows) {
  const long long s = (rows + IMF_TILE_ROWS - 1) / IMF_TILE_ROWS * IMF_TILE_ROWS + p.slots_extra;
  return s < p.n_slots ? s : p.n_slots;
}

// Packed weight image: [y][k][cc][j][cb][lane][t] with
//   ci = cc*CI_CHUNK + 16 j + 4 (lane>>4) + t,  co = y*CW + 16 cb + (lane&15)
// i.e. one "stage" (y,k,cc) is J*CO_BLK B-fragment quads, each 64 lanes x float4, contiguous.
__host__ __device__ inline int ci_chunk_of(int cin) { return (cin % 64 == 0) ? 64 : 32; }
__host__ __device__ inline int co_blk_of(int cout) { return (cout % 64 == 0) ? 4 : 2; }

__device__ __forceinline__ int row_of_slot(const ConvParams &p, long long slot) {
  if (p.tile_rows) return p.tile_rows[slot];
  return slot < conv_rows(p) ? (int)slot : -1;
}

__device__ __forceinline__ float4 gather_a(const ConvParams &p, int irow, int ci) {
  if (irow < 0) return make_float4(0.f, 0.f, 0.f, 0.f);
  const float *src = (ci < p.c_a) ? p.in_a + (long long)irow * p.c_a + ci
                                  : p.in_b + (long long)irow * p.c_b + (ci - p.c_a);
  return *reinterpret_cast<const float4 *>(src);
}

// acc[cb][r] = out[row 4*q4 + r of the wavefront's 16][col 16*cb + r16]
template <int CO_BLK>
__device__ __forceinline__ void conv_epilogue(const ConvParams &p, const f32x4 (&acc)[CO_BLK], int tile,
                                              int y, int wave, int r16, int q4, float unscale = 1.f) {
  const int CW = 16 * CO_BLK;
  int orow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    orow[r] = row_of_slot(p, (long long)tile * IMF_TILE_ROWS + wave * 16 + q4 * 4 + r);

  if (CO_BLK == 4 && p.geglu) {   // column blocks 0, 1: values; 2, 3: the gates of the same hidden units
    const int half = p.cout / 2;
    bool bad = false;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int pc = y * CW + cb * 16 + r16;               // packed column of the value; its gate sits 32 further
      const float bv = p.shift ? p.shift[pc] : 0.f, bg = p.shift ? p.shift[pc + 32] : 0.f;
      const int oc = y * 32 + cb * 16 + r16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float val = acc[cb][r] * unscale + bv, gate = acc[(cb + 2) % CO_BLK][r] * unscale + bg;
        const float g = val * (0.5f * gate * (1.f + erff(gate * 0.70710678118654752f)));
        bad |= orow[r] >= 0 && out_of_f16_range(g);
        if (orow[r] >= 0) p.out[(long long)orow[r] * half + oc] = g;
      }
    }
    if (p.err && __ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(p.err, 32);
    return;
  }

  float v[CO_BLK][4];
#pragma unroll
  for (int cb = 0; cb < CO_BLK; ++cb) {
    const int col = y * CW + cb * 16 + r16;
    const float sc = p.scale ? p.scale[col] : 1.f;
    const float sh = p.shift ? p.shift[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x = (acc[cb][r] * unscale) * sc + sh;
      if (p.residual && orow[r] >= 0) x += p.residual[(long long)orow[r] * p.cout + col];
      if (p.relu) x = fmaxf(x, 0.f);
      v[cb][r] = x;
    }
  }
  if (p.err) {      // range guard for the consumer's f16 operands: one atomic per offending wavefront, none normally
    bool bad = false;
#pragma unroll
    for (int cb = 0; cb < CO_BLK; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) bad |= orow[r] >= 0 && out_of_f16_range(v[cb][r]);
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(p.err, 32);
  }
  if (p.l2norm) {   // whole row lives in this workgroup slab (cout == CW): reduce over 16 lanes
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float ss = 0.f;
#pragma unroll
      for (int cb = 0; cb < CO_BLK; ++cb) ss += v[cb][r] * v[cb][r];
      ss += __shfl_xor(ss, 1, 64);
      ss += __shfl_xor(ss, 2, 64);
      ss += __shfl_xor(ss, 4, 64);
      ss += __shfl_xor(ss, 8, 64);
      const float nrm = sqrtf(ss);
#pragma unroll
      for (int cb = 0; cb < CO_BLK; ++cb) v[cb][r] = v[cb][r] / nrm;   // no eps: resunet.py:230
    }
  }
#pragma unroll
  for (int cb = 0; cb < CO_BLK; ++cb) {
    const int col = y * CW + cb * 16 + r16;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (orow[r] >= 0) p.out[(long long)orow[r] * p.cout + col] = v[cb][r];
  }
}

// Sum of the S partial slabs of one 64-row tile (ascending partition order) + epilogue, by the 256
// threads of the LAST workgroup to arrive at the tile (same arithmetic as k_spconv_reduce).
template <int CW>
__device__ __forceinline__ void fused_reduce_tile(const ConvParams &p, int S, long long slot0, int y, int tid) {
  constexpr int LPR = CW / 4, RPI = 256 / LPR;
  const int c4 = tid % LPR, rsub = tid / LPR;
  const int col = y * CW + 4 * c4;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.scale) sc = *reinterpret_cast<const float4 *>(p.scale + col);
  if (p.shift) sh = *reinterpret_cast<const float4 *>(p.shift + col);
  const float un = p.w_unscale ? *p.w_unscale : 1.f;
#pragma unroll 1
  for (int it = 0; it < IMF_TILE_ROWS / RPI; ++it) {
    const long long slot = slot0 + it * RPI + rsub;
    const int orow = row_of_slot(p, slot);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (orow >= 0) {
      for (int zz = 0; zz < S; ++zz) {
        const float4 v = *reinterpret_cast<const float4 *>(p.partial + ((long long)zz * p.n_slots + slot) * p.cout + col);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      s.x = (s.x * un) * sc.x + sh.x; s.y = (s.y * un) * sc.y + sh.y; s.z = (s.z * un) * sc.z + sh.z; s.w = (s.w * un) * sc.w + sh.w;
      if (p.residual) {
        const float4 rr = *reinterpret_cast<const float4 *>(p.residual + (long long)orow * p.cout + col);
        s.x += rr.x; s.y += rr.y; s.z += rr.z; s.w += rr.w;
      }
      if (p.relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
      if (p.err && (out_of_f16_range(s.x) || out_of_f16_range(s.y) || out_of_f16_range(s.z) || out_of_f16_range(s.w)))
        atomicOr(p.err, 32);
    }
    if (p.l2norm) {
      float ss = s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w;
#pragma unroll
      for (int o = 1; o < LPR; o <<= 1) ss += __shfl_xor(ss, o, 64);
      const float nrm = sqrtf(ss);
      s.x /= nrm; s.y /= nrm; s.z /= nrm; s.w /= nrm;
    }
    if (orow >= 0) *reinterpret_cast<float4 *>(p.out + (long long)orow * p.cout + col) = s;
  }
}

constexpr int kKCache = 28;   // active offsets cached per workgroup (kvol <= 27 uses the pipelined kernels)
constexpr int kSubTab = 27 * 8 + 8;   // variant 6: sub-stage table entries per workgroup; kvol * cin / 32 must stay below it

// spconv_h3.hip: variant 6 (split-f16 MFMA); grid = (tiles, cout / (16 CB), split)
void launch_spconv_h3(const ConvParams &p, dim3 grid, int co_blk, hipStream_t st, int use = 0);
// spconv_g.hip: the same arithmetic with both operands staged by LDS-DMA (default; launch_spconv_h3 dispatches)
void launch_spconv_g(const ConvParams &p, dim3 grid, int co_blk, hipStream_t st, int use = 0);

// spconv_w.hip: variant 6 for the coarse levels -- one workgroup per (tile, 64-column slab), the tile's sub-stages split
// over its `waves` (8 or 4) wavefronts, partial tiles combined through LDS, epilogue in the same launch
void launch_spconv_w(const ConvParams &p, unsigned tiles, int waves, hipStream_t st);

// spconv.hip: imf_conv_first_bitgrid_dyn on a grid the caller already zeroed and filled (geometry.hip: k_emit_unique)
int conv_first_bitgrid_dyn_cleared(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, const int32_t *bbox_dev,
                                   int32_t *err, int ksize, uint32_t *grid, size_t grid_words, const float *w, int cout,
                                   const float *scale, const float *shift, int relu, float *out, hipStream_t stream);

}  // namespace imf
