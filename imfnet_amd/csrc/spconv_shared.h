// Pieces shared by the sparse-convolution kernels (spconv.hip, spconv_h3.hip).
#pragma once
#include "common.h"

namespace imf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Arithmetic of the LDS-DMA convolution kernels' main loops (template argument AR of k_spconv_g / k_spconv_w).
constexpr int kArF16x2 = 0;      // variant 6: fp32 rows split into f16 hi + lo in registers
constexpr int kArF16x2Pre = 1;   // variant 6 on split-f16 operand images (ConvParams::a_split)
constexpr int kArF32 = 2;        // variant 0: fp32 operands, v_mfma_f32_16x16x4_f32 (the reference's arithmetic)
constexpr int kArBf16x3 = 3;     // variant 3: fp32 operands as three bf16 parts each (exact), six v_mfma_f32_16x16x32_bf16

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// Eight fp32 values -> three bf16 parts each, x = p0 + p1 + p2 EXACTLY: p0 = bf16(x), p1 = bf16(x - p0), p2 = x - p0 - p1
// (round-to-nearest-even; the residuals are exact fp32 differences of <= 16 and <= 8 significant bits).  hipcc emits
// v_cvt_pk_bf16_f32 / v_pk_add_f32: ~4.5 VALU instructions per value.
__device__ __forceinline__ void split_b3(const float4 &x0, const float4 &x1, bf16x8 &p0, bf16x8 &p1, bf16x8 &p2) {
#ifdef IMF_B3_NOSPLIT_ABL   // timing experiment only (wrong results): what the in-register split costs
  p0 = __builtin_bit_cast(bf16x8, x0); p1 = __builtin_bit_cast(bf16x8, x1); p2 = __builtin_bit_cast(bf16x8, x0);
  return;
#endif
  const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const __bf16 h0 = (__bf16)v[t];
    const float r1 = v[t] - (float)h0;
    const __bf16 h1 = (__bf16)r1;
    p0[t] = h0;
    p1[t] = h1;
    p2[t] = (__bf16)(r1 - (float)h1);
  }
}
// The six products of one (row block, column block, 32-channel chunk), smallest terms first; a = {a0, a1, a2}, b likewise.
#define IMF_B3_TERMS(X) X(0, 2) X(1, 1) X(2, 0) X(0, 1) X(1, 0) X(0, 0)

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
  // Split-f16 operand images (round 3).  A variant-6 convolution multiplies hi/lo f16 halves of its input; instead of every
  // consumer converting the fp32 rows again (27 times per row for a 3x3x3 map: ~40 % of the main loop's VALU work), the
  // PRODUCER's epilogue can write the operand image itself: per row and 32-channel chunk 128 bytes (the size of the fp32
  // chunk, so strides, DMA and buffer sizes do not change) = [4 hi pieces | 4 lo pieces], piece j = the 8 halves of
  // channels {4j..4j+3, 16+4j..16+4j+3} -- exactly what lane (row, j) of a consumer builds from the fp32 chunk with split8,
  // so the products are bit for bit the same.  The fp32 value is not kept: a residual read takes float(hi) + float(lo),
  // which drops the last 2 of the 24 significant bits (relative 2^-22; the convolutions never saw them anyway).
  int a_split;              // in_a (and in_b) are operand images
  int res_split;            // `residual` is an operand image
  int out_split;            // write `out` as an operand image (not with l2norm / geglu)
  int arith;                // kArF16x2 (with a_split: kArF16x2Pre) or kArF32: which weight image w_packed is and which MFMAs run
  int32_t *err;             // flag word (optional): 16 = the rule wanted more partitions than the launch covers
                            // (capacity mode); 32 = an output value left the f16 range (|y| >= 65504 or NaN): the
                            // next split-f16 convolution would turn it into inf -- see IMF_FLAG_RANGE
};

// IMF_FLAG_RANGE is raised by split-f16 launches only: with fp32 / bf16x3 operands a large value is a value, not an error
__device__ __forceinline__ bool range_guard(const ConvParams &p) { return p.err && p.arith <= kArF16x2Pre; }

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

// Operand image addressing (ConvParams::a_split): the hi half of (row, channel) in halves; its lo half sits 32 further.
__device__ __forceinline__ long long split_index(long long row, int channels, int ch) {
  const int c32 = ch & 31;
  return row * channels * 2 + (ch >> 5) * 64 + ((c32 & 15) >> 2) * 8 + ((c32 >> 4) << 2) + (c32 & 3);
}
__device__ __forceinline__ void store_split(float *img, long long row, int channels, int ch, float x) {
  const _Float16 h = (_Float16)x;
  _Float16 *const dst = reinterpret_cast<_Float16 *>(img) + split_index(row, channels, ch);
  dst[0] = h;
  dst[32] = (_Float16)(x - (float)h);
}
__device__ __forceinline__ float load_split(const float *img, long long row, int channels, int ch) {
  const _Float16 *const src = reinterpret_cast<const _Float16 *>(img) + split_index(row, channels, ch);
  return (float)src[0] + (float)src[32];
}
// four consecutive channels ch .. ch + 3 (ch % 4 == 0) of one row: two 8-byte accesses
__device__ __forceinline__ void store_split4(float *img, long long row, int channels, int ch, const float4 &x) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const float v[4] = {x.x, x.y, x.z, x.w};
  f16x4 h, l;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    h[t] = (_Float16)v[t];
    l[t] = (_Float16)(v[t] - (float)h[t]);
  }
  _Float16 *const dst = reinterpret_cast<_Float16 *>(img) + split_index(row, channels, ch);
  *reinterpret_cast<f16x4 *>(dst) = h;
  *reinterpret_cast<f16x4 *>(dst + 32) = l;
}
__device__ __forceinline__ float4 load_split4(const float *img, long long row, int channels, int ch) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const _Float16 *const src = reinterpret_cast<const _Float16 *>(img) + split_index(row, channels, ch);
  const f16x4 h = *reinterpret_cast<const f16x4 *>(src), l = *reinterpret_cast<const f16x4 *>(src + 32);
  return make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]);
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
        if (orow[r] >= 0) {
          if (p.out_split) store_split(p.out, orow[r], half, oc, g);
          else p.out[(long long)orow[r] * half + oc] = g;
        }
      }
    }
    if (range_guard(p) && __ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(p.err, 32);
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
      if (p.residual && orow[r] >= 0)
        x += p.res_split ? load_split(p.residual, orow[r], p.cout, col) : p.residual[(long long)orow[r] * p.cout + col];
      if (p.relu) x = fmaxf(x, 0.f);
      v[cb][r] = x;
    }
  }
  if (range_guard(p)) {      // range guard for the consumer's f16 operands: one atomic per offending wavefront, none normally
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
    if (p.out_split) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (orow[r] >= 0) store_split(p.out, orow[r], p.cout, col, v[cb][r]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (orow[r] >= 0) p.out[(long long)orow[r] * p.cout + col] = v[cb][r];
    }
  }
}

// The same epilogue through LDS (round 3): the wavefront parks its 16 x CW accumulator block in a private LDS region and
// every lane takes whole 8-channel PIECES of a row -- channels {4j..4j+3, 16+4j..16+4j+3} of a 32-channel chunk, the unit
// of an operand image -- so a split residual is read and a split output written with 16-byte accesses (the per-value form
// above needs two 2-byte accesses each), and fp32 outputs go out as float4.  Element for element the arithmetic of
// conv_epilogue; not for l2norm / geglu launches.  `stage`: 16 x (CW + 4) floats of LDS owned by this wavefront.
template <int CO_BLK>
__device__ __forceinline__ void conv_epilogue_staged(const ConvParams &p, const f32x4 (&acc)[CO_BLK], float *stage, int tile,
                                                     int y, int wave, int lane, float unscale) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  constexpr int CW = 16 * CO_BLK, LD = CW + 4, PPR = CW / 8;   // pieces per row: 8 or 4
  const int r16 = lane & 15, q4 = lane >> 4;
#pragma unroll
  for (int cb = 0; cb < CO_BLK; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) stage[(q4 * 4 + r) * LD + cb * 16 + r16] = acc[cb][r];
  bool bad = false;
#pragma unroll
  for (int i = 0; i < CO_BLK / 2; ++i) {
    const int item = lane + 64 * i;
    const int row = item / PPR, pp = item % PPR;
    const int cloc = (pp >> 2) * 32 + (pp & 3) * 4;            // first channel of the piece inside the slab
    const int col = y * CW + cloc;
    const float4 a0 = *reinterpret_cast<const float4 *>(stage + row * LD + cloc);
    const float4 a1 = *reinterpret_cast<const float4 *>(stage + row * LD + cloc + 16);
    float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const long long orow = row_of_slot(p, (long long)tile * IMF_TILE_ROWS + wave * 16 + row);
    float sc[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, sh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.scale) {
      const float4 s0 = *reinterpret_cast<const float4 *>(p.scale + col), s1 = *reinterpret_cast<const float4 *>(p.scale + col + 16);
      sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
    }
    if (p.shift) {
      const float4 s0 = *reinterpret_cast<const float4 *>(p.shift + col), s1 = *reinterpret_cast<const float4 *>(p.shift + col + 16);
      sh[0] = s0.x; sh[1] = s0.y; sh[2] = s0.z; sh[3] = s0.w; sh[4] = s1.x; sh[5] = s1.y; sh[6] = s1.z; sh[7] = s1.w;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = (x[t] * unscale) * sc[t] + sh[t];   // the expression of conv_epilogue
    if (orow < 0) continue;
    if (p.residual) {
      if (p.res_split) {
        const _Float16 *const src = reinterpret_cast<const _Float16 *>(p.residual) + split_index(orow, p.cout, col);
        const h8 rh = *reinterpret_cast<const h8 *>(src), rl = *reinterpret_cast<const h8 *>(src + 32);
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] += (float)rh[t] + (float)rl[t];
      } else {
        const float4 r0 = *reinterpret_cast<const float4 *>(p.residual + orow * p.cout + col);
        const float4 r1 = *reinterpret_cast<const float4 *>(p.residual + orow * p.cout + col + 16);
        x[0] += r0.x; x[1] += r0.y; x[2] += r0.z; x[3] += r0.w; x[4] += r1.x; x[5] += r1.y; x[6] += r1.z; x[7] += r1.w;
      }
    }
    if (p.relu) {
#pragma unroll
      for (int t = 0; t < 8; ++t) x[t] = fmaxf(x[t], 0.f);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) bad |= out_of_f16_range(x[t]);
    if (p.out_split) {
      h8 oh, ol;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        oh[t] = (_Float16)x[t];
        ol[t] = (_Float16)(x[t] - (float)oh[t]);
      }
      _Float16 *const dst = reinterpret_cast<_Float16 *>(p.out) + split_index(orow, p.cout, col);
      *reinterpret_cast<h8 *>(dst) = oh;
      *reinterpret_cast<h8 *>(dst + 32) = ol;
    } else {
      *reinterpret_cast<float4 *>(p.out + orow * p.cout + col) = make_float4(x[0], x[1], x[2], x[3]);
      *reinterpret_cast<float4 *>(p.out + orow * p.cout + col + 16) = make_float4(x[4], x[5], x[6], x[7]);
    }
  }
  if (range_guard(p) && __ballot(bad) != 0ull && lane == 0) atomicOr(p.err, 32);
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
      if (range_guard(p) && (out_of_f16_range(s.x) || out_of_f16_range(s.y) || out_of_f16_range(s.z) || out_of_f16_range(s.w)))
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
void launch_spconv_w(const ConvParams &p, unsigned tiles, int waves, hipStream_t st, int use = 0);

// spconv.hip: imf_conv_first_bitgrid_dyn on a grid the caller already zeroed and filled (geometry.hip: k_emit_unique)
// the fusion block with its output optionally written as a split-f16 operand image (fusion.hip; for imf_resunet_forward)
int fusion_attention_dyn_fmt(const float *x, int64_t n_cap, const int32_t *n_dev, const int32_t *item_starts_dev,
                             int n_items, int32_t *err, const float *const *kt_packed, const float *const *v_packed,
                             int n_tokens, int tokens_padded, const imf_fusion_weights *w, float scale, float *out,
                             void *workspace, size_t workspace_bytes, void *stream, int out_split, int variant = 6);
int fusion_attention_batched_fmt(const float *x, int n_items, const int64_t *item_row0, const int64_t *item_rows,
                                 const float *const *kt_packed, const float *const *v_packed, int n_tokens,
                                 int tokens_padded, const imf_fusion_weights *w, float scale, float *out,
                                 void *workspace, size_t workspace_bytes, int32_t *flags, void *stream, int out_split,
                                 int variant = 6);
int conv_first_bitgrid_flags_fmt(const int32_t *coords, int64_t n, const int32_t *bbox, int ksize, uint32_t *grid,
                                 size_t grid_words, const float *w, int cout, const float *scale, const float *shift,
                                 int relu, float *out, int32_t *flags, hipStream_t stream, int out_split);
int conv_first_bitgrid_dyn_fmt(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, const int32_t *bbox_dev,
                               int32_t *err, int ksize, uint32_t *grid, size_t grid_words, const float *w, int cout,
                               const float *scale, const float *shift, int relu, float *out, hipStream_t stream,
                               int out_split);
int conv_first_and_map_dyn(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, const int32_t *bbox_dev, int32_t *err,
                           int ksize, uint32_t *grid, size_t grid_words, const float *w, int cout, const float *scale,
                           const float *shift, int relu, float *out, int out_split, const imf_slot *table, int64_t capacity,
                           int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask, hipStream_t st,
                           const float *w_image = nullptr);
int conv_first_bitgrid_dyn_cleared(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, const int32_t *bbox_dev,
                                   int32_t *err, int ksize, uint32_t *grid, size_t grid_words, const float *w, int cout,
                                   const float *scale, const float *shift, int relu, float *out, hipStream_t stream,
                                   int out_split = 0, const float *w_image = nullptr);

}  // namespace imf
