// Bottleneck fusion block (model/attention_fusion.py:132-154 with depth 0, one head):
//   x  = Attn(LN(x), ctx = LN(img tokens)) + x        q: 256->128, softmax over the image tokens, out 128->256
//   x  = FF(LN(x)) + x                                 256 -> 2 x 1024 (GEGLU, exact-erf GELU) -> 256
// In PyTorch this is 15 launches of ~5 us each on 1 k rows.  Round 3 runs it as THREE launches:
//   k_fusion_attn   the attention half (16 % of the FLOPs): a workgroup of 8 wavefronts owns 16 point rows, keeps LN, q,
//                   scores, probabilities and the attention output in LDS and walks its four GEMMs on fp32 MFMA with the
//                   N dimension split over the waves; writes y and LN2(y)
//   k_spconv_g      g = GEGLU(LN2(y) W1^T + b1): a Linear layer is a 1x1x1 convolution over the rows -- the sparse
//                   convolution kernel with a GEGLU epilogue (spconv_shared.h), 34 x 32 workgroups for a pair
//   k_spconv_w      z = g W2^T + b2 + y: the wave-split kernel, K = 1024 over 8 wavefronts, bias + residual epilogue
// The feed-forward (84 % of the FLOPs) thus runs on the split-f16 matrix pipe with weights staged through LDS by DMA,
// parallel over (row tile, column slab) with full K per accumulator -- no hidden-dimension split over workgroups, no
// partial sums, no reduce pass, no row-count-dependent variant choice.  Rounds 1-2 ran the whole block as one fp32-MFMA
// kernel whose 16-row workgroups each streamed all 3.5 MB of weights from L2 (114 us for a pair).
// The attention weight matrices -- and the image-dependent K^T / V, packed once per fragment on the image-branch
// stream -- are in the fragment-major layout of imf_pack_weights (kvol = 1); W1 / W2 are imf_pack_weights_split16
// images.  Deterministic (fixed summation order), fp32-class arithmetic throughout.
#include <stdlib.h>
#include <string.h>

#include "spconv_shared.h"   // (common.h + the internal *_fmt declarations)

namespace imf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFD = 256;     // point feature dim (CHANNELS[4])
constexpr int kFQ = 128;     // attention inner dim
constexpr int kFH = 1024;    // GEGLU hidden (dim * 4)
constexpr int kFRows = 16;

struct FusionParams {
  const float *x;
  long long n;               // total rows (all items)
  int n_items;               // batch items: rows [row0[b], row0[b] + rows[b]) attend to image b (gridDim.z)
  long long row0[IMF_MAX_BATCH], rows[IMF_MAX_BATCH];
  const float *ktp_b[IMF_MAX_BATCH], *vp_b[IMF_MAX_BATCH];   // packed K^T [kFQ x tokp] and V [tokp x kFQ] per item
  int ntok, tokp;            // valid tokens, padded to a multiple of 64
  float scale;
  const float *ln1g, *ln1b, *wq, *wo, *bo, *ln2g, *ln2b;
  float *y, *n2;             // out: y = Attn(LN(x)) + x and LN2(y), [n, 256] each (the feed-forward's residual and input)
  // Capacity mode: n is a capacity; the row count and every item's first row are read from the device (meta
  // block of imf_pyramid_build: count of the stride-8 level, its item-start words).
  const int32_t *n_dev, *starts_dev;
  int32_t *err;
};

// rows [row0, row_end) of batch item `item`; false = nothing to do for this workgroup
__device__ __forceinline__ bool fusion_item_rows(const FusionParams &p, int item, long long &row0, long long &row_end) {
  if (!p.n_dev) {
    row0 = p.row0[item];
    row_end = p.row0[item] + p.rows[item];
    return true;
  }
  long long n = *p.n_dev;
  n = n < p.n ? n : p.n;
  const int s0 = p.starts_dev[item];
  const int s1 = item + 1 < p.n_items ? p.starts_dev[item + 1] : (int)n;
  if (s0 < 0 || s1 < s0) {          // an item without rows: flagged, as the exact-size path raises
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(p.err, 8);
    return false;
  }
  row0 = s0;
  row_end = s1;
  return true;
}

// float4 holding the B fragments of MFMA steps 4u'..4u'+3 (u = 16-channel step) for column block c
__device__ __forceinline__ const float4 *bfrag(const float *packed, int ncc, int u, int c, int lane) {
  const int y = c >> 2, cb = c & 3, cc = u >> 2, j = u & 3;
  return reinterpret_cast<const float4 *>(packed) + ((((long long)y * ncc + cc) * 4 + j) * 4 + cb) * 64 + lane;
}

// acc[i] += A[16 x K] . B[K x 16] for NC column blocks cblk[i]; A in LDS (row stride lda, lda % 4 == 0).
// The B fragments come straight from L2 (each is used by exactly one wave of one workgroup), so the
// loop is a register software pipeline D steps deep: without it every 16-channel step exposes a full
// L2 round trip (measured: 110 us for the whole block vs ~35 with the pipeline).  K/16 % D == 0.
// (kfull, ub): the packed matrix has kfull rows and the A slice covers its 16-row steps ub .. ub + K/16.
template <int NC, int D>
__device__ __forceinline__ void gemm16(f32x4 (&acc)[NC], const float *A, int lda, int K, const float *packed,
                                       const int (&cblk)[NC], int lane, int kfull = 0, int ub = 0) {
  const int r16 = lane & 15, q4 = lane >> 4, ncc = (kfull ? kfull : K) / 64, U = K / 16;
  const float *arow = A + r16 * lda + 4 * q4;
  float4 bq[D][NC];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int i = 0; i < NC; ++i) bq[d][i] = *bfrag(packed, ncc, ub + d, cblk[i], lane);
#pragma unroll 1
  for (int u0 = 0; u0 < U; u0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int u = u0 + d;
      const float4 a = *reinterpret_cast<const float4 *>(arow + 16 * u);
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const float4 b = bq[d][i];
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[i], 0, 0, 0);
      }
      if (u + D < U) {
#pragma unroll
        for (int i = 0; i < NC; ++i) bq[d][i] = *bfrag(packed, ncc, ub + u + D, cblk[i], lane);
      }
    }
  }
}

// The same product in two calls, so that the B fragments of GEMM n + 1 (weights, K^T, V: none of them depends on what the
// workgroup computes) can be REQUESTED before GEMM n runs and land under it: frags_load issues U x NC 16-byte loads,
// frags_mma consumes them in the order gemm16 does (u ascending, 4 k-steps each, column blocks inner) -- identical sums.
// `uact` (<= U, uniform) = the steps that exist (K / 16).
template <int NC, int U>
struct Frags { float4 v[U][NC]; };
template <int NC, int U>
__device__ __forceinline__ void frags_load(Frags<NC, U> &f, const float *packed, int kfull, const int (&cblk)[NC], int lane,
                                           int uact = U) {
  const int ncc = kfull / 64;
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (u < uact) {
#pragma unroll
      for (int i = 0; i < NC; ++i) f.v[u][i] = *bfrag(packed, ncc, u, cblk[i], lane);
    }
}
template <int NC, int U>
__device__ __forceinline__ void frags_mma(f32x4 (&acc)[NC], const float *A, int lda, const Frags<NC, U> &f, int lane, int uact = U) {
  const int r16 = lane & 15, q4 = lane >> 4;
  const float *arow = A + r16 * lda + 4 * q4;
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (u < uact) {
      const float4 a = *reinterpret_cast<const float4 *>(arow + 16 * u);
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const float4 b = f.v[u][i];
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[i], 0, 0, 0);
      }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// LayerNorm of row `row` (256 wide, eps 1e-5, biased variance) from src to dst; one wave, 4 floats / lane
__device__ __forceinline__ void layer_norm_row(const float *src, float *dst, const float *g, const float *b, int lane) {
  const float4 v = *reinterpret_cast<const float4 *>(src + 4 * lane);
  const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / kFD);
  const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
  const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / kFD);
  const float rstd = rsqrtf(var + 1e-5f);
  const float4 gg = *reinterpret_cast<const float4 *>(g + 4 * lane), bb = *reinterpret_cast<const float4 *>(b + 4 * lane);
  *reinterpret_cast<float4 *>(dst + 4 * lane) =
      make_float4(dx * rstd * gg.x + bb.x, dy * rstd * gg.y + bb.y, dz * rstd * gg.z + bb.z, dw * rstd * gg.w + bb.w);
}

constexpr int kLdX = kFD + 4;       // 260: (row * lda) % 64 == 4 * row -> conflict-free ds_read_b128 of A
constexpr int kLdQ = kFQ + 4;       // 132
constexpr int kMaxTokP = 320;
constexpr int kLdS = kMaxTokP + 4;  // 324

// The attention half: y = to_out(softmax(to_q(LN1(x)) K^T * scale) V) + x, and n2 = LN2(y) for the feed-forward.
// 16 rows per workgroup of 8 wavefronts; every intermediate lives in LDS; exact fp32 MFMA (16 % of the block's FLOPs).
__global__ void __launch_bounds__(512)
k_fusion_attn(const FusionParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *X = lds;                          // [16][260]  x, later y
  float *N = X + kFRows * kLdX;            // [16][260]  LN(x), later LN(y)
  float *S = N + kFRows * kLdX;            // [16][324]  scores -> probabilities
  float *Q = S + kFRows * kLdS;            // [16][132]  q, later attention output
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, q4 = lane >> 4;
  const int item = blockIdx.z;
  long long row0, row_end;
  if (!fusion_item_rows(p, item, row0, row_end)) return;
  row0 += (long long)blockIdx.x * kFRows;
  if (row0 >= row_end) return;                                      // grid.x covers the largest item
  const float *ktp = p.ktp_b[item], *vp = p.vp_b[item];

  // Every GEMM's B fragments are requested one phase ahead (round 5): W_q before x is even loaded, K^T before the q GEMM
  // runs, V before the scores, W_o before P V -- each of the four dependent L2 round trips (~2.5 us) now lands under the
  // previous phase instead of in front of its own MFMAs.  Same fragments, same MFMA order: bit-identical to round 4.
  const int cb_q[1] = {wave};
  Frags<1, 16> f_q;
  frags_load<1, 16>(f_q, p.wq, kFD, cb_q, lane);       // K = 256: 16 steps

  // ---- load x, LayerNorm 1 (wave w: rows 2w, 2w+1) ------------------------------------------
  for (int rr = 0; rr < 2; ++rr) {
    const int r = 2 * wave + rr;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < row_end) v = *reinterpret_cast<const float4 *>(p.x + (row0 + r) * kFD + 4 * lane);
    *reinterpret_cast<float4 *>(X + r * kLdX + 4 * lane) = v;
  }
  __syncthreads();
  for (int rr = 0; rr < 2; ++rr) layer_norm_row(X + (2 * wave + rr) * kLdX, N + (2 * wave + rr) * kLdX, p.ln1g, p.ln1b, lane);

  // scores: N = tokp (<= 320 -> <= 20 column blocks, round-robin over waves; a wavefront's blocks c, c + 8, c + 16 in one
  // pass, a missing block repeats the last valid one)
  const int ncb_s = p.tokp / 16;
  const bool has_s = wave < ncb_s;
  const int c0 = has_s ? wave : 0, c1 = wave + 8 < ncb_s ? wave + 8 : c0, c2 = wave + 16 < ncb_s ? wave + 16 : c1;
  const int cb_s[3] = {c0, c1, c2};
  Frags<3, 8> f_s;
  frags_load<3, 8>(f_s, ktp, kFQ, cb_s, lane);        // K = 128: 8 steps x 3 blocks
  __syncthreads();

  // ---- q = LN(x) Wq^T : N = 128 -> column block = wave -----------------------------------------
  {
    f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
    frags_mma<1, 16>(acc, N, kLdX, f_q, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) Q[(4 * q4 + r) * kLdQ + wave * 16 + r16] = acc[0][r];
  }
  // o = P V: K = tokp (16-token steps), N = 128 -> column block = wave
  const int U_v = p.tokp / 16;
  Frags<1, kMaxTokP / 16> f_v;
  frags_load<1, kMaxTokP / 16>(f_v, vp, p.tokp, cb_q, lane, U_v);
  __syncthreads();

  // ---- scores = q K^T * scale ---------------------------------------------------------------------
  if (has_s) {
    f32x4 acc[3] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    frags_mma<3, 8>(acc, Q, kLdQ, f_s, lane);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i > 0 && cb_s[i] == cb_s[i - 1]) continue;       // a repeated (missing) block
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(4 * q4 + r) * kLdS + cb_s[i] * 16 + r16] = acc[i][r] * p.scale;
    }
  }
  // y = o Wo^T: N = 256 -> 2 column blocks per wave, K = 128
  const int cb_o[2] = {2 * wave, 2 * wave + 1};
  Frags<2, 8> f_o;
  frags_load<2, 8>(f_o, p.wo, kFQ, cb_o, lane);
  __syncthreads();

  // ---- softmax over the valid tokens (wave w: rows 2w, 2w+1; 5 columns per lane) ----------------
  for (int rr = 0; rr < 2; ++rr) {
    float *srow = S + (2 * wave + rr) * kLdS;
    float v[5], m = -3.0e38f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int c = lane + 64 * i;
      v[i] = (c < p.ntok) ? srow[c] : -3.0e38f;
      m = fmaxf(m, v[i]);
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int c = lane + 64 * i;
      v[i] = (c < p.ntok) ? expf(v[i] - m) : 0.f;
      s += v[i];
    }
    s = wave_sum(s);
    const float inv = 1.f / s;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int c = lane + 64 * i;
      if (c < p.tokp) srow[c] = v[i] * inv;
    }
  }
  __syncthreads();

  // ---- o = P V (written over q: q was last read before the previous barriers) --------------------
  {
    f32x4 acc[1] = {(f32x4){0.f, 0.f, 0.f, 0.f}};
    frags_mma<1, kMaxTokP / 16>(acc, S, kLdS, f_v, lane, U_v);
#pragma unroll
    for (int r = 0; r < 4; ++r) Q[(4 * q4 + r) * kLdQ + wave * 16 + r16] = acc[0][r];
  }
  __syncthreads();

  // ---- y = o Wo^T + bo + x; y replaces x in place -------------------------------------------------
  {
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    frags_mma<2, 8>(acc, Q, kLdQ, f_o, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = cb_o[i] * 16 + r16;
      const float bias = p.bo[col];
#pragma unroll
      for (int r = 0; r < 4; ++r) X[(4 * q4 + r) * kLdX + col] += acc[i][r] + bias;   // each element owned by one lane
    }
  }
  __syncthreads();
  for (int rr = 0; rr < 2; ++rr) layer_norm_row(X + (2 * wave + rr) * kLdX, N + (2 * wave + rr) * kLdX, p.ln2g, p.ln2b, lane);
  // ---- y and LN2(y) to HBM (wave w wrote rows 2w, 2w+1 of N itself; X was completed before the barrier above) ----
  for (int rr = 0; rr < 2; ++rr) {
    const int r = 2 * wave + rr;
    if (row0 + r < row_end) {
      const float4 yv = *reinterpret_cast<const float4 *>(X + r * kLdX + 4 * lane);
      const float4 nv = *reinterpret_cast<const float4 *>(N + r * kLdX + 4 * lane);
      // n2 feeds a split-f16 GEMM: f16 range guard (LayerNorm output: only a degenerate gain / bias can trip it)
      if (p.err && (!(fabsf(nv.x) < 65504.f) || !(fabsf(nv.y) < 65504.f) || !(fabsf(nv.z) < 65504.f) || !(fabsf(nv.w) < 65504.f)))
        atomicOr(p.err, 32);
      *reinterpret_cast<float4 *>(p.y + (row0 + r) * kFD + 4 * lane) = yv;
      *reinterpret_cast<float4 *>(p.n2 + (row0 + r) * kFD + 4 * lane) = nv;
    }
  }
}

static int launch_attn(const FusionParams &p, hipStream_t st) {
  const size_t lds = (size_t)kFRows * (2 * kLdX + kLdS + kLdQ) * sizeof(float);
  static bool attr_set = false;   // idempotent; a benign race at worst sets it twice
  if (!attr_set) {
    IMF_CHECK_HIP(hipFuncSetAttribute((const void *)k_fusion_attn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  long long max_rows = p.n_dev ? p.n : 0;
  for (int b = 0; b < p.n_items; ++b) max_rows = p.rows[b] > max_rows ? p.rows[b] : max_rows;
  k_fusion_attn<<<dim3((unsigned)div_up(max_rows, kFRows), 1, p.n_items), 512, lds, st>>>(p);
  IMF_CHECK_LAUNCH("k_fusion_attn");
  return IMF_OK;
}

// workspace: y [n,256] | n2 [n,256] | g [n,1024]
constexpr size_t kWsFloatsPerRow = 2 * kFD + kFH;

// The feed-forward half on the convolution kernels (a Linear layer is a 1x1x1 convolution over the rows):
//   g = GEGLU(n2 W1^T + b1)   k_spconv_g, 64-column slabs [32 values | 32 gates], GEGLU epilogue      (84 % of the
//   z = g W2^T + b2 + y       k_spconv_w, 8 wavefronts split K = 1024, bias + residual epilogue        block's FLOPs)
// both on the split-f16 matrix pipe (fp32-class arithmetic, spconv_g.hip) with weights staged through LDS by DMA.
// The GEGLU output travels as a split-f16 operand image (the only reader is the second GEMM: same products, no
// conversions in its loop); out_split: the block's output too (its reader is conv4_tr, imf_resunet_forward decides).
static int run_feed_forward(const imf_fusion_weights *w, long long n, const int32_t *n_dev, float *ws, float *out,
                            int32_t *err, hipStream_t st, int out_split, int variant = 6, int n_items = 0) {
  float *y = ws, *n2 = ws + (size_t)n * kFD, *g = ws + (size_t)n * 2 * kFD;
  const long long slots = (n + IMF_TILE_ROWS - 1) / IMF_TILE_ROWS * IMF_TILE_ROWS;
  imf_conv_args a;
  if (variant != 6 && variant != 3) {   // the network runs on fp32 MFMA (the f16-range recompute): so does the feed-forward, on fp32 images
    IMF_REQUIRE(variant == 0 && w->w1_f32 && w->w2_f32 && !out_split,
                "imf_fusion_attention: variant %d needs the fp32 feed-forward images (w1_f32 / w2_f32) and fp32 buffers", variant);
    memset(&a, 0, sizeof(a));
    a.in_a = n2; a.c_a = kFD; a.w_packed = w->w1_f32; a.kvol = 1; a.cout = 2 * kFH;
    a.n_slots = slots; a.n_out = n; a.shift = w->b1; a.out = g; a.split_k = 1; a.variant = 0; a.geglu = 1;
    a.n_out_dev = n_dev; a.dyn_err = err;
    int rc0 = imf_spconv_fwd(&a, st);
    if (rc0) return rc0;
    memset(&a, 0, sizeof(a));
    a.in_a = g; a.c_a = kFH; a.w_packed = w->w2_f32; a.kvol = 1; a.cout = kFD;
    a.n_slots = slots; a.n_out = n; a.shift = w->b2; a.residual = y; a.out = out; a.split_k = 1; a.variant = 0;
    a.kernel_tag = n_items >= 2 ? (4 | 128) : 4;   // wave-split like the launch below (AR = kArF32)
    a.n_out_dev = n_dev; a.dyn_err = err;
    return imf_spconv_fwd(&a, st);
  }
  // variant 6: split-f16 images, the GEGLU output travels as an operand image.  variant 3: bf16x3 images (w1_p / w2_p packed
  // by imf_pack_weights_bf16x3), fp32 buffers throughout
  const bool b3 = variant == 3;
  IMF_REQUIRE(!b3 || !out_split, "imf_fusion_attention: variant 3 writes fp32 rows");
  memset(&a, 0, sizeof(a));
  a.in_a = n2; a.c_a = kFD; a.w_packed = w->w1_p; a.kvol = 1; a.cout = 2 * kFH;
  a.n_slots = slots; a.n_out = n; a.shift = w->b1; a.out = g; a.split_k = 1; a.variant = variant; a.geglu = 1;
  a.n_out_dev = n_dev; a.dyn_err = err;
  a.operand_format = b3 ? 0 : IMF_FMT_OUT_SPLIT;
  int rc = imf_spconv_fwd(&a, st);
  if (rc) return rc;
  memset(&a, 0, sizeof(a));
  a.in_a = g; a.c_a = kFH; a.w_packed = w->w2_p; a.kvol = 1; a.cout = kFD;
  a.n_slots = slots; a.n_out = n; a.shift = w->b2; a.residual = y; a.out = out; a.split_k = 1; a.variant = variant;
  // wave-split, 8 wavefronts: K = 1024 is 32 sub-stages per tile; one fragment per forward on bf16x3 (17 tiles x 4 slabs):
  // half-tile workgroups of 4 wavefronts, a pair: 48-row units (imf_resunet_conv_kernel_tag's rules for the stride-8 level)
  a.kernel_tag = (b3 && n_items == 1) ? (8 | 64) : n_items >= 2 ? (4 | 128) : 4;
  a.n_out_dev = n_dev; a.dyn_err = err;              // z feeds conv4_tr (split-f16): range guard
  a.operand_format = b3 ? 0 : (IMF_FMT_A_SPLIT | (out_split ? IMF_FMT_OUT_SPLIT : 0));
  return imf_spconv_fwd(&a, st);
}

static void fill_params(FusionParams &p, const float *x, const imf_fusion_weights *w, int n_tokens, int tokens_padded,
                        float scale, float *ws, long long n) {
  p.x = x; p.n = n; p.ntok = n_tokens; p.tokp = tokens_padded; p.scale = scale;
  p.ln1g = w->ln1_g; p.ln1b = w->ln1_b; p.wq = w->wq_p; p.wo = w->wo_p; p.bo = w->bo; p.ln2g = w->ln2_g; p.ln2b = w->ln2_b;
  p.y = ws; p.n2 = ws + (size_t)n * kFD;
}

}  // namespace imf

using namespace imf;

extern "C" {

size_t imf_fusion_workspace_bytes(int64_t n) { return (size_t)n * kWsFloatsPerRow * sizeof(float); }
size_t imf_fusion_workspace_bytes_cap(int64_t n_cap) { return imf_fusion_workspace_bytes(n_cap); }

int imf_fusion_attention_dyn(const float *x, int64_t n_cap, const int32_t *n_dev, const int32_t *item_starts_dev,
                             int n_items, int32_t *err, const float *const *kt_packed, const float *const *v_packed,
                             int n_tokens, int tokens_padded, const imf_fusion_weights *w, float scale, float *out,
                             void *workspace, size_t workspace_bytes, void *stream) {
  return imf::fusion_attention_dyn_fmt(x, n_cap, n_dev, item_starts_dev, n_items, err, kt_packed, v_packed, n_tokens,
                                       tokens_padded, w, scale, out, workspace, workspace_bytes, stream, 0);
}

}  // extern "C"

namespace imf {
int fusion_attention_dyn_fmt(const float *x, int64_t n_cap, const int32_t *n_dev, const int32_t *item_starts_dev,
                             int n_items, int32_t *err, const float *const *kt_packed, const float *const *v_packed,
                             int n_tokens, int tokens_padded, const imf_fusion_weights *w, float scale, float *out,
                             void *workspace, size_t workspace_bytes, void *stream, int out_split, int variant) {
  IMF_REQUIRE(x && n_dev && item_starts_dev && err && kt_packed && v_packed && w && out && workspace,
              "imf_fusion_attention_dyn: null pointer");
  IMF_REQUIRE(n_items >= 1 && n_items <= IMF_MAX_BATCH && n_cap > 0, "imf_fusion_attention_dyn: n_items=%d", n_items);
  IMF_REQUIRE(tokens_padded % 64 == 0 && tokens_padded <= kMaxTokP && n_tokens > 0 && n_tokens <= tokens_padded,
              "imf_fusion_attention_dyn: tokens=%d padded=%d", n_tokens, tokens_padded);
  IMF_REQUIRE(workspace_bytes >= imf_fusion_workspace_bytes_cap(n_cap) && ((uintptr_t)workspace & 15) == 0,
              "imf_fusion_attention_dyn: needs %zu workspace bytes, 16-byte aligned", imf_fusion_workspace_bytes_cap(n_cap));
  FusionParams p;
  memset(&p, 0, sizeof(p));
  for (int b = 0; b < n_items; ++b) {
    IMF_REQUIRE(kt_packed[b] && v_packed[b], "imf_fusion_attention_dyn: item %d", b);
    p.ktp_b[b] = kt_packed[b];
    p.vp_b[b] = v_packed[b];
  }
  fill_params(p, x, w, n_tokens, tokens_padded, scale, (float *)workspace, n_cap);
  p.n_items = n_items;
  p.n_dev = n_dev; p.starts_dev = item_starts_dev; p.err = err;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_attn(p, st);
  if (rc) return rc;
  return run_feed_forward(w, n_cap, n_dev, (float *)workspace, out, err, st, out_split, variant, n_items);
}
}  // namespace imf

extern "C" {

int imf_fusion_attention_batched(const float *x, int n_items, const int64_t *item_row0, const int64_t *item_rows,
                                 const float *const *kt_packed, const float *const *v_packed, int n_tokens,
                                 int tokens_padded, const imf_fusion_weights *w, float scale, float *out,
                                 void *workspace, size_t workspace_bytes, void *stream) {
  return imf_fusion_attention_batched_flags(x, n_items, item_row0, item_rows, kt_packed, v_packed, n_tokens, tokens_padded,
                                            w, scale, out, workspace, workspace_bytes, nullptr, stream);
}

int imf_fusion_attention_batched_flags(const float *x, int n_items, const int64_t *item_row0, const int64_t *item_rows,
                                       const float *const *kt_packed, const float *const *v_packed, int n_tokens,
                                       int tokens_padded, const imf_fusion_weights *w, float scale, float *out,
                                       void *workspace, size_t workspace_bytes, int32_t *flags, void *stream) {
  return imf::fusion_attention_batched_fmt(x, n_items, item_row0, item_rows, kt_packed, v_packed, n_tokens, tokens_padded, w,
                                           scale, out, workspace, workspace_bytes, flags, stream, 0, 6);
}

int imf_fusion_attention_batched_v(const float *x, int n_items, const int64_t *item_row0, const int64_t *item_rows,
                                   const float *const *kt_packed, const float *const *v_packed, int n_tokens,
                                   int tokens_padded, const imf_fusion_weights *w, float scale, float *out,
                                   void *workspace, size_t workspace_bytes, int32_t *flags, int variant, void *stream) {
  return imf::fusion_attention_batched_fmt(x, n_items, item_row0, item_rows, kt_packed, v_packed, n_tokens, tokens_padded, w,
                                           scale, out, workspace, workspace_bytes, flags, stream, 0, variant);
}

}  // extern "C"

namespace imf {
int fusion_attention_batched_fmt(const float *x, int n_items, const int64_t *item_row0, const int64_t *item_rows,
                                 const float *const *kt_packed, const float *const *v_packed, int n_tokens,
                                 int tokens_padded, const imf_fusion_weights *w, float scale, float *out,
                                 void *workspace, size_t workspace_bytes, int32_t *flags, void *stream, int out_split,
                                 int variant) {
  IMF_REQUIRE(x && item_row0 && item_rows && kt_packed && v_packed && w && out, "imf_fusion_attention: null pointer");
  IMF_REQUIRE(n_items >= 1 && n_items <= IMF_MAX_BATCH, "imf_fusion_attention: n_items=%d", n_items);
  IMF_REQUIRE(w->ln1_g && w->ln1_b && w->wq_p && w->wo_p && w->bo && w->ln2_g && w->ln2_b && w->w1_p && w->b1 &&
                  w->w2_p && w->b2, "imf_fusion_attention: null weight pointer");
  IMF_REQUIRE(tokens_padded % 64 == 0 && tokens_padded <= kMaxTokP && n_tokens > 0 && n_tokens <= tokens_padded,
              "imf_fusion_attention: tokens=%d padded=%d (padded %% 64 == 0, <= %d)", n_tokens, tokens_padded, kMaxTokP);
  FusionParams p;
  memset(&p, 0, sizeof(p));
  long long n = 0;
  for (int b = 0; b < n_items; ++b) {
    IMF_REQUIRE(item_rows[b] > 0 && item_row0[b] >= 0 && kt_packed[b] && v_packed[b], "imf_fusion_attention: item %d", b);
    p.row0[b] = item_row0[b];
    p.rows[b] = item_rows[b];
    p.ktp_b[b] = kt_packed[b];
    p.vp_b[b] = v_packed[b];
    n = n > item_row0[b] + item_rows[b] ? n : item_row0[b] + item_rows[b];
  }
  IMF_REQUIRE(workspace && workspace_bytes >= imf_fusion_workspace_bytes(n) && ((uintptr_t)workspace & 15) == 0,
              "imf_fusion_attention: needs %zu workspace bytes, 16-byte aligned", imf_fusion_workspace_bytes(n));
  fill_params(p, x, w, n_tokens, tokens_padded, scale, (float *)workspace, n);
  p.n_items = n_items;
  p.err = flags;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_attn(p, st);
  if (rc) return rc;
  return run_feed_forward(w, n, nullptr, (float *)workspace, out, flags, st, out_split, variant, n_items);
}
}  // namespace imf

extern "C" {

int imf_fusion_attention(const float *x, int64_t n, const float *kt_packed, const float *v_packed, int n_tokens,
                         int tokens_padded, const imf_fusion_weights *w, float scale, float *out, void *workspace,
                         size_t workspace_bytes, void *stream) {
  IMF_REQUIRE(n > 0, "imf_fusion_attention: n");
  const int64_t row0 = 0;
  return imf_fusion_attention_batched(x, 1, &row0, &n, &kt_packed, &v_packed, n_tokens, tokens_padded, w, scale, out,
                                      workspace, workspace_bytes, stream);
}

}  // extern "C"
