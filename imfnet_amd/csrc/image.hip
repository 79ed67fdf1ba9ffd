// Image branch of the fusion block: ResNet-34 trunk truncated after layer2 (model/resnet.py:195-216,
// model/Img_Encoder.py:15-18) + the image-only half of the cross attention (LayerNorm of the tokens and the
// K/V projection, model/attention_fusion.py:36-46,84), on the same matrix-pipe kernel as the sparse
// convolutions.
//
// A dense 3x3 convolution over an H x W map IS a sparse convolution whose rulebook happens to be full:
// rows = pixels (NHWC, so the [B*H*W, C] feature matrix is exactly the layout imf_spconv_fwd gathers
// from), kernel volume 9, nbr[k][pixel] = the pixel one tap away (-1 outside the map), stride 2 and the
// 1x1 down-sample projection are just other static tables.  The tables depend only on (B, H, W), are
// built once per image shape by k_img_rulebook and stay in HBM; BatchNorm is folded into the epilogue
// scale/shift, ReLU and the BasicBlock residual ride there too.  The 7x7/2 stem has 3 input channels,
// so it runs as im2col ([pixels, 147 -> 160]) + a pointwise convolution.  Net effect per fragment:
// ~20 launches of this library instead of ~100 MIOpen / aten launches of whole-GPU fp32 Winograd +
// separate BatchNorm / ReLU / add kernels, and the trunk's arithmetic goes through the split-f16 MFMA path.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace imf {
namespace {

struct ImgShape {
  int B, H, W;          // input image
  int H2, W2;           // after the 7x7 stride-2 stem
  int H4, W4;           // after the 3x3 stride-2 max pool (layer1 resolution)
  int H8, W8;           // layer2 resolution; tokens per image = H8 * W8
  int64_t n2, n4, n8;   // rows (B * H * W) at each resolution
};

ImgShape shape_of(int B, int H, int W) {
  ImgShape s;
  s.B = B; s.H = H; s.W = W;
  s.H2 = (H + 6 - 7) / 2 + 1; s.W2 = (W + 6 - 7) / 2 + 1;
  s.H4 = (s.H2 + 2 - 3) / 2 + 1; s.W4 = (s.W2 + 2 - 3) / 2 + 1;
  s.H8 = (s.H4 + 2 - 3) / 2 + 1; s.W8 = (s.W4 + 2 - 3) / 2 + 1;
  s.n2 = (int64_t)B * s.H2 * s.W2; s.n4 = (int64_t)B * s.H4 * s.W4; s.n8 = (int64_t)B * s.H8 * s.W8;
  return s;
}

constexpr int kStemK = 160;     // 7*7*3 = 147 im2col columns, zero-padded to a multiple of 32
constexpr int kC1 = 64, kC2 = 128, kKV = 256;

struct Table {   // static rulebook of one dense convolution geometry
  int32_t *tile_rows, *nbr;
  uint32_t *tile_mask;
  int64_t n_slots, n_out;
  int kvol;
};

size_t table_words(int64_t n_out, int kvol) {
  const int64_t n_slots = imf_rulebook_slots(n_out);
  return (size_t)n_slots * (1 + kvol) + (size_t)(n_slots / IMF_TILE_ROWS) * IMF_MASK_WORDS;
}

int32_t *place_table(Table &t, int32_t *p, int64_t n_out, int kvol) {
  t.n_out = n_out; t.kvol = kvol; t.n_slots = imf_rulebook_slots(n_out);
  t.tile_rows = p; p += t.n_slots;
  t.nbr = p; p += (size_t)kvol * t.n_slots;
  t.tile_mask = (uint32_t *)p; p += (size_t)(t.n_slots / IMF_TILE_ROWS) * IMF_MASK_WORDS;
  return p;
}

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// Workspace layout: [tables: s1@4, s2 4->8, down 4->8, s1@8] [im2col] [stem out] [4 buffers @4] [4 buffers @8]
// [LayerNorm'ed tokens] [kv] [split-K scratch]
struct Plan {
  Table t4, t48, td, t8;
  float *im2col, *stem, *b4[3], *b8[4], *ln, *kv, *splitk;
  size_t splitk_bytes, total_bytes, table_bytes;
};

size_t splitk_need(int64_t n_out, int cout, int kvol) {
  const int64_t n_slots = imf_rulebook_slots(n_out);
  return imf_spconv_workspace_bytes(n_slots, cout, imf_spconv_auto_split(n_slots, cout, kvol));
}

Plan plan_of(const ImgShape &s, void *ws) {
  Plan p;
  char *base = (char *)ws, *q = base;
  int32_t *ip = (int32_t *)q;
  ip = place_table(p.t4, ip, s.n4, 9);
  ip = place_table(p.t48, ip, s.n8, 9);
  ip = place_table(p.td, ip, s.n8, 1);
  ip = place_table(p.t8, ip, s.n8, 9);
  p.table_bytes = align256((char *)ip - base);
  q = base + p.table_bytes;
  auto take = [&](size_t floats) {
    float *r = (float *)q;
    q += align256(floats * sizeof(float));
    return r;
  };
  p.im2col = take((size_t)s.n2 * kStemK);
  p.stem = take((size_t)s.n2 * kC1);
  for (int i = 0; i < 3; ++i) p.b4[i] = take((size_t)s.n4 * kC1);
  for (int i = 0; i < 4; ++i) p.b8[i] = take((size_t)s.n8 * kC2);
  p.ln = take((size_t)s.n8 * kC2);
  p.kv = take((size_t)s.n8 * kKV);
  size_t need = 0;
  auto consider = [&](size_t v) { need = v > need ? v : need; };
  consider(splitk_need(s.n2, kC1, 1));
  consider(splitk_need(s.n4, kC1, 9));
  consider(splitk_need(s.n8, kC2, 9));
  consider(splitk_need(s.n8, kC2, 1));
  consider(splitk_need(s.n8, kKV, 1));
  p.splitk_bytes = need;
  p.splitk = (float *)q;
  q += align256(need);
  p.total_bytes = (size_t)(q - base);
  return p;
}

// ---- static tables ------------------------------------------------------------------------------------
// one thread per (slot, k), slot fastest (the layout of geometry.hip's k_rulebook): in = out * stride + tap - pad
__global__ void __launch_bounds__(256)
k_img_rulebook(int B, int Hin, int Win, int Hout, int Wout, int ksize, int stride, int pad, int64_t n_slots,
               int32_t *__restrict__ tile_rows, int32_t *__restrict__ nbr, uint32_t *__restrict__ tile_mask) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int kvol = ksize * ksize;
  if (idx >= n_slots * kvol) return;
  const int k = (int)(idx / n_slots);
  const int64_t slot = idx - (int64_t)k * n_slots;
  const int64_t n_out = (int64_t)B * Hout * Wout;
  const int row = slot < n_out ? (int)slot : -1;
  if (k == 0) tile_rows[slot] = row;
  int found = -1;
  if (row >= 0) {
    const int b = row / (Hout * Wout), r = row - b * (Hout * Wout);
    const int oy = r / Wout, ox = r - oy * Wout;
    const int iy = oy * stride + k / ksize - pad, ix = ox * stride + k % ksize - pad;
    if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win) found = (b * Hin + iy) * Win + ix;
  }
  nbr[idx] = found;
  const unsigned long long any = __ballot(found >= 0);
  if (any != 0ull && (threadIdx.x & 63) == 0)
    atomicOr(tile_mask + (slot >> 6) * IMF_MASK_WORDS + (k >> 5), 1u << (k & 31));
}

// ---- stem: im2col of the 7x7 stride-2 pad-3 convolution (model/resnet.py:141,199) ---------------------
// column c = (ky*7 + kx)*3 + ch (the order the host lays the stem weight out in); columns 147..159 = 0
__global__ void __launch_bounds__(256)
k_img_im2col7(const float *__restrict__ img, int B, int H, int W, int Ho, int Wo, float *__restrict__ out,
              int32_t *flags) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * Ho * Wo * kStemK;
  if (idx >= total) return;
  const int c = (int)(idx % kStemK);
  const int64_t row = idx / kStemK;
  float v = 0.f;
  if (c < 147) {
    const int tap = c / 3, ch = c - 3 * tap, ky = tap / 7, kx = tap - 7 * ky;
    const int b = (int)(row / (Ho * Wo)), r = (int)(row - (int64_t)b * Ho * Wo);
    const int oy = r / Wo, ox = r - oy * Wo;
    const int iy = 2 * oy + ky - 3, ix = 2 * ox + kx - 3;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = img[(((int64_t)b * 3 + ch) * H + iy) * W + ix];
  }
  if (flags && !(fabsf(v) < 65504.f)) atomicOr(flags, 32);   // pixel values must fit the f16 operands of the stem
  out[idx] = v;
}

// ---- 3x3 stride-2 pad-1 max pool on NHWC rows (model/resnet.py:144,202), 4 channels per thread ----------
__global__ void __launch_bounds__(256)
k_img_maxpool(const float *__restrict__ in, int B, int Hin, int Win, int Hout, int Wout, int C,
              float *__restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4n = C / 4;
  const int64_t total = (int64_t)B * Hout * Wout * c4n;
  if (idx >= total) return;
  const int c4 = (int)(idx % c4n);
  const int64_t row = idx / c4n;
  const int b = (int)(row / (Hout * Wout)), r = (int)(row - (int64_t)b * Hout * Wout);
  const int oy = r / Wout, ox = r - oy * Wout;
  float4 m = make_float4(-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = 2 * oy + ky - 1;
    if (iy < 0 || iy >= Hin) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = 2 * ox + kx - 1;
      if (ix < 0 || ix >= Win) continue;
      const float4 v = *reinterpret_cast<const float4 *>(in + (((int64_t)b * Hin + iy) * Win + ix) * C + 4 * c4);
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
  }
  *reinterpret_cast<float4 *>(out + row * C + 4 * c4) = m;
}

// ---- LayerNorm over 128-wide token rows (norm_context, attention_fusion.py:36-46): one wave per row -----
__global__ void __launch_bounds__(256)
k_img_layernorm128(const float *__restrict__ x, int64_t n, const float *__restrict__ g, const float *__restrict__ b,
                   float *__restrict__ y, int32_t *flags) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float2 v = *reinterpret_cast<const float2 *>(x + row * kC2 + 2 * lane);
  float s = v.x + v.y;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s * (1.f / kC2);
  const float dx = v.x - mean, dy = v.y - mean;
  float q = dx * dx + dy * dy;
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q * (1.f / kC2) + 1e-5f);
  const float2 gg = *reinterpret_cast<const float2 *>(g + 2 * lane), bb = *reinterpret_cast<const float2 *>(b + 2 * lane);
  const float2 o = make_float2(dx * rstd * gg.x + bb.x, dy * rstd * gg.y + bb.y);
  if (flags && (!(fabsf(o.x) < 65504.f) || !(fabsf(o.y) < 65504.f))) atomicOr(flags, 32);
  *reinterpret_cast<float2 *>(y + row * kC2 + 2 * lane) = o;
}

// ---- K^T / V of every image in the fragment-major fp32 layout the fusion kernel streams ------------------
// (imf_pack_weights with kvol = 1 applied to K^T [128, tokp] and V [tokp, 128], tokens zero-padded to tokp):
//   packed index -> (y, cc, j, cb, lane, t):  ci = cc*64 + 16 j + 4 (lane>>4) + t,  co = y*64 + 16 cb + (lane&15)
__global__ void __launch_bounds__(256)
k_img_pack_kv(const float *__restrict__ kv, int ntok, int tokp, float *__restrict__ kt_packed,
              float *__restrict__ v_packed) {
  const int per = 128 * tokp;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int item = blockIdx.y;
  if (idx >= 2 * (int64_t)per) return;
  const bool is_v = idx >= per;
  int r = (int)(is_v ? idx - per : idx);
  const int cin = is_v ? tokp : 128;
  const int ncc = cin / 64;
  const int t = r & 3; r >>= 2;
  const int lane = r & 63; r >>= 6;
  const int cb = r & 3; r >>= 2;
  const int j = r & 3; r >>= 2;
  const int cc = r % ncc;
  const int y = r / ncc;
  const int ci = cc * 64 + 16 * j + 4 * (lane >> 4) + t;
  const int co = y * 64 + 16 * cb + (lane & 15);
  const float *src = kv + (int64_t)item * ntok * kKV;
  float v = 0.f;
  if (!is_v) {          // K^T[ci = channel][co = token]
    if (co < ntok) v = src[(int64_t)co * kKV + ci];
    kt_packed[(int64_t)item * per + (idx)] = v;
  } else {              // V[ci = token][co = channel]
    if (ci < ntok) v = src[(int64_t)ci * kKV + 128 + co];
    v_packed[(int64_t)item * per + (idx - per)] = v;
  }
}

}  // namespace
}  // namespace imf

using namespace imf;

extern "C" {

size_t imf_image_workspace_bytes(int B, int H, int W) {
  if (B < 1 || H < 8 || W < 8) return 0;
  return plan_of(shape_of(B, H, W), nullptr).total_bytes + 256;
}

int imf_image_tokens(int H, int W) {
  const ImgShape s = shape_of(1, H, W);
  return s.H8 * s.W8;
}

int imf_image_tables_build(int B, int H, int W, void *workspace, size_t workspace_bytes, void *stream) {
  IMF_REQUIRE(workspace && B >= 1 && B <= IMF_MAX_BATCH && H >= 8 && W >= 8, "imf_image_tables_build: bad argument");
  IMF_REQUIRE(((uintptr_t)workspace & 255) == 0, "imf_image_tables_build: workspace must be 256-byte aligned");
  IMF_REQUIRE(workspace_bytes >= imf_image_workspace_bytes(B, H, W), "imf_image_tables_build: workspace too small");
  const ImgShape s = shape_of(B, H, W);
  const Plan p = plan_of(s, workspace);
  hipStream_t st = (hipStream_t)stream;
  IMF_CHECK_HIP(hipMemsetAsync(workspace, 0, p.table_bytes, st));
  auto build = [&](const Table &t, int Hin, int Win, int Hout, int Wout, int ksize, int stride, int pad) {
    k_img_rulebook<<<(unsigned)div_up(t.n_slots * t.kvol, 256), 256, 0, st>>>(
        s.B, Hin, Win, Hout, Wout, ksize, stride, pad, t.n_slots, t.tile_rows, t.nbr, t.tile_mask);
  };
  build(p.t4, s.H4, s.W4, s.H4, s.W4, 3, 1, 1);
  build(p.t48, s.H4, s.W4, s.H8, s.W8, 3, 2, 1);
  build(p.td, s.H4, s.W4, s.H8, s.W8, 1, 2, 0);
  build(p.t8, s.H8, s.W8, s.H8, s.W8, 3, 1, 1);
  IMF_CHECK_LAUNCH("k_img_rulebook");
  return IMF_OK;
}

int imf_image_branch(const imf_image_desc *net, const float *image, int B, int H, int W, void *workspace,
                     size_t workspace_bytes, float *feat_out, float *kt_packed, float *v_packed,
                     int tokens_padded, int32_t *flags, void *stream) {
  IMF_REQUIRE(net && image && workspace, "imf_image_branch: null pointer");
  IMF_REQUIRE(B >= 1 && B <= IMF_MAX_BATCH && H >= 8 && W >= 8, "imf_image_branch: B=%d H=%d W=%d", B, H, W);
  IMF_REQUIRE(((uintptr_t)workspace & 255) == 0, "imf_image_branch: workspace must be 256-byte aligned");
  IMF_REQUIRE(workspace_bytes >= imf_image_workspace_bytes(B, H, W), "imf_image_branch: workspace too small");
  IMF_REQUIRE(net->stem_w && net->stem_scale && net->stem_shift, "imf_image_branch: stem weights missing");
  for (int i = 0; i < 15; ++i)
    IMF_REQUIRE(net->conv[i].w_packed && net->conv[i].scale && net->conv[i].shift, "imf_image_branch: conv %d missing", i);
  const ImgShape s = shape_of(B, H, W);
  const int ntok = s.H8 * s.W8;
  const bool want_kv = kt_packed || v_packed;
  if (want_kv) {
    IMF_REQUIRE(kt_packed && v_packed && net->ln_g && net->ln_b && net->kv_w, "imf_image_branch: K/V outputs need ln / kv weights");
    IMF_REQUIRE(tokens_padded % 64 == 0 && tokens_padded >= ntok, "imf_image_branch: tokens_padded=%d for %d tokens",
                tokens_padded, ntok);
  }
  const Plan p = plan_of(s, workspace);
  hipStream_t st = (hipStream_t)stream;

  auto conv = [&](const imf_net_conv &c, const Table *t, int64_t n_rows, const float *in, const float *residual,
                  float *out) -> int {
    imf_conv_args a;
    memset(&a, 0, sizeof(a));
    a.in_a = in; a.c_a = c.cin; a.w_packed = c.w_packed; a.kvol = c.kvol; a.cout = c.cout;
    if (t) {
      IMF_REQUIRE(t->kvol == c.kvol, "imf_image_branch: kernel volume %d on a %d-tap table", c.kvol, t->kvol);
      a.tile_rows = t->tile_rows; a.nbr = t->nbr; a.tile_mask = t->tile_mask;
      a.n_slots = t->n_slots; a.n_out = t->n_out;
    } else {
      a.n_slots = imf_rulebook_slots(n_rows); a.n_out = n_rows;
    }
    a.scale = c.scale; a.shift = c.shift; a.residual = residual; a.relu = c.relu; a.out = out;
    a.variant = c.variant; a.split_k = 0;
    a.dyn_err = c.variant == 6 ? flags : nullptr;
    a.kernel_tag = 1;
    // 3x3 convolutions of the trunk: the wave-split kernel (one workgroup per tile, the 9 taps x cin/32 sub-stages split
    // over its wavefronts, no split-K partials and no reduce launch).  Measured on the pair's 2 x 120 x 160 images (38 and
    // 10 tiles): branch alone 229 -> 182 us, pair step 1.030 -> 1.018 ms on the same box (14 launches fewer on a chain that
    // reaches the fusion's join last).  Always the 8-wavefront instance: the wavefront count is part of the arithmetic (it
    // cuts the sub-stage ranges, csrc/spconv_w.hip), so it is a static choice per layer -- never a function of the batch or
    // image size, or an image's features would differ in the last bits between batch sizes.  (This holds for the
    // IMAGE trunk only: the ResUNet's own layers choose their workgroup shape by batch size since round 5,
    // imf_resunet_conv_kernel_tag -- a fragment's descriptors agree between batch sizes to round-off, not bit for bit.)
    const int waves = c.kvol == 9 && (c.variant == 6 || c.variant == 0 || c.variant == 3) && c.cout % 64 == 0 ? 8 : 0;
    if (waves == 8) a.kernel_tag = 4 | 1;
    else if (waves == 4) a.kernel_tag = 8 | 1;
    if (waves) a.split_k = 1;
    a.workspace = p.splitk; a.workspace_bytes = p.splitk_bytes;
    return imf_spconv_fwd(&a, st);
  };

  // stem: conv7x7/2 + bn + relu (im2col + pointwise), 3x3/2 max pool
  {
    const int64_t total = s.n2 * kStemK;
    k_img_im2col7<<<(unsigned)div_up(total, 256), 256, 0, st>>>(image, B, H, W, s.H2, s.W2, p.im2col, net->variant == 6 ? flags : nullptr);
    IMF_CHECK_LAUNCH("k_img_im2col7");
    imf_net_conv c;
    memset(&c, 0, sizeof(c));
    c.w_packed = net->stem_w; c.kvol = 1; c.cin = kStemK; c.cout = kC1;
    c.scale = net->stem_scale; c.shift = net->stem_shift; c.relu = 1; c.variant = net->variant;
    int rc = conv(c, nullptr, s.n2, p.im2col, nullptr, p.stem);
    if (rc) return rc;
    const int64_t tot = s.n4 * (kC1 / 4);
    k_img_maxpool<<<(unsigned)div_up(tot, 256), 256, 0, st>>>(p.stem, B, s.H2, s.W2, s.H4, s.W4, kC1, p.b4[0]);
    IMF_CHECK_LAUNCH("k_img_maxpool");
  }
  int rc;
  // layer1: three BasicBlocks at 64 channels (model/resnet.py:35-72): x -> relu(bn(conv)) -> bn(conv) + x -> relu
  int cur = 0;
  for (int blk = 0; blk < 3; ++blk) {
    const int mid = (cur + 1) % 3, nxt = (cur + 2) % 3;
    if ((rc = conv(net->conv[2 * blk], &p.t4, s.n4, p.b4[cur], nullptr, p.b4[mid]))) return rc;
    if ((rc = conv(net->conv[2 * blk + 1], &p.t4, s.n4, p.b4[mid], p.b4[cur], p.b4[nxt]))) return rc;
    cur = nxt;
  }
  // layer2.0: stride-2 conv + 1x1 stride-2 projection of the identity; then three plain blocks at 128 channels
  if ((rc = conv(net->conv[6], &p.t48, s.n8, p.b4[cur], nullptr, p.b8[0]))) return rc;
  if ((rc = conv(net->conv[7], &p.td, s.n8, p.b4[cur], nullptr, p.b8[1]))) return rc;
  const bool last_is_out = feat_out != nullptr;
  if ((rc = conv(net->conv[8], &p.t8, s.n8, p.b8[0], p.b8[1], p.b8[2]))) return rc;
  int c8 = 2;
  for (int blk = 1; blk < 4; ++blk) {
    const int mid = (c8 + 1) % 4, nxt = (c8 + 2) % 4;
    float *dst = (blk == 3 && last_is_out) ? feat_out : p.b8[nxt];
    if ((rc = conv(net->conv[7 + 2 * blk], &p.t8, s.n8, p.b8[c8], nullptr, p.b8[mid]))) return rc;
    if ((rc = conv(net->conv[8 + 2 * blk], &p.t8, s.n8, p.b8[mid], p.b8[c8], dst))) return rc;
    c8 = nxt;
  }
  const float *feat = last_is_out ? feat_out : p.b8[c8];
  if (!want_kv) return IMF_OK;

  // context half of the cross attention: LayerNorm(tokens) -> to_kv (no bias) -> packed K^T / V per image
  k_img_layernorm128<<<(unsigned)div_up(s.n8, 4), 256, 0, st>>>(feat, s.n8, net->ln_g, net->ln_b, p.ln, net->variant == 6 ? flags : nullptr);
  IMF_CHECK_LAUNCH("k_img_layernorm128");
  {
    imf_net_conv c;
    memset(&c, 0, sizeof(c));
    c.w_packed = net->kv_w; c.kvol = 1; c.cin = kC2; c.cout = kKV; c.variant = net->variant;
    if ((rc = conv(c, nullptr, s.n8, p.ln, nullptr, p.kv))) return rc;
  }
  const int64_t per2 = 2ll * 128 * tokens_padded;
  k_img_pack_kv<<<dim3((unsigned)div_up(per2, 256), (unsigned)B), 256, 0, st>>>(p.kv, ntok, tokens_padded, kt_packed,
                                                                              v_packed);
  IMF_CHECK_LAUNCH("k_img_pack_kv");
  return IMF_OK;
}

}  // extern "C"
