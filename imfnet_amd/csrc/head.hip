// Pointwise head: conv1_tr (1x1x1, cat(c_a, c_b) -> 64) + norm1_tr + ReLU + final (1x1x1, 64 -> 32, bias) + L2
// normalisation as ONE launch (model/resunet.py:219-233).
//
// Why.  As two k_spconv_g launches the head cost 17 + 15 us per pair step plus a kernel boundary: every
// 64-row tile re-staged the 24 KiB / 8 KiB weight image of its layer through LDS (more bytes than the 24 KiB of rows it
// multiplies), and the [n, 64] intermediate (26 MB for the pair) made a round trip through HBM.  Both layers are
// row-local, so here a workgroup keeps a 16-row block per wavefront in registers from the gathered inputs to the
// normalised descriptor:
//   * persistent workgroups (two per CU) walk the tiles with stride gridDim.x: the conv1_tr weight image is copied to
//     LDS ONCE per workgroup, the `final` weights (8 B fragments) live in 32 VGPRs for the whole kernel;
//   * the rows of the NEXT tile arrive by LDS-DMA (k_spconv_g's lane-swizzled 1 KiB row images, csrc/spconv_g.hip) into
//     the wavefront's second buffer while the current tile is multiplied -- wavefront-private regions, no workgroup
//     barrier after the prologue, the only wait is the wavefront's own vmcnt;
//   * the hidden block (16 rows x 64 columns, BatchNorm + ReLU applied, range-checked for the split-f16 operands of
//     `final`) goes from accumulator layout to A-fragment layout through the wavefront's own, now free, row buffer;
//   * bias + L2 norm (no eps: resunet.py:230) in the second epilogue, 64-byte row segments stored.
// Arithmetic, operand layout and MFMA order per accumulator are k_spconv_g's (lo*hi, hi*lo, hi*hi per 32 channels,
// chunks ascending, the same epilogue expressions), the intermediate is rounded to fp32 exactly where the two-launch
// path stored it: the result is bit-identical to imf_spconv_fwd x 2 (tests/test_gpu_parity.py::test_pointwise_head).
#include "spconv_shared.h"

namespace imf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

namespace {

__device__ __forceinline__ void hd_split8(const float4 &x0, const float4 &x1, f16x8 &hi, f16x8 &lo) {
#ifdef IMF_NOSPLIT_ABL   // timing experiment only (wrong results): what the conversion costs
  hi = __builtin_bit_cast(f16x8, x0); lo = __builtin_bit_cast(f16x8, x1);
  return;
#endif
  const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const _Float16 h = (_Float16)v[t];
    hi[t] = h;
    lo[t] = (_Float16)(v[t] - (float)h);
  }
}

// fragment reads behind __restrict__ parameters (alias-scope metadata): see spconv_g.hip
__device__ __forceinline__ float4 hd_lds16(const float4 *__restrict__ src) { return *src; }
__device__ __forceinline__ f16x8 hd_lds_f16x8(const float4 *__restrict__ src) {
  return *reinterpret_cast<const f16x8 *>(src);
}

struct HeadParams {
  const float *in_a, *in_b;
  int c_a, c_b;
  const float *w1, *scale1, *shift1;
  int relu1;
  const float *w2, *scale2, *shift2;
  int l2norm;
  long long n;
  const int32_t *n_dev;
  float *out;
  int32_t *err;
};

}  // namespace

// NCC = (c_a + c_b) / 32 input chunks; hidden width 64, output width 32.
// PRE: the two sources are split-f16 operand images written by their producers (imf_head_args.a_split).
template <int NCC, bool PRE = false>
__global__ void __launch_bounds__(256, NCC <= 3 ? 2 : 1)
k_pointwise_head(const HeadParams p) {
  constexpr int W1_F4 = NCC * 512;                   // conv1_tr image: NCC sub-stages of 8 KiB
  constexpr int AW_F4 = NCC * 128;                   // one wavefront's 16 rows x NCC x 128 B
  __shared__ float4 smem[W1_F4 + 2 * 4 * AW_F4];
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  long long n = p.n;
  if (p.n_dev) {
    const long long nd = *p.n_dev;
    n = nd < n ? nd : n;
  }
  const long long n_tiles = (n + IMF_TILE_ROWS - 1) / IMF_TILE_ROWS;
  long long tile = blockIdx.x;
  if (tile >= n_tiles) return;
  const int ncc_a = p.c_a >> 5;

  // buffer windows end at the last row: rows beyond it read as zeros (their results are never stored)
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w1), (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.in_a), (short)0, (int)(n * p.c_a * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.in_b ? p.in_b : p.in_a), (short)0, (int)(n * (p.in_b ? p.c_b : p.c_a) * 4), 0x00020000);
  const unsigned stride_a = (unsigned)p.c_a * 4u, stride_b = (unsigned)p.c_b * 4u;
  // writer role of the lane in a 16-row block's gather (spconv_g.hip): row lane >> 2, piece (lane & 3) ^ f(row >> 2)
  const int row_w = lane >> 2;
  const unsigned wr_byte = 16u * (unsigned)((lane & 3) ^ ((4 - (row_w >> 2)) & 3));
  // reader role: MFMA A fragment, row r16, pieces q4 and 4 + q4
  const int rd_slot = 4 * r16 + (q4 ^ ((4 - (r16 >> 2)) & 3));
  float4 *const abase = smem + W1_F4 + wave * (2 * AW_F4);   // the wavefront's two row buffers

#define IMF_HD_DMA(t, buf)                                                                                          \
  {                                                                                                                 \
    const unsigned row_ = (unsigned)((t) * IMF_TILE_ROWS + wave * 16 + row_w);                                      \
    float4 *const ab_ = abase + (buf) * AW_F4;                                                                      \
    _Pragma("unroll") for (int cc_ = 0; cc_ < NCC; ++cc_) {                                                         \
      const bool second_ = cc_ >= ncc_a;                                                                            \
      const unsigned voff_ = row_ * (second_ ? stride_b : stride_a) + wr_byte;                                      \
      const unsigned soff_ = (unsigned)(second_ ? cc_ - ncc_a : cc_) << 7;                                          \
      const __amdgpu_buffer_rsrc_t rs_ = second_ ? rs_b : rs_a;                                                     \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_void *)(ab_ + 128 * cc_), 16, voff_, soff_, 0, 0);         \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_void *)(ab_ + 128 * cc_ + 64), 16, voff_ + 64u, soff_, 0, 0); \
    }                                                                                                               \
  }

  // prologue: the first tile's rows, the conv1_tr weights (verbatim copy of the packed image), `final` in registers
  IMF_HD_DMA(tile, 0)
#pragma unroll
  for (int j = 0; j < 2 * NCC; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void *)(smem + j * 256 + wave * 64), 16,
                                             (unsigned)tid * 16u + (unsigned)j * 4096u, 0, 0, 0);
  f16x8 w2h[2][2], w2l[2][2];                         // [chunk][column block]
  {
    const float4 *const w2 = reinterpret_cast<const float4 *>(p.w2);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        w2h[cc][cb] = __builtin_bit_cast(f16x8, w2[(cc * 4 + 2 * cb) * 64 + lane]);
        w2l[cc][cb] = __builtin_bit_cast(f16x8, w2[(cc * 4 + 2 * cb + 1) * 64 + lane]);
      }
  }
  const float un1 = p.w1[(long long)(p.c_a + p.c_b) * 64 + 1], un2 = p.w2[64 * 32 + 1];
  float sc1[4], sh1[4], sc2[2], sh2[2];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    sc1[cb] = p.scale1 ? p.scale1[cb * 16 + r16] : 1.f;
    sh1[cb] = p.shift1 ? p.shift1[cb * 16 + r16] : 0.f;
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    sc2[cb] = p.scale2 ? p.scale2[cb * 16 + r16] : 1.f;
    sh2[cb] = p.shift2 ? p.shift2[cb * 16 + r16] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                    // the weight image is complete for every wavefront

  int cur = 0;
  bool bad = false;
#pragma unroll 1
  for (; tile < n_tiles; tile += gridDim.x) {
    // this tile's rows have landed (the wavefront's own DMAs and the previous tile's stores)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long next = tile + gridDim.x;
    if (next < n_tiles) IMF_HD_DMA(next, cur ^ 1)     // lands under this tile's arithmetic
    float4 *const abuf = abase + cur * AW_F4;
    cur ^= 1;

    // ---- conv1_tr: [16, 32 NCC] x [32 NCC, 64] ----
    f32x4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc) {
      f16x8 ah, al;
      if (PRE) {
        ah = __builtin_bit_cast(f16x8, hd_lds16(&abuf[128 * cc + rd_slot]));
        al = __builtin_bit_cast(f16x8, hd_lds16(&abuf[128 * cc + 64 + rd_slot]));
      } else {
        hd_split8(hd_lds16(&abuf[128 * cc + rd_slot]), hd_lds16(&abuf[128 * cc + 64 + rd_slot]), ah, al);
      }
      const float4 *const wbuf = smem + cc * 512;
      f16x8 bh[4], bl[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        bh[cb] = hd_lds_f16x8(&wbuf[(2 * cb) * 64 + lane]);
        bl[cb] = hd_lds_f16x8(&wbuf[(2 * cb + 1) * 64 + lane]);
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[cb], acc[cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[cb], acc[cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[cb], acc[cb], 0, 0, 0);
    }
    // ---- epilogue 1 (conv_epilogue's expressions) -> the hidden block in A-fragment layout, in the free row buffer:
    // element (row, col) of chunk c = col >> 5 at float4 slot 128 c + 64 half + 4 row + (q ^ f(row >> 2)), piece = 4 half + q
    const long long row0 = tile * IMF_TILE_ROWS + wave * 16 + q4 * 4;
    float *const hbuf = reinterpret_cast<float *>(abuf);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int col32 = (cb & 1) * 16 + r16, piece = col32 >> 2;
      const int chunk = cb >> 1, half = piece >> 2, q = piece & 3;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = (acc[cb][r] * un1) * sc1[cb] + sh1[cb];
        if (p.relu1) x = fmaxf(x, 0.f);
        bad |= row0 + r < n && out_of_f16_range(x);
        const int row = q4 * 4 + r;
        hbuf[(128 * chunk + 64 * half + 4 * row + (q ^ ((4 - (row >> 2)) & 3))) * 4 + (col32 & 3)] = x;
      }
    }
    // ---- final: [16, 64] x [64, 32] ----
    f32x4 acc2[2];
    acc2[0] = acc2[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      f16x8 ah, al;
      hd_split8(hd_lds16(&abuf[128 * cc + rd_slot]), hd_lds16(&abuf[128 * cc + 64 + rd_slot]), ah, al);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, w2h[cc][cb], acc2[cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, w2l[cc][cb], acc2[cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, w2h[cc][cb], acc2[cb], 0, 0, 0);
    }
    // ---- epilogue 2: bias, L2 norm over the row's 32 columns (16 lanes x 2 blocks), store ----
    float v[2][4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[cb][r] = (acc2[cb][r] * un2) * sc2[cb] + sh2[cb];
    if (p.l2norm) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float ss = 0.f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) ss += v[cb][r] * v[cb][r];
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        ss += __shfl_xor(ss, 4, 64);
        ss += __shfl_xor(ss, 8, 64);
        const float nrm = sqrtf(ss);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) v[cb][r] = v[cb][r] / nrm;   // no eps: resunet.py:230
      }
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (row0 + r < n) p.out[(row0 + r) * 32 + cb * 16 + r16] = v[cb][r];
  }
#undef IMF_HD_DMA
  if (p.err && __ballot(bad) != 0ull && lane == 0) atomicOr(p.err, 32);
}

// bf16x3 (variant 3, round 5): the same head on fp32-exact operands.  fp32 rows in, the conv1_tr image of
// imf_pack_weights_bf16x3 (12 KiB per 32 input channels) once per workgroup in LDS, `final`'s image (12 fragments) in 48
// VGPRs, rows and the hidden block split into three bf16 parts in registers, six MFMAs per 32 channels in k_spconv_g's
// order (IMF_B3_TERMS per chunk, chunks ascending) -- bit-identical to the two k_spconv_g launches it replaces.  The image is
// 1.5x the split-f16 one, so the workgroup is EIGHT wavefronts sharing it (36 + 96 KiB of LDS for 96 input channels, one
// workgroup per CU, two wavefronts per SIMD); a workgroup step covers 128 rows.  No range guard (nothing to guard).
template <int NCC>
__global__ void __launch_bounds__(512, 1)
k_pointwise_head_b3(const HeadParams p) {
  constexpr int NW = 8;
  constexpr int W1_F4 = NCC * 768;                   // conv1_tr image: NCC sub-stages of 12 KiB
  constexpr int AW_F4 = NCC * 128;                   // one wavefront's 16 rows x NCC x 128 B
  __shared__ float4 smem[W1_F4 + 2 * NW * AW_F4];
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  long long n = p.n;
  if (p.n_dev) {
    const long long nd = *p.n_dev;
    n = nd < n ? nd : n;
  }
  constexpr int STEP_ROWS = 16 * NW;
  const long long n_steps = (n + STEP_ROWS - 1) / STEP_ROWS;
  long long step = blockIdx.x;
  if (step >= n_steps) return;
  const int ncc_a = p.c_a >> 5;

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w1), (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.in_a), (short)0, (int)(n * p.c_a * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.in_b ? p.in_b : p.in_a), (short)0, (int)(n * (p.in_b ? p.c_b : p.c_a) * 4), 0x00020000);
  const unsigned stride_a = (unsigned)p.c_a * 4u, stride_b = (unsigned)p.c_b * 4u;
  const int row_w = lane >> 2;
  const unsigned wr_byte = 16u * (unsigned)((lane & 3) ^ ((4 - (row_w >> 2)) & 3));
  const int rd_slot = 4 * r16 + (q4 ^ ((4 - (r16 >> 2)) & 3));
  float4 *const abase = smem + W1_F4 + wave * (2 * AW_F4);   // the wavefront's two row buffers

#define IMF_HB_DMA(t, buf)                                                                                          \
  {                                                                                                                 \
    const unsigned row_ = (unsigned)((t) * STEP_ROWS + wave * 16 + row_w);                                          \
    float4 *const ab_ = abase + (buf) * AW_F4;                                                                      \
    _Pragma("unroll") for (int cc_ = 0; cc_ < NCC; ++cc_) {                                                         \
      const bool second_ = cc_ >= ncc_a;                                                                            \
      const unsigned voff_ = row_ * (second_ ? stride_b : stride_a) + wr_byte;                                      \
      const unsigned soff_ = (unsigned)(second_ ? cc_ - ncc_a : cc_) << 7;                                          \
      const __amdgpu_buffer_rsrc_t rs_ = second_ ? rs_b : rs_a;                                                     \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_void *)(ab_ + 128 * cc_), 16, voff_, soff_, 0, 0);         \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_void *)(ab_ + 128 * cc_ + 64), 16, voff_ + 64u, soff_, 0, 0); \
    }                                                                                                               \
  }

  IMF_HB_DMA(step, 0)
  // the conv1_tr image, verbatim: NCC x 12 one-KiB pieces over the 8 wavefronts (piece j of the image at float4 64 j)
  for (int j = wave; j < 12 * NCC; j += NW)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void *)(smem + j * 64), 16, (unsigned)lane * 16u, (unsigned)j * 1024u, 0, 0);
  bf16x8 w2[2][2][3];                                 // `final`: [chunk][column block][part]
  {
    const float4 *const w2g = reinterpret_cast<const float4 *>(p.w2);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int h = 0; h < 3; ++h) w2[cc][cb][h] = __builtin_bit_cast(bf16x8, w2g[cc * 384 + (3 * cb + h) * 64 + lane]);
  }
  float sc1[4], sh1[4], sc2[2], sh2[2];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    sc1[cb] = p.scale1 ? p.scale1[cb * 16 + r16] : 1.f;
    sh1[cb] = p.shift1 ? p.shift1[cb * 16 + r16] : 0.f;
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    sc2[cb] = p.scale2 ? p.scale2[cb * 16 + r16] : 1.f;
    sh2[cb] = p.shift2 ? p.shift2[cb * 16 + r16] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                    // the weight image is complete for every wavefront

  int cur = 0;
#pragma unroll 1
  for (; step < n_steps; step += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this step's rows have landed (and the previous step's stores)
    const long long next = step + gridDim.x;
    if (next < n_steps) IMF_HB_DMA(next, cur ^ 1)     // lands under this step's arithmetic
    float4 *const abuf = abase + cur * AW_F4;
    cur ^= 1;

    // ---- conv1_tr: [16, 32 NCC] x [32 NCC, 64] ----
    f32x4 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc) {
      bf16x8 ap[3], bp[4][3];
      split_b3(hd_lds16(&abuf[128 * cc + rd_slot]), hd_lds16(&abuf[128 * cc + 64 + rd_slot]), ap[0], ap[1], ap[2]);
      const float4 *const wbuf = smem + cc * 768;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int h = 0; h < 3; ++h) bp[cb][h] = __builtin_bit_cast(bf16x8, hd_lds16(&wbuf[(3 * cb + h) * 64 + lane]));
#define IMF_HB_TERM(I, J)                                                                              \
  _Pragma("unroll") for (int cb = 0; cb < 4; ++cb)                                                     \
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[I], bp[cb][J], acc[cb], 0, 0, 0);
      IMF_B3_TERMS(IMF_HB_TERM)
#undef IMF_HB_TERM
    }
    // ---- epilogue 1 (conv_epilogue's expressions) -> the hidden block in A-fragment layout, in the free row buffer ----
    const long long row0 = step * STEP_ROWS + wave * 16 + q4 * 4;
    float *const hbuf = reinterpret_cast<float *>(abuf);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int col32 = (cb & 1) * 16 + r16, piece = col32 >> 2;
      const int chunk = cb >> 1, half = piece >> 2, q = piece & 3;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = (acc[cb][r] * 1.f) * sc1[cb] + sh1[cb];
        if (p.relu1) x = fmaxf(x, 0.f);
        const int row = q4 * 4 + r;
        hbuf[(128 * chunk + 64 * half + 4 * row + (q ^ ((4 - (row >> 2)) & 3))) * 4 + (col32 & 3)] = x;
      }
    }
    // ---- final: [16, 64] x [64, 32] ----
    f32x4 acc2[2];
    acc2[0] = acc2[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      bf16x8 ap[3];
      split_b3(hd_lds16(&abuf[128 * cc + rd_slot]), hd_lds16(&abuf[128 * cc + 64 + rd_slot]), ap[0], ap[1], ap[2]);
#define IMF_HB_TERM(I, J)                                                                              \
  _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)                                                     \
      acc2[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[I], w2[cc][cb][J], acc2[cb], 0, 0, 0);
      IMF_B3_TERMS(IMF_HB_TERM)
#undef IMF_HB_TERM
    }
    // ---- epilogue 2: bias, L2 norm over the row's 32 columns (16 lanes x 2 blocks), store ----
    float v[2][4];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[cb][r] = (acc2[cb][r] * 1.f) * sc2[cb] + sh2[cb];
    if (p.l2norm) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float ss = 0.f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) ss += v[cb][r] * v[cb][r];
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        ss += __shfl_xor(ss, 4, 64);
        ss += __shfl_xor(ss, 8, 64);
        const float nrm = sqrtf(ss);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) v[cb][r] = v[cb][r] / nrm;   // no eps: resunet.py:230
      }
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (row0 + r < n) p.out[(row0 + r) * 32 + cb * 16 + r16] = v[cb][r];
  }
#undef IMF_HB_DMA
}

}  // namespace imf

using namespace imf;

extern "C" {

int imf_pointwise_head(const imf_head_args *a, void *stream) {
  IMF_REQUIRE(a, "imf_pointwise_head: null args");
  IMF_REQUIRE(a->in_a && a->w1_packed && a->w2_packed && a->out, "imf_pointwise_head: null pointer");
  IMF_REQUIRE(a->c_a > 0 && a->c_a % 32 == 0 && a->c_b >= 0 && a->c_b % 32 == 0 && (a->c_b == 0) == (a->in_b == nullptr),
              "imf_pointwise_head: c_a=%d c_b=%d must be multiples of 32 (c_b == 0 <=> in_b == NULL)", a->c_a, a->c_b);
  const int ncc = (a->c_a + a->c_b) / 32;
  IMF_REQUIRE(ncc >= 2 && ncc <= 4, "imf_pointwise_head: %d input channels (64 .. 128 supported)", 32 * ncc);
  IMF_REQUIRE(a->c_mid == 64 && a->c_out == 32, "imf_pointwise_head: hidden width %d / output width %d (64 / 32 supported)",
              a->c_mid, a->c_out);
  IMF_REQUIRE(a->n > 0 && a->n * (int64_t)(a->c_a > a->c_b ? a->c_a : a->c_b) * 4 < (1ll << 31),
              "imf_pointwise_head: n=%lld (inputs must stay below 2 GiB each: raw-buffer addressing)", (long long)a->n);
  HeadParams p{a->in_a, a->in_b, a->c_a, a->c_b, a->w1_packed, a->scale1, a->shift1, a->relu1,
               a->w2_packed, a->scale2, a->shift2, a->l2norm, (long long)a->n, a->n_dev, a->out, a->flags};
  hipStream_t st3 = (hipStream_t)stream;
  if (a->variant == 3) {   // bf16x3 images, fp32 rows
    IMF_REQUIRE(!a->a_split && ncc <= 3, "imf_pointwise_head: variant 3 takes fp32 rows and 64 or 96 input channels (%d, a_split %d)",
                32 * ncc, a->a_split);
    const long long steps = div_up(a->n, 128);
    const unsigned grid3 = (unsigned)(steps < 256 ? steps : 256);   // one resident 8-wavefront workgroup per CU
    if (a->ev_begin) IMF_CHECK_HIP(hipEventRecord((hipEvent_t)a->ev_begin, st3));
    if (ncc == 2) k_pointwise_head_b3<2><<<grid3, 512, 0, st3>>>(p); else k_pointwise_head_b3<3><<<grid3, 512, 0, st3>>>(p);
    IMF_CHECK_LAUNCH("k_pointwise_head_b3");
    if (a->ev_end) IMF_CHECK_HIP(hipEventRecord((hipEvent_t)a->ev_end, st3));
    return IMF_OK;
  }
  IMF_REQUIRE(a->variant == 0 || a->variant == 6, "imf_pointwise_head: variant %d (6 = split-f16 images, 3 = bf16x3 images)", a->variant);
  const long long tiles = div_up(a->n, IMF_TILE_ROWS);
  const unsigned grid = (unsigned)(tiles < 512 ? tiles : 512);   // two resident workgroups per CU walk the tiles
  hipStream_t st = (hipStream_t)stream;
  if (a->ev_begin) IMF_CHECK_HIP(hipEventRecord((hipEvent_t)a->ev_begin, st));
  switch (ncc) {
    case 2: if (a->a_split) k_pointwise_head<2, true><<<grid, 256, 0, st>>>(p); else k_pointwise_head<2><<<grid, 256, 0, st>>>(p); break;
    case 3: if (a->a_split) k_pointwise_head<3, true><<<grid, 256, 0, st>>>(p); else k_pointwise_head<3><<<grid, 256, 0, st>>>(p); break;
    default: if (a->a_split) k_pointwise_head<4, true><<<grid, 256, 0, st>>>(p); else k_pointwise_head<4><<<grid, 256, 0, st>>>(p); break;
  }
  IMF_CHECK_LAUNCH("k_pointwise_head");
  if (a->ev_end) IMF_CHECK_HIP(hipEventRecord((hipEvent_t)a->ev_end, st));
  return IMF_OK;
}

}  // extern "C"
