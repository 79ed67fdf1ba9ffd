// Sparse convolution, variant 6, third implementation: the "wave-split" kernel for the levels that cannot fill the
// chip with 64-row tiles (tensor stride >= 2: a few hundred tiles or fewer).
//
// Why (round 3).  k_spconv_g gives such a level its parallelism by splitting a tile's kernel offsets over up to eight
// WORKGROUPS (split-K): every partition pays the launch prologue, writes a 64 x 64 slab of raw partial sums to HBM, and
// a second launch (k_spconv_reduce) adds the slabs and applies the epilogue.  Measured on the pair's stride-4 / 8
// levels (profiles/r02_kernel_stats.txt): 19-36 us per convolution + 6-8 us of reduce + a kernel boundary, of which
// the data path is < 15 % (profiles/r02_conv_dma_ablations.txt) -- and inside a partition the four wavefronts share
// the weight block of a sub-stage, so each sub-stage costs a workgroup barrier for 12 MFMAs per wavefront and every
// wavefront re-reads the whole 8 KiB B block from LDS.
//
// Here ONE workgroup owns a (64-row tile, 64-column slab) for ALL of its kernel offsets, and the split runs over its
// W wavefronts instead: the tile's sub-stage list (active offset x 32-channel chunk, ascending) is cut into W
// contiguous ranges, wavefront w walks range w for all 64 rows x 64 columns (16 accumulators).  Consequences:
//   * a wavefront's operands are private: its own 8 KiB row image (64 gathered rows x 128 B) and 8 KiB weight block per
//     sub-stage, fetched by LDS-DMA into its own 16 KiB of LDS -- no workgroup barrier in the main loop, only the
//     wavefront's own `s_waitcnt vmcnt(0)`;
//   * one read of the B fragments feeds four row blocks: 16 ds_read_b128 per 48 MFMAs instead of 10 per 12;
//   * the LDS buffer is single: once the 16 fragments of sub-stage t sit in registers the DMA of t + 1 is issued into the
//     same 16 KiB and lands under the 48 MFMAs of t (registers are the second buffer);
//   * the W partial tiles meet in LDS (the staging area, reused), are added in wavefront order by all threads, and the
//     epilogue (BatchNorm scale / shift, residual, ReLU, range flag, L2 norm) runs in the same launch: no partial sums
//     in HBM, no reduce launch, no kernel boundary.
// The partition depends on the tile's own active-offset list and on W only -- not on the row count, the grid or the
// capacity -- so a tile's sums are the same in every launch that contains it (exact mode == capacity mode bit for bit
// with no device-side split rule).  W is part of the arithmetic (the ranges), so it is the CALLER's static choice
// (imf_conv_args.kernel_tag), never a function of the row count.
//
// Operand layout, MFMA sequence per (row block, column block, sub-stage) and the weight image are k_spconv_g's
// (csrc/spconv_g.hip): DMA row images with the conflict-free lane swizzle, `lo*hi, hi*lo, hi*hi` per 32 channels.
#include "spconv_shared.h"

#ifndef IMF_W_ABL
#define IMF_W_ABL 0   // timing experiments only (wrong results; tools/w_ablations.sh): 1 no main loop, 2 no neighbour-table loads,
                      // 4 no combine / epilogue, 8 leave right after the tile test (launch + dispatch only); bf16x3 loop (round 6):
                      // 16 no weight loads in the loop (stale registers), 32 no row DMAs (stale LDS), 64 no MFMAs, 128 no split
                      // (parts = raw bits), 256 no LDS fragment reads
#endif

#ifndef IMF_W_EXP
#define IMF_W_EXP 0   // experiment (OCC 3 whole tiles): 2 = half B of t requested behind the split (no gain: LAB_NOTES 4g-8)
#endif

namespace imf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

namespace {

constexpr int kDummyJkW = kKCache - 1;        // neighbour-table row that is always "no input"
constexpr unsigned kNoRowW = 0x00FFFFFFu;     // 24-bit row index whose byte offset falls outside the buffer window

__device__ __forceinline__ void w_split8(const float4 &x0, const float4 &x1, f16x8 &hi, f16x8 &lo) {
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
__device__ __forceinline__ float4 w_lds16(const float4 *__restrict__ src) { return *src; }
__device__ __forceinline__ f16x8 w_lds_f16x8(const float4 *__restrict__ src) {
  return *reinterpret_cast<const f16x8 *>(src);
}
__device__ __forceinline__ unsigned w_lds_u32(const unsigned *__restrict__ src) { return *src; }

}  // namespace

// AR (spconv_shared.h): the arithmetic of the main loop.
//   kArF16x2      fp32 rows split into f16 hi + lo in registers, 3 x v_mfma_f32_16x16x32_f16 per 32 channels (variant 6)
//   kArF16x2Pre   the same products; the input rows are split-f16 operand images (ConvParams::a_split) -- no conversion
//   kArF32        fp32 rows and the fp32 weight image (imf_pack_weights) straight into 8 x v_mfma_f32_16x16x4_f32 per 32
//                 channels (variant 0: the reference's arithmetic).  Same DMA pieces, same LDS images, no conversion at all:
//                 lane (r16, q4) reads channels {4 q4 .. + 3} and {16 + 4 q4 .. + 3} of its row as two float4 -- the A
//                 operands of the 8 k-steps -- and the image's [j][cb][lane] quads are the matching B operands.
//   kArBf16x3     fp32 rows split into three bf16 parts in registers (exact), the bf16x3 weight image (12 KiB per sub-stage),
//                 6 x v_mfma_f32_16x16x32_bf16 per 32 channels (variant 3).  The wavefront's weight region stays 8 KiB: a
//                 sub-stage's weights arrive in two halves of 6 KiB (column blocks 0-1, then 2-3), each landing under the
//                 48 MFMAs of the other; the rows of sub-stage t + 1 are requested as soon as those of t sit in registers.
//                 (Measured and dropped, round 5: the MFMAs in row-block-major order with the split of block b + 1 placed
//                 between the MFMAs of block b -- the compiler interleaves them, the times do not move: 509 vs 515 us
//                 over the network's wave-split shapes, step 1.295 vs 1.297 ms; the split's VALU work is hidden by the
//                 SIMD's other wavefront already.  What the split costs is the g kernel's: ablated, step -8.5 %.)
// USE: profiling label only (the identical kernel under a second symbol; 1 = the image trunk's dense 3 x 3 convolutions, so that
// per-kernel statistics keep them apart from the ResUNet's launches -- imf_conv_args.kernel_tag bit 0, as k_spconv_g's USE).
// RB (round 5): 16-row blocks per workgroup, 4 = a whole 64-row tile, 2 = HALF a tile (rows 32 h .. 32 h + 31 of tile u / 2,
// unit u = launch index).  The stride-8 level of a fragment pair has 34 tiles x 4 slabs = 136 workgroups for 256 CUs; as
// 272 half-tile workgroups of 4 wavefronts (40 KiB of LDS: up to three per CU) every CU works.  The rulebook is untouched --
// a half tile walks its tile's offset list (the tile's mask is a superset of the half's) -- and a row's sums depend on
// (W, the tile's offset list) only, as before.
// (RB 1, quarter tiles of 4 wavefronts -- 94 VGPRs, 24 KiB -- was built and measured too: slower than half tiles for a single
// fragment on every level (sum of the wave-split shapes 304 -> 379 us, forward 0.867 -> 0.912 ms) and than 48-row units
// for a pair's stride-8 level (52 -> 84 us): each workgroup streams its slab's whole weight image for 16 rows.)
// RB 3 (every arithmetic) = 48-row UNITS that ignore the tile boundaries (unit u = slots 48 u .. 48 u + 47; it walks the union of the offset
// lists of the one or two tiles it touches): 2 176 rows are 46 units instead of 34 tiles, x 4 slabs = 184 workgroups with
// 3 / 4 of a tile's work each.
// OCC (round 6, 4 wavefronts of bf16x3): the register budget in wavefronts per SIMD.  48-row units fit three at 145 VGPRs; whole
// tiles need 168 for it, reached (without spills in the loop) by reading the row indices of t + 1 at the head of t instead of
// between the two MFMA groups.  A separate symbol: where the side streams' kernels run beside a layer, two wavefronts per SIMD
// and the CU space they leave measured better.
template <bool CAT, int W, int AR = kArF16x2, int USE = 0, int RB = 4, int OCC = (RB == 2 || (RB == 3 && W == 4 && AR == kArBf16x3)) ? 3 : 2>
__global__ void __launch_bounds__(64 * W, OCC)
k_spconv_w(const ConvParams p) {
  static_assert(OCC == 2 || (OCC == 3 && RB == 2) || (AR == kArBf16x3 && ((W == 4 && OCC == 3) || RB == 2)), "three (half tiles: four) wavefronts per SIMD: the bf16x3 kernels");
  static_assert(RB == 4 || RB == 3 || (RB == 2 && AR == kArBf16x3), "half tiles: bf16x3 only (its weights need no LDS region)");
  constexpr int UR = 16 * RB;                        // rows (slots) per workgroup: the UNIT
  constexpr bool PRE = AR == kArF16x2Pre;
  constexpr unsigned SUB_BYTES = AR == kArBf16x3 ? 12288u : 8192u;   // weight image bytes per (offset, 32-channel) sub-stage
  constexpr int NT = 64 * W;
  // per wavefront: rows 512 float4 (4 blocks x 2 KiB) + weights 512.  bf16x3 keeps no weights in LDS: its region is the rows
  // (128 RB float4) or one PASS of the partial tile (PB row blocks, 256 float4 each), whichever is larger -- with the tile
  // combined two row blocks at a time a 4-wavefront workgroup of 48- or 64-row units needs 41 KiB: three per CU (round 6)
  constexpr int PB = AR == kArBf16x3 ? ((RB > 2 && W == 4) ? 2 : (OCC == 4 ? 1 : RB)) : RB;  // row blocks per pass of the combine (W 8: LDS is not what limits it)
  constexpr int REG_F4 = AR != kArBf16x3 ? 1024 : (128 * RB > 256 * PB ? 128 * RB : 256 * PB);
  constexpr int NBR_F4 = kKCache * IMF_TILE_ROWS / 4;
  constexpr int TAB_F4 = (kSubTab + 3) / 4;
  constexpr int KL_F4 = (kKCache + 3) / 4;
  __shared__ float4 smem[W * REG_F4 + NBR_F4 + TAB_F4 + KL_F4];
  unsigned *const nbr_lds = reinterpret_cast<unsigned *>(smem + W * REG_F4);              // [kKCache][64]
  unsigned *const stab = reinterpret_cast<unsigned *>(smem + W * REG_F4 + NBR_F4);        // [kSubTab]
  int *const klist = reinterpret_cast<int *>(smem + W * REG_F4 + NBR_F4 + TAB_F4);        // [kKCache]

  // XCD-aware (tile, slab) order (p.w_xcd): workgroups go to the 8 XCDs round-robin in launch order, so with the plain
  // (x = tile, y = slab) order every XCD's 4 MiB L2 sees every slab of the weight image (7 MB for 256 -> 256).  Here the
  // slab is a function of the XCD (launch index mod 8), each L2 then holds 1 / n_slabs of the weights.
  int tile = blockIdx.x, y = blockIdx.y;
  long long slots_act = p.n_slots;
  if (p.n_out_dev) slots_act = conv_slots(p, conv_rows(p));   // capacity mode: tiles beyond the actual rows leave
  const unsigned ns = gridDim.y;
  if (p.w_xcd == 2 && ns <= 8 && (ns & (ns - 1)) == 0) {
    // contiguous: the XCD's workgroups walk ONE range of consecutive tiles -- rows of neighbouring tiles are
    // neighbours in space, the XCD's L2 then serves a fraction of the input rows instead of all of them.  Ranges are cut
    // from the ACTUAL tiles; the launcher pads gridDim.x to a multiple of 8 so that every XCD has enough workgroups.
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned xcd = lin & 7u, j = lin >> 3;
    const unsigned groups = 8u / ns, t_act = (unsigned)((slots_act + UR - 1) / UR);   // (units)
    const unsigned chunk = (t_act + groups - 1) / groups;
    y = (int)(xcd % ns);
    if (j >= chunk) return;
    tile = (int)((xcd / ns) * chunk + j);
  } else if (p.w_xcd && ns > 1 && ns <= 8 && (ns & (ns - 1)) == 0) {
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned xcd = lin & 7u, j = lin >> 3;
    y = (int)(xcd % ns);
    tile = (int)(j * (8u / ns) + xcd / ns);
  }
  const long long row0 = (long long)tile * UR;       // (`tile` is the UNIT index up to here) first slot of this workgroup
  if (row0 >= slots_act) return;
  const long long last = row0 + UR - 1 < slots_act - 1 ? row0 + UR - 1 : slots_act - 1;
  tile = (int)(row0 / IMF_TILE_ROWS);
  const int tile_b = (int)(last / IMF_TILE_ROWS);    // != tile only for units that ignore the tile boundaries (RB 3)
  if (IMF_W_ABL & 8) return;
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cin = p.c_a + (CAT ? p.c_b : 0);
  const int ncc = cin / 32;

  uint32_t m = p.tile_mask ? p.tile_mask[tile * IMF_MASK_WORDS] : 1u;            // kvol == 1: offset 0, every tile
  if (RB == 3 && p.tile_mask && tile_b != tile) m |= p.tile_mask[tile_b * IMF_MASK_WORDS];
  const int nk = __builtin_popcount(m);
  if (nk == 0) return;                               // padding tile
  if (tid < 32 && ((m >> tid) & 1u)) klist[__builtin_popcount(m & ((1u << tid) - 1u))] = tid;
  __syncthreads();
  const int n_sub = nk * ncc;
  {   // the tile's slice of the neighbour table (24-bit row indices) and the sub-stage table; unconditional loads
    constexpr int JSTEP = NT / IMF_TILE_ROWS;        // offsets covered per pass of the workgroup: 8 / 4
    constexpr int kPer = (kKCache + JSTEP - 1) / JSTEP;
    const int srow = tid & 63, j0 = tid >> 6;
    const bool in_unit = srow < UR && row0 + srow < slots_act;      // (table rows beyond the unit: "no input")
    const long long slot = in_unit ? row0 + srow : row0;
    int v[kPer];
    if (p.nbr) {
      const int32_t *const src = p.nbr + slot;
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int j = j0 + JSTEP * i;
        if (IMF_W_ABL & 2) { v[i] = srow + j; continue; }
        v[i] = src[(long long)klist[j < nk ? j : 0] * p.n_slots];
      }
    } else {                                         // kvol == 1 (a pointwise layer): the slot's own row
#pragma unroll
      for (int i = 0; i < kPer; ++i) v[i] = row_of_slot(p, slot);
    }
    if (tid < kSubTab) {
      unsigned e = (unsigned)kDummyJkW << 9;
      if (tid < n_sub) {
        const int jk = tid / ncc, cc = tid - jk * ncc;
        const int ch0 = cc * 32;
        const bool second = CAT && ch0 >= p.c_a;
        const int cch = second ? (ch0 - p.c_a) >> 5 : cc;
        e = (unsigned)(klist[jk] * ncc + cc) | ((unsigned)jk << 9) | ((second ? 1u : 0u) << 14) | ((unsigned)cch << 15);
      }
      stab[tid] = e;
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int j = j0 + JSTEP * i;
      if (j < nk) nbr_lds[j * IMF_TILE_ROWS + srow] = (v[i] >= 0 && in_unit) ? (unsigned)v[i] : kNoRowW;
      else if (j == kDummyJkW) nbr_lds[j * IMF_TILE_ROWS + srow] = kNoRowW;
    }
  }
  __syncthreads();

  f32x4 acc[RB][4];
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[b][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.w_packed), (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.in_a), (short)0, 0x7FFFF000, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(CAT ? p.in_b : p.in_a), (short)0, 0x7FFFF000, 0x00020000);
  const unsigned stride_a = (unsigned)p.c_a * 4u, stride_b = (unsigned)(CAT ? p.c_b : p.c_a) * 4u;
  const unsigned wslab = (unsigned)((long long)y * p.kvol * ncc * SUB_BYTES);      // bytes (image < 2 GiB)
  const unsigned woff = (unsigned)lane * 16u;
  // writer role of the lane in a 16-row block's gather: row lane >> 2, piece (lane & 3) ^ f(row >> 2)   (spconv_g.hip)
  const int row_w = lane >> 2;
  const unsigned wr_byte = 16u * (unsigned)((lane & 3) ^ ((4 - (row_w >> 2)) & 3));
  // reader role: MFMA A fragment, row r16, pieces q4 and 4 + q4
  const int rd_slot = 4 * r16 + (q4 ^ ((4 - (r16 >> 2)) & 3));
  float4 *const areg = smem + wave * REG_F4;         // rows: block b at + 128 b (two 1 KiB images)
  float4 *const wreg = areg + 512;                   // weights: fragment (2 cb + {hi, lo}) at + 64 (2 cb + h)

  struct Rows { unsigned r[4]; };
#define IMF_W_ROWS(dst, e)                                                                                         \
  {                                                                                                                \
    const unsigned *const base_ = nbr_lds + ((((unsigned)(e)) >> 9) & 31u) * IMF_TILE_ROWS + row_w;                \
    _Pragma("unroll") for (int b_ = 0; b_ < RB; ++b_) (dst).r[b_] = w_lds_u32(base_ + 16 * b_);                    \
  }
  // LDS-DMA of one sub-stage into the wavefront's region: 8 KiB of weights verbatim, 64 rows x 128 B as 8 images
#define IMF_W_DMA(e, rows)                                                                                         \
  {                                                                                                                \
    const unsigned ee = (unsigned)(e);                                                                             \
    const unsigned wso = wslab + (ee & 511u) * SUB_BYTES;                                                          \
    _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                  \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void *)(wreg + 64 * j), 16, woff + 1024u * j, wso, 0, 0); \
    const bool second = CAT && ((ee >> 14) & 1u);                                                                  \
    const unsigned soff = (ee >> 15) << 7;                                                                         \
    const __amdgpu_buffer_rsrc_t rs = second ? rs_b : rs_a;                                                        \
    _Pragma("unroll") for (int b_ = 0; b_ < RB; ++b_) {                                                               \
      const unsigned voff = __umul24((rows).r[b_], second ? stride_b : stride_a) + wr_byte;                        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(areg + 128 * b_), 16, voff, soff, 0, 0);           \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(areg + 128 * b_ + 64), 16, voff + 64u, soff, 0, 0); \
    }                                                                                                              \
  }

  // bf16x3: the gathered rows of a sub-stage (8 pieces) and one 6 KiB half of its weights (6 pieces) as separate requests
#define IMF_W_DMA_ROWS(e, rows)                                                                                    \
  {                                                                                                                \
    const unsigned ee = (unsigned)(e);                                                                             \
    const bool second = CAT && ((ee >> 14) & 1u);                                                                  \
    const unsigned soff = (ee >> 15) << 7;                                                                         \
    const __amdgpu_buffer_rsrc_t rs = second ? rs_b : rs_a;                                                        \
    _Pragma("unroll") for (int b_ = 0; b_ < RB; ++b_) {                                                               \
      const unsigned voff = __umul24((rows).r[b_], second ? stride_b : stride_a) + wr_byte;                        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(areg + 128 * b_), 16, voff, soff, 0, 0);           \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(areg + 128 * b_ + 64), 16, voff + 64u, soff, 0, 0); \
    }                                                                                                              \
  }
  // ... and one 6 KiB half of its weights (column blocks 2 h, 2 h + 1) straight into REGISTERS: the image is in fragment
  // order and the block is this wavefront's alone, so the six 1 KiB pieces are six plain buffer loads -- no LDS-DMA piece
  // (~100 cycles of issue each in a phase that carries row pieces and fragment reads, MI355X_MICROARCH.md), no LDS write,
  // no ds_read, no hand-counted wait (the compiler waits for the registers).  Units of 3 / 4 row blocks carry the piece offset on
  // the scalar side (one address VGPR instead of six: 150 -> 145 / 182 -> 176 VGPRs); half tiles keep six (measured: 136 vs 139 us)
#define IMF_W_LD_WHALF(dst, e, h)                                                                                  \
  {                                                                                                                \
    const unsigned wso = wslab + ((unsigned)(e) & 511u) * SUB_BYTES + (unsigned)(h) * 6144u;                       \
    _Pragma("unroll") for (int cb_ = 0; cb_ < 2; ++cb_)                                                            \
        _Pragma("unroll") for (int h_ = 0; h_ < 3; ++h_)                                                           \
            (dst)[cb_][h_] = __builtin_bit_cast(bf16x8, (RB == 2 && OCC != 4)                                                 \
                ? __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff + 1024u * (unsigned)(3 * cb_ + h_), wso, 0)     \
                : __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff, wso + 1024u * (unsigned)(3 * cb_ + h_), 0));   \
  }

  // this wavefront's range of the tile's sub-stages
  const int t0 = (int)((long long)wave * n_sub / W), t1 = (IMF_W_ABL & 1) ? t0 : (int)((long long)(wave + 1) * n_sub / W);
  unsigned e_cur = 0, e_nxt = 0;
  Rows rows_nxt;
  if (t0 < t1) {
    e_cur = (unsigned)__builtin_amdgcn_readfirstlane((int)w_lds_u32(&stab[t0]));
    Rows rows0;
    IMF_W_ROWS(rows0, e_cur)
    if constexpr (AR == kArBf16x3) {
      IMF_W_DMA_ROWS(e_cur, rows0)
    } else {
      IMF_W_DMA(e_cur, rows0)
    }
    e_nxt = (unsigned)__builtin_amdgcn_readfirstlane((int)w_lds_u32(&stab[t0 + 1 < kSubTab ? t0 + 1 : kSubTab - 1]));
    IMF_W_ROWS(rows_nxt, e_nxt)
  }
  // A sub-stage of a wavefront is two segments, LOAD (wait for its DMAs, 16 fragment reads, issue the 16 DMA pieces of the
  // next sub-stage: ~1.0 k cycles of the SIMD's address path at ~64 cycles per 1 KiB piece) and COMPUTE (hi / lo split,
  // 48 MFMAs: ~0.8 k cycles of matrix pipe + ~0.4 k of VALU).  A wavefront issues in order, so the two never overlap
  // inside it; what overlaps is whatever the SIMD's other wavefront happens to be doing (counters of round 3,
  // profiles/r03_pmc_counters.txt: 38 % of the wavefront cycles are issue stalls, matrix pipe 27 % busy).  Two schedules
  // that tried to force the overlap were built on this loop, gave identical sums and were NOT faster (tools/conv_iso.py):
  //   * the 16 pieces issued one after every third MFMA instead of back to back: 128 -> 128 at 7.7 k rows 25.7 -> 26.5 us,
  //     64 -> 64 at 103 k rows (4 wavefronts) 89.7 -> 99.2 us -- a piece holds the wavefront wherever it stands;
  //   * ping-pong: the two halves of an 8-wavefront workgroup half a period apart, held by two workgroup barriers per
  //     sub-stage (SIMD partners w / w + 4 in opposite segments, MI355X_MICROARCH.md "Two waves per SIMD"): the
  //     stride-4 / 8 launches 35.2 -> 43.4 us on average -- with a single buffer per wavefront the DMAs issued at the end
  //     of LOAD get one COMPUTE segment to land, and every wavefront then waits for them in lock step.
  //   * one extra load per sub-stage that touches a line per lane of the weight block two sub-stages ahead (a software
  //     prefetch towards L2 / L1): the stride-4 / 8 launches 35.8 -> 38.3 us, the stride-2 ones 29.3 -> 34.6 us -- every
  //     additional vector-memory instruction costs the wavefront more than the shorter DMA latency returns.
  if constexpr (AR == kArBf16x3) {
    // Issue order per sub-stage t: WB(t) | [rows of t landed, 8 fragment reads] R(t + 1) | 48 MFMAs on half A | WA(t + 1) |
    // 48 MFMAs on half B.  R = 8 LDS-DMA pieces, WA / WB = 6 register loads each.  The loop body is branch-free ON PURPOSE:
    // the compiler's vmcnt for the register loads is the minimum over the paths that reach a use, so a request behind
    // `if (more)` makes it wait for everything younger as well (seen: vmcnt(6) / vmcnt(3) where 14 are allowed).  The
    // wavefront's last sub-stage therefore requests a next one too: rows that do not exist (no memory access, zeros into
    // the free row region) and a weight half nobody reads; both are waited for before the region is reused below.
    // (Measured and dropped: the gathered rows as register loads too -- lane (r16, q4) loading its two A-fragment pieces of
    // row 16 b + r16 straight from global, no LDS in the main loop at all, 210 VGPRs: bit-identical, the 8-wavefront shapes
    // 525-530 -> 542 us in sum, the 4-wavefront ones 512-518 -> 504 us, pair step 1.231-1.242 vs 1.237-1.243 ms: a wash.)
    // (Also dropped: both weight halves of t + 1 requested during sub-stage t -- four register sets, 206 VGPRs, one register
    // copy of 48 VGPRs per sub-stage: the wave-split shapes +4-5 % in sum, pair step 1.283-1.292 -> 1.335 ms.)
    // (Round 6, measured and dropped again, this time WITHOUT register copies: the whole weight block of t + 1 requested at the
    // head of t into a second register pair, loop unrolled by two, pairs alternating by code -- 226 VGPRs, bit-identical; whole
    // tiles 64 -> 64 at 103 k rows 147 -> 151 us, 8 wavefronts 151 -> 172, 48-row units 147 -> 178, pair step 1.204 -> 1.257 ms:
    // more requests in flight delay the row pieces more than the earlier weights help.  tools/experiments/
    // spconv_w_full_stage_prefetch.hip)
    bf16x8 bA[2][3], bB[2][3];
    if (t0 < t1) IMF_W_LD_WHALF(bA, e_cur, 0)
#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
      const bool more = t + 1 < t1;
      constexpr bool LATE_B = (IMF_W_EXP & 2) && RB == 4 && W == 4;
      constexpr bool HEAD_ROWS = RB == 4 && OCC == 3;
      if (!LATE_B) { if (!(IMF_W_ABL & 16) || t == t0) IMF_W_LD_WHALF(bB, e_cur, 1) }   // half B of t: lands under the first 48 MFMAs
      if (HEAD_ROWS) IMF_W_ROWS(rows_nxt, e_nxt)
      if (LATE_B) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // rows of t have landed (the two weight halves may be in flight)
      float4 a0[RB], a1[RB];
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        if (IMF_W_ABL & 256) { a0[b] = make_float4((float)t, 1.f, 2.f, (float)lane); a1[b] = a0[b]; continue; }
        a0[b] = w_lds16(&areg[128 * b + rd_slot]);
        a1[b] = w_lds16(&areg[128 * b + 64 + rd_slot]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the row region is free
      if (!more) {
#pragma unroll
        for (int b = 0; b < RB; ++b) rows_nxt.r[b] = kNoRowW;
      }
      if (!(IMF_W_ABL & 32)) IMF_W_DMA_ROWS(e_nxt, rows_nxt)      // rows of t + 1: a whole sub-stage to land
      __builtin_amdgcn_sched_barrier(0);                          // (the scheduler otherwise sinks the requests below ~40 MFMAs)
      bf16x8 ap[RB][3];
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        if (IMF_W_ABL & 128) {
          ap[b][0] = __builtin_bit_cast(bf16x8, a0[b]); ap[b][1] = __builtin_bit_cast(bf16x8, a1[b]); ap[b][2] = ap[b][0];
        } else {
          split_b3(a0[b], a1[b], ap[b][0], ap[b][1], ap[b][2]);
        }
      }
      if (LATE_B) { __builtin_amdgcn_sched_barrier(0); IMF_W_LD_WHALF(bB, e_cur, 1) __builtin_amdgcn_sched_barrier(0); }
#define IMF_W_TERM(I, J)                                                                                 \
  _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                          \
      _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) {                                                 \
        if (IMF_W_ABL & 64) asm volatile("" : "+v"(acc[b][CB0 + cb]) : "v"(ap[b][I]), "v"(BP[cb][J]));   /* no instruction: operands stay alive */ \
        else acc[b][CB0 + cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[b][I], BP[cb][J], acc[b][CB0 + cb], 0, 0, 0); \
      }
      {
        constexpr int CB0 = 0;
#define BP bA
        IMF_B3_TERMS(IMF_W_TERM)
#undef BP
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(IMF_W_ABL & 16)) IMF_W_LD_WHALF(bA, e_nxt, 0)         // half A of t + 1: lands under the second 48 MFMAs
      e_cur = e_nxt;
      e_nxt = (unsigned)__builtin_amdgcn_readfirstlane((int)w_lds_u32(&stab[t + 2 < kSubTab ? t + 2 : kSubTab - 1]));
      if (!HEAD_ROWS) IMF_W_ROWS(rows_nxt, e_nxt)
      __builtin_amdgcn_sched_barrier(0);
      {
        constexpr int CB0 = 2;
#define BP bB
        IMF_B3_TERMS(IMF_W_TERM)
#undef BP
      }
#undef IMF_W_TERM
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the trailing requests (see above)
  } else {
  // (fp32 / split-f16 keep BOTH operands as LDS-DMA images.  Their 8 KiB weight block as eight register loads a sub-stage
  // ahead -- the bf16x3 scheme above, 179-192 VGPRs -- was built and measured: fp32 MFMA pair step 1.917 -> 1.98 ms, the
  // wave-split shapes in isolation +3 % (fp32) / +3-5 % (split-f16), split-f16 step +-0.  With 8 instead of 12 pieces per
  // sub-stage and one register set more to copy there is nothing to win.)
#pragma unroll 1
  for (int t = t0; t < t1; ++t) {
    // sub-stage t has landed: the region is private, the wavefront's own counter is the only wait
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float4 a0[RB], a1[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      a0[b] = w_lds16(&areg[128 * b + rd_slot]);
      a1[b] = w_lds16(&areg[128 * b + 64 + rd_slot]);
    }
    if constexpr (AR == kArF32) {
      // B operands: quad (j, cb) of the fp32 image = W[16 j + 4 q4 + t][16 cb + r16], t = 0 .. 3
      float4 b0[4], b1[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        b0[cb] = w_lds16(&wreg[cb * 64 + lane]);
        b1[cb] = w_lds16(&wreg[(4 + cb) * 64 + lane]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (t + 1 < t1) {
        IMF_W_DMA(e_nxt, rows_nxt)                      // lands under the 128 MFMAs below
        e_nxt = (unsigned)__builtin_amdgcn_readfirstlane((int)w_lds_u32(&stab[t + 2 < kSubTab ? t + 2 : kSubTab - 1]));
        IMF_W_ROWS(rows_nxt, e_nxt)
      }
      // k-step (j, t): channel 16 j + 4 q4 + t; sixteen independent accumulators between two MFMAs of one accumulator
#define IMF_W_STEP(AV, BV, C)                                                                            \
  _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                          \
      _Pragma("unroll") for (int cb = 0; cb < 4; ++cb)                                                   \
          acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[b].C, BV[cb].C, acc[b][cb], 0, 0, 0);
      IMF_W_STEP(a0, b0, x) IMF_W_STEP(a0, b0, y) IMF_W_STEP(a0, b0, z) IMF_W_STEP(a0, b0, w)
      IMF_W_STEP(a1, b1, x) IMF_W_STEP(a1, b1, y) IMF_W_STEP(a1, b1, z) IMF_W_STEP(a1, b1, w)
#undef IMF_W_STEP
    } else {
    f16x8 bh[4], bl[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      bh[cb] = w_lds_f16x8(&wreg[(2 * cb) * 64 + lane]);
      bl[cb] = w_lds_f16x8(&wreg[(2 * cb + 1) * 64 + lane]);
    }
    // every fragment is in registers before the region is refilled
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t + 1 < t1) {
      IMF_W_DMA(e_nxt, rows_nxt)                      // lands under the 48 MFMAs below
      e_nxt = (unsigned)__builtin_amdgcn_readfirstlane((int)w_lds_u32(&stab[t + 2 < kSubTab ? t + 2 : kSubTab - 1]));
      IMF_W_ROWS(rows_nxt, e_nxt)
    }
    f16x8 ah[RB], al[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      if (PRE) { ah[b] = __builtin_bit_cast(f16x8, a0[b]); al[b] = __builtin_bit_cast(f16x8, a1[b]); }
      else w_split8(a0[b], a1[b], ah[b], al[b]);
    }
    // per accumulator: lo*hi, hi*lo, hi*hi (k_spconv_g's order); consecutive MFMAs on different accumulators
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
        acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[b], bh[cb], acc[b][cb], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
        acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b], bl[cb], acc[b][cb], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
        acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b], bh[cb], acc[b][cb], 0, 0, 0);
    }
  }
  }
#undef IMF_W_DMA_ROWS
#undef IMF_W_LD_WHALF
#undef IMF_W_DMA
#undef IMF_W_ROWS

  if ((IMF_W_ABL & 4) && acc[0][0][0] != 12345.f) return;
  // ---- the W partial tiles meet in LDS (each wavefront's own region: its DMAs have all landed and been read), PB row blocks per
  // pass ---- element (row, col) of wavefront w and the pass at float index  w * 4 REG_F4 + row * 64 + (((col >> 2) ^ f(row)) << 2) + (col & 3),
  // f(row) = 4 * ((row >> 2) & 1): conflict-free for the ds_write_b32 of the accumulator layout and the ds_read_b128 below
  constexpr int PT = (256 * PB + NT - 1) / NT;       // float4 per thread and pass
  const float un = p.w_unscale ? *p.w_unscale : 1.f;
#pragma unroll
  for (int b0 = 0; b0 < RB; b0 += PB) {              // PB row blocks per pass (bf16x3 units of 3 / 4 blocks: two passes)
    const int nb = RB - b0 < PB ? RB - b0 : PB;
    if (b0) __syncthreads();                         // the previous pass has been read
    {
      float *const mine = reinterpret_cast<float *>(areg);
#pragma unroll
      for (int bb = 0; bb < PB; ++bb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (bb >= nb) continue;
            const int row = 16 * bb + 4 * q4 + r, col = 16 * cb + r16;
            mine[row * 64 + ((((col >> 2) ^ (((row >> 2) & 1) << 2))) << 2) + (col & 3)] = acc[b0 + bb][cb][r];
          }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PT; ++i) {
      const int idx = i * NT + tid, lrow = idx >> 4, c4 = idx & 15;
      if (((256 * PB) % NT != 0 || nb != PB) && idx >= 256 * nb) break;
      const int row = 16 * b0 + lrow;
      const int col = y * 64 + 4 * c4;
      const int phys = lrow * 16 + (c4 ^ (((lrow >> 2) & 1) << 2));
      float4 s = w_lds16(&smem[phys]);
#pragma unroll
      for (int w = 1; w < W; ++w) {                  // wavefront order: fixed, deterministic
        const float4 v = w_lds16(&smem[w * REG_F4 + phys]);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      const int orow = row0 + row < slots_act ? row_of_slot(p, row0 + row) : -1;
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.scale) sc = *reinterpret_cast<const float4 *>(p.scale + col);
      if (p.shift) sh = *reinterpret_cast<const float4 *>(p.shift + col);
      s.x = (s.x * un) * sc.x + sh.x; s.y = (s.y * un) * sc.y + sh.y;
      s.z = (s.z * un) * sc.z + sh.z; s.w = (s.w * un) * sc.w + sh.w;
      if (p.residual && orow >= 0) {
        const float4 rr = p.res_split ? load_split4(p.residual, orow, p.cout, col)
                                      : *reinterpret_cast<const float4 *>(p.residual + (long long)orow * p.cout + col);
        s.x += rr.x; s.y += rr.y; s.z += rr.z; s.w += rr.w;
      }
      if (p.relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
      if (range_guard(p)) {                          // range guard for the consumer's f16 operands
        const bool bad = orow >= 0 && (out_of_f16_range(s.x) || out_of_f16_range(s.y) || out_of_f16_range(s.z) ||
                                       out_of_f16_range(s.w));
        if (__ballot(bad) != 0ull && lane == 0) atomicOr(p.err, 32);
      }
      if (p.l2norm) {                                // cout == 64: the row is these 16 consecutive lanes
        float ss = s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w;
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        ss += __shfl_xor(ss, 4, 64);
        ss += __shfl_xor(ss, 8, 64);
        const float nrm = sqrtf(ss);
        s.x /= nrm; s.y /= nrm; s.z /= nrm; s.w /= nrm;   // no eps: resunet.py:230
      }
      if (orow >= 0) {
        if (p.out_split) store_split4(p.out, orow, p.cout, col, make_float4(s.x, s.y, s.z, s.w));
        else *reinterpret_cast<float4 *>(p.out + (long long)orow * p.cout + col) = s;
      }
    }
  }
}

// grid = (tiles, cout / 64); `waves` = 8 (512 threads, one workgroup per CU) or 4 (256 threads, two per CU)
void launch_spconv_w(const ConvParams &p_in, unsigned tiles, int waves, hipStream_t st, int use) {
  // use bit 1: half-tile workgroups (RB 2; bf16x3: 4 wavefronts, with bit 3 also 8); bit 2: 48-row units (RB 3; 8 wavefronts, bf16x3
  // also 4); bit 3 (bf16x3): the build for one more wavefront per SIMD -- whole tiles of 4 wavefronts for three, half tiles for four
  const bool half8 = (use & 2) != 0 && (use & 8) != 0 && waves == 8 && p_in.arith == kArBf16x3;   // half tiles of 8 wavefronts, two workgroups per CU
  const bool half = ((use & 2) != 0 && waves == 4 && p_in.arith == kArBf16x3) || half8;
  const bool u48 = (use & 4) != 0 && (waves == 8 || (waves == 4 && p_in.arith == kArBf16x3)) && !half;
  const bool occ_bit = (use & 8) != 0;
  const bool occ3 = occ_bit && waves == 4 && p_in.arith == kArBf16x3 && !half && !u48;   // whole tiles, three wavefronts per SIMD
  use &= 1;
  if (half) tiles *= 2;
  if (u48) tiles = (tiles * 4u + 2u) / 3u;
  // w_xcd 1 = slab by XCD, tiles interleaved (pair step 1.092 -> 1.074 ms, round 3); 2 = slab by XCD AND one range of
  // consecutive tiles per XCD (0.953 -> 0.940 ms on top: the XCD's L2 serves a fraction of the input rows)
  const int xcd_env = 2;
  ConvParams p = p_in;
  // (a transposed map's tiles are grouped by parity class: consecutive tiles there are not neighbours in space -- mode 1)
  const bool transposed = p.n_slots != (p.n_out + IMF_TILE_ROWS - 1) / IMF_TILE_ROWS * IMF_TILE_ROWS;
  // (round 6, in the step with conv2_tr on this kernel: mode 1 for transposed maps 1.1277 / 1.1254 ms, mode 2: 1.1706 / 1.1712, plain order 1.1339 / 1.1331)
  p.w_xcd = xcd_env == 2 && transposed ? 1 : xcd_env;
  const unsigned slabs = (unsigned)(p.cout / 64);
  const dim3 grid(p.w_xcd == 2 && slabs <= 8 && (slabs & (slabs - 1)) == 0 ? (tiles + 7u) / 8u * 8u : tiles, slabs, 1);
  const bool cat = p.c_b > 0;
  const int ar = (p.arith == kArF32 || p.arith == kArBf16x3) ? p.arith : (p.a_split ? kArF16x2Pre : kArF16x2);
#define IMF_W_LAUNCH(AR)                                                    \
  do {                                                                      \
    if (use == 1 && waves == 8 && !cat) {                                   \
      k_spconv_w<false, 8, AR, 1><<<grid, 512, 0, st>>>(p);                 \
    } else if (waves == 8) {                                                \
      if (cat) k_spconv_w<true, 8, AR><<<grid, 512, 0, st>>>(p);            \
      else     k_spconv_w<false, 8, AR><<<grid, 512, 0, st>>>(p);           \
    } else {                                                                \
      if (cat) k_spconv_w<true, 4, AR><<<grid, 256, 0, st>>>(p);            \
      else     k_spconv_w<false, 4, AR><<<grid, 256, 0, st>>>(p);           \
    }                                                                       \
  } while (0)
  if (ar == kArF32 && !u48) IMF_W_LAUNCH(kArF32);
  else if (half8) {
    if (cat) k_spconv_w<true, 8, kArBf16x3, 0, 2, 4><<<grid, 512, 0, st>>>(p);
    else     k_spconv_w<false, 8, kArBf16x3, 0, 2, 4><<<grid, 512, 0, st>>>(p);
  }
  else if (half && occ_bit) {                        // half tiles under the four-wavefronts-per-SIMD budget (127 VGPRs, 25 KiB)
    if (cat) k_spconv_w<true, 4, kArBf16x3, 0, 2, 4><<<grid, 256, 0, st>>>(p);
    else     k_spconv_w<false, 4, kArBf16x3, 0, 2, 4><<<grid, 256, 0, st>>>(p);
  }
  else if (half) {
    if (cat) k_spconv_w<true, 4, kArBf16x3, 0, 2><<<grid, 256, 0, st>>>(p);
    else     k_spconv_w<false, 4, kArBf16x3, 0, 2><<<grid, 256, 0, st>>>(p);
  }
  else if (u48) {
#define IMF_W_LAUNCH_U(AR)                                                  \
  do {                                                                      \
    if (cat) k_spconv_w<true, 8, AR, 0, 3><<<grid, 512, 0, st>>>(p);        \
    else     k_spconv_w<false, 8, AR, 0, 3><<<grid, 512, 0, st>>>(p);       \
  } while (0)
    if (ar == kArF32) IMF_W_LAUNCH_U(kArF32);
    else if (ar == kArBf16x3 && waves == 4) {
      if (cat) k_spconv_w<true, 4, kArBf16x3, 0, 3><<<grid, 256, 0, st>>>(p);
      else     k_spconv_w<false, 4, kArBf16x3, 0, 3><<<grid, 256, 0, st>>>(p);
    }
    else if (ar == kArBf16x3) IMF_W_LAUNCH_U(kArBf16x3);
    else if (ar == kArF16x2Pre) IMF_W_LAUNCH_U(kArF16x2Pre);
    else IMF_W_LAUNCH_U(kArF16x2);
#undef IMF_W_LAUNCH_U
  }
  else if (occ3) {
    if (cat) k_spconv_w<true, 4, kArBf16x3, 0, 4, 3><<<grid, 256, 0, st>>>(p);
    else     k_spconv_w<false, 4, kArBf16x3, 0, 4, 3><<<grid, 256, 0, st>>>(p);
  }
  else if (ar == kArBf16x3) IMF_W_LAUNCH(kArBf16x3);
  else if (ar == kArF16x2Pre) IMF_W_LAUNCH(kArF16x2Pre);
  else IMF_W_LAUNCH(kArF16x2);
#undef IMF_W_LAUNCH
}

}  // namespace imf
