// Sparse convolution, variant 6, second implementation: both operands go global -> LDS directly
// (`buffer_load_dwordx4 ... lds`, no VGPR destination); same arithmetic as k_spconv_h3, bit for bit.
//
// Why (round 2; tools/ubench/gather_shape.hip, profiles/r02_gather_shape.txt): k_spconv_h3 is bound by the
// vector-memory RETURN path of the CU, not by L2 / HBM, the matrix pipe or instruction issue.  Per wavefront and
// 16 KiB stage it issues four row gathers in MFMA-fragment shape (lane l: row l & 15, 16-byte piece l >> 4) and four
// contiguous 1 KiB weight loads, all into VGPRs.  Measured per wave-level load and CU on L2-resident data:
//   fragment-shaped gather -> VGPRs      31 cycles (52 % of the rows present)
//   contiguous 1 KiB block -> VGPRs      28 cycles
//   either of them          -> LDS       15 cycles
// i.e. 16 wavefronts x (4 x 31 + 4 x 28) = 3.8 k cycles per CU and stage round, against 1.5 k cycles of MFMA
// issue per SIMD -- the measured stage period is 3.2 k.  The same bytes cost half when their destination is LDS.
//
// So here a sub-stage (one kernel offset x 32 input channels) is staged as
//   W: the 8 / 4 KiB B-fragment block of the packed image, copied verbatim (thread t: 16 bytes at 16 t), and
//   A: per wavefront its 16 gathered rows x 128 bytes as two lane-linear 1 KiB images; four consecutive lanes
//      fetch one contiguous 64-byte run of ONE row (image i holds the 16-byte pieces 4i .. 4i+3),
// both by LDS-DMA into one of two buffers, one sub-stage ahead of the MFMAs; the only wait is the
// `s_waitcnt vmcnt(0)` of the per-sub-stage barrier.  The MFMA A fragment (lane l: row l & 15, pieces l >> 4 and
// 4 + (l >> 4)) is read back with two ds_read_b128.  LDS-DMA images are lane-linear (no padding possible), so the
// writer lane of (row r, piece q) is 4 r + (q ^ f(r >> 2)), f = {0, 3, 2, 1}: conflict-free for the hardware's
// ds_read_b128 lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...; MI355X_MICROARCH.md, LDS).  A row
// without an input reads beyond the buffer window, which an LDS-destination load turns into zeros like a VGPR one
// (checked on the hardware by the micro-benchmark).  Everything lives in ONE __shared__ array: with a second
// object hipcc 7.2 drains the DMA queue before every ds_read.
//
// Not covered here (launch_spconv_h3 keeps them on k_spconv_h3): the in-launch split-K combine (tickets), the
// balanced tail, the s_memtime stamps.
#include "spconv_shared.h"

#ifndef IMF_G_ABL
#define IMF_G_ABL 0   // timing experiments only (wrong results): 4 no row gathers, 8 no weight copies, 1 no MFMAs, 2 no hi/lo split,
                      // 16 no main loop at all, 32 no LDS fragment reads, 64 no barrier, 128 no prologue table loads, 256 no output stores,
                      // 512 every gather folded onto the first 1024 rows (all L2 / L1 hits)
#endif

#ifdef IMF_G_STAMPS   // diagnostic build only (make stamps_g; tools/conv_stamps_g.py): shader-clock stamps -> p.partial
#define IMF_GSTAMP(i)                                                                             \
  do {                                                                                            \
    if (lane == 0 && (i) < 256)                                                                   \
      reinterpret_cast<long long *>(p.partial)[((long long)blockIdx.x * 4 + wave) * 256 + (i)] =  \
          (long long)__builtin_readcyclecounter();                                                \
  } while (0)
#else
#define IMF_GSTAMP(i) do { } while (0)
#endif

namespace imf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

namespace {

constexpr int kDummyJk = kKCache - 1;        // neighbour-table row that is always "no input"
constexpr unsigned kNoRow = 0x00FFFFFFu;     // see spconv_h3.hip

__device__ __forceinline__ void split8(const float4 &x0, const float4 &x1, f16x8 &hi, f16x8 &lo) {
  const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const _Float16 h = (_Float16)v[t];
    hi[t] = h;
    lo[t] = (_Float16)(v[t] - (float)h);
  }
}

// LDS fragment reads behind __restrict__ parameters: after inlining they carry alias-scope metadata, and hipcc's
// waitcnt pass then orders them only against LDS-DMA stores it can prove to alias (ours carry no scope info: none).
// Without the metadata it puts `s_waitcnt vmcnt(0)` before every ds_read that follows a DMA issue -- the ordering is
// ours to guarantee (counted waits + barriers in the loops below).
__device__ __forceinline__ float4 lds_read16(const float4 *__restrict__ src) { return *src; }
__device__ __forceinline__ f16x8 lds_read_f16x8(const float4 *__restrict__ src) {
  return *reinterpret_cast<const f16x8 *>(src);
}

}  // namespace

// NB = LDS buffers per workgroup = sub-stages in flight + 1.  NB 2 (four workgroups per CU at RB 1) for launches that
// fill the chip several times over; NB 4 (two per CU, three sub-stages in flight) for launches whose few workgroups
// cannot hide the ~2 k-cycle DMA latency behind each other (launch_spconv_g picks).
//
// RB = 16-row blocks per wavefront.  With the operands arriving by DMA the kernel is bound by LDS READ bandwidth
// (ablations in tools/g_ablations.sh: with gathers, weight copies, split and MFMAs all compiled out a 64 -> 64 layer
// at 103 k rows still takes 65 of its 91 us, and 19 us once the fragment reads go too): every wavefront reads the whole
// 8 KiB B-fragment block of a sub-stage for its 12 MFMAs.  RB 2 = a workgroup takes TWO consecutive 64-row tiles, each
// wavefront two row blocks against one read of the B fragments (12 instead of 20 KiB of LDS reads per 32 rows and
// sub-stage, one weight copy per 128 rows).  Sums stay bit-identical: each tile keeps its OWN partition of its own
// active offsets (an offset outside it reads as "no input" for that tile's rows), and every accumulator sees the same
// MFMA sequence as in k_spconv_h3.
//
// Two further schedules were built on this kernel, measured bit-identical and SLOWER, and live in
// tools/experiments/spconv_g_rb2_pipe.hip: RB 2 as a launch option (64 -> 64 at 103 k rows 93 -> 97 us; the transposed
// maps up to 1.6x slower: only two workgroups fit a CU) and register double-buffering of the LDS -> VGPR fragment reads
// (93 -> 98 us at 121 VGPRs).  A third one, eight-wavefront workgroups sharing one weight block between two tiles (DMA
// instructions per row -25 % at the same 16 wavefronts per CU; tools/experiments/spconv_g_tw2.hip), was bit-identical
// and slower too (91 -> 99 us).  RB 2 was tried once more on operand images (PRE, no conversions left in the loop; round 3):
// bit-identical, pair step 0.989 -> 1.001 ms.  Only RB 1 with four wavefronts is instantiated here.
//
// PRE (round 3): the input rows are split-f16 operand images written by their producer (ConvParams::a_split) -- the two
// 16-byte pieces a lane reads ARE its hi and lo A fragments, no conversion in the loop.  Same DMA, same sums.
//
// AR (spconv_shared.h) = the arithmetic: kArF16x2 / kArF16x2Pre as above, or kArF32 (round 5) -- the fp32 weight image
// (imf_pack_weights; its (k, 32-channel) sub-stage is the same 8 / 4 KiB block at the same address as the split-f16
// image's) and fp32 rows through the same DMA pieces and LDS images, 8 x v_mfma_f32_16x16x4_f32 per 32 channels and
// column block, no conversion: variant 0, the reference's arithmetic, on this kernel's skeleton.
//
// WS1 (round 5, bf16x3): ONE buffer for the weight block and -- at NB 1 -- one for the gathered rows: every wavefront has the
// sub-stage's A and B fragments in registers before the next sub-stage is requested (a second barrier per sub-stage, the
// registers are the second buffer, as in k_spconv_w), so the requests of t + 1 land under the MFMAs of t.  What it buys is
// residency: 28.0 / 22.0 KiB of LDS per workgroup instead of 48.8 / 36.8, i.e. FIVE (64-column slabs, 96 VGPRs) or SEVEN
// (32-column slabs, 64 VGPRs) workgroups per CU instead of three / four.  Measured on the pair (same box, tools/conv_iso.py
// and the step): ring of 2 with both operands double-buffered, 3 per CU: 64 -> 64 at 103 k rows 150 us, step 1.30 ms; weights
// single, rows double, 4 per CU: 140 us, 1.27 ms; both single, 5 / 7 per CU: 132-137 us, 32 -> 32 51 -> 49 us, step -1.3 %
// and -2.9 % on two boxes.  The other direction loses every time: rows in a ring of 3 behind the single weight buffer (3
// per CU) 149 us, two tiles per workgroup (RB 2, 2 per CU) 142 us, both 153 us; rows requested at the loop head instead of
// after the reads (4 per CU): 141 us, no change.  Same fragments, same MFMA order in all of them: bit-identical sums.
template <int CO_BLK, int USE, bool CAT, int NB, int RB, int AR = kArF16x2, bool WS1 = false>
__global__ void __launch_bounds__(256, RB == 1 ? (NB == 1 ? (CO_BLK == 2 ? 7 : AR == kArBf16x3 ? 5 : 6) : NB == 2 ? 4 : (WS1 ? 3 : 2)) : 2)
k_spconv_g(const ConvParams p) {
  static_assert(!WS1 || NB <= 3, "single weight buffer: one or two row sub-stages in flight");
  constexpr bool PRE = AR == kArF16x2Pre;
  constexpr int ROWS = IMF_TILE_ROWS * RB;           // output rows per workgroup
  constexpr int SUB_F4 = (AR == kArBf16x3 ? 3 : 2) * CO_BLK * 64;   // float4 of weights per sub-stage: 512 / 256 (bf16x3: 768 / 384)
  constexpr unsigned SUB_BYTES = SUB_F4 * 16;        // bytes per weight sub-stage
  constexpr int QPS = (SUB_F4 + 255) / 256;          // weight DMAs per thread per sub-stage: 2 or 1 (bf16x3: 3 or 2)
  // bf16x3 with 32-column slabs: 384 float4 = one and a half passes of the 256 threads; in the last pass wavefronts 2, 3
  // repeat the pieces of wavefronts 0, 1 (same bytes to the same LDS addresses) so that every wavefront issues the same
  // number of DMA instructions -- the counted vmcnt waits below rely on it
  constexpr bool W_PARTIAL = SUB_F4 % 256 != 0;
  constexpr int AW_F4 = 128 * RB;                    // gathered rows per wavefront and sub-stage: RB x 2 KiB
  constexpr int BUF_F4 = SUB_F4 + 4 * AW_F4;
  // WS1: one weight buffer, NB row buffers (see above)
  constexpr int LDS_BUFS = WS1 ? SUB_F4 + NB * 4 * AW_F4 : NB * BUF_F4;
  constexpr int NBR_F4 = kKCache * ROWS / 4;
  constexpr int TAB_F4 = (kSubTab + 3) / 4;
  constexpr int KL_F4 = (kKCache + 3) / 4;
  constexpr int ABL = IMF_G_ABL;
  constexpr int D = NB == 1 ? 1 : NB - 1;            // sub-stages in flight (NB 1, WS1 only: the registers are the second buffer)
  static_assert(NB > 1 || WS1, "a single buffer needs the mid-iteration requests");
  constexpr int PER = ((IMF_G_ABL & 8) ? 0 : QPS) + ((IMF_G_ABL & 4) ? 0 : 2 * RB);   // DMA instructions per thread and sub-stage
  __shared__ float4 smem[LDS_BUFS + NBR_F4 + TAB_F4 + KL_F4];
  unsigned *const nbr_lds = reinterpret_cast<unsigned *>(smem + LDS_BUFS);               // [kKCache][ROWS]
  unsigned *const stab = reinterpret_cast<unsigned *>(smem + LDS_BUFS + NBR_F4);          // [kSubTab]
  // [kKCache] bits 0-7: the j-th offset of the list; bits 8-11 (round 6): 16-row block i of the tile has an input at it.  (No
  // array of its own: the NB 2 kernels sit exactly at four workgroups per CU.)  Entry j is written, and read in full, by the
  // wavefront that loads offset j only; everybody else masks the low bits.
  int *const klist = reinterpret_cast<int *>(smem + LDS_BUFS + NBR_F4 + TAB_F4);
  // weight block / this wavefront's row images of ring slot b
#define IMF_WBUF(b) (WS1 ? smem : smem + (b) * BUF_F4)
#define IMF_ABUF(b) (WS1 ? smem + SUB_F4 + (b) * (4 * AW_F4) + wave * AW_F4 : smem + (b) * BUF_F4 + SUB_F4 + wave * AW_F4)

  int super = blockIdx.x, z = blockIdx.z, S = gridDim.z;
  long long slots_act = p.n_slots;
  if (p.n_out_dev) slots_act = conv_slots(p, conv_rows(p));   // capacity mode: the actual rows
  if (!p.no_xcd_swizzle && (gridDim.x & 7u) == 0) {
    // XCD-contiguous tile order (launch_spconv_g pads gridDim.x to a multiple of 8, so the XCD of a workgroup is
    // blockIdx.x & 7 whatever y / z): XCD x walks ONE range of consecutive tiles, cut from the ACTUAL tiles -- the rows
    // neighbouring tiles gather are neighbours in space, its L2 then serves ~1/8 of the input rows instead of all of them.
    const unsigned t_act = (unsigned)((slots_act / IMF_TILE_ROWS + RB - 1) / RB);
    const unsigned chunk = (t_act + 7u) >> 3;
    const unsigned j = blockIdx.x >> 3;
    if (j >= chunk) return;
    super = (int)((blockIdx.x & 7u) * chunk + j);
    if ((unsigned)super >= t_act) return;
  }
  const int tile0 = super * RB;
  if (p.n_out_dev) {   // capacity mode: padding tiles leave; the split is the rule applied to the actual rows
    if ((long long)tile0 * IMF_TILE_ROWS >= slots_act) return;
    if (p.dyn_split_kvol) {
      S = auto_split_rule(slots_act, p.cout, p.dyn_split_kvol, p.split_min_blocks, p.split_target);
      if (S > (int)gridDim.z) {
        if (p.err && blockIdx.x == 0 && blockIdx.y == 0 && z == 0 && threadIdx.x == 0) atomicOr(p.err, 16);
        S = gridDim.z;
      }
      if (z >= S) return;
    }
  }
  const int y = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cin = p.c_a + (CAT ? p.c_b : 0);
  const int ncc = cin / 32;
  IMF_GSTAMP(0);

  // Per tile: its active offsets (kvol <= 27: one mask word) and, of those, the ones of partition z -- `sel`.
  // The workgroup walks the union of the tiles' selections in ascending order.
  bool valid[RB];
  uint32_t sel[RB];
  int total[RB];
  uint32_t uni = 0u;
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    valid[b] = (long long)(tile0 + b) * IMF_TILE_ROWS < slots_act;
    uint32_t m = 0u;
    if (valid[b]) m = p.tile_mask ? p.tile_mask[(tile0 + b) * IMF_MASK_WORDS] : 1u;
    total[b] = __builtin_popcount(m);
    const int lo = (int)((long long)z * total[b] / S), hi = (int)((long long)(z + 1) * total[b] / S);
    const int k = lane & 31;
    const int ord = __builtin_popcount(m & ((1u << k) - 1u));
    const bool in = ((m >> k) & 1u) && ord >= lo && ord < hi;
    sel[b] = (uint32_t)__ballot(in && lane < 32);
    uni |= sel[b];
  }
  bool any_out = S > 1;
#pragma unroll
  for (int b = 0; b < RB; ++b) any_out |= total[b] > 0;
  if (!any_out) return;                              // padding tiles only
  const int nk = __builtin_popcount(uni);
  if (tid < 32 && ((uni >> tid) & 1u)) klist[__builtin_popcount(uni & ((1u << tid) - 1u))] = tid;
  __syncthreads();
  const int n_sub = (IMF_G_ABL & 16) ? 0 : nk * ncc;
  {   // the tiles' slice of the neighbour table as 24-bit row indices, and the sub-stage table (spconv_h3.hip).
      // The loads are unconditional (clamped offset index, clamped tile) and sit outside any per-element branch: a
      // "load or constant" select per element makes hipcc 7.2 branch around every load and wait for each one --
      // seven dependent memory round trips per workgroup instead of one.
    constexpr int kPer = kKCache * ROWS / 256;       // 7 / 14
    constexpr int JSTEP = 256 / ROWS;                // offsets covered per pass of the 256 threads: 4 / 2
    const int srow = tid & (ROWS - 1), j0 = tid / ROWS;
    const int b_of = srow >> 6;                      // tile of this thread's row
    const bool vb = RB == 1 ? valid[0] : (b_of ? valid[RB - 1] : valid[0]);
    const uint32_t sb = RB == 1 ? sel[0] : (b_of ? sel[RB - 1] : sel[0]);
    const long long slot = (long long)(tile0 + (vb ? b_of : 0)) * IMF_TILE_ROWS + (srow & 63);
    int v[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) v[i] = -1;
    if (p.nbr) {
      if (nk > 0) {
        const int32_t *const src = p.nbr + slot;
        int kk[kPer];
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
          const int j = j0 + JSTEP * i;
          kk[i] = klist[j < nk ? j : 0] & 31;
          if (IMF_G_ABL & 128) { v[i] = lane + j; continue; }
          v[i] = src[(long long)kk[i] * p.n_slots];
        }
#pragma unroll
        for (int i = 0; i < kPer; ++i)
          if (j0 + JSTEP * i >= nk || !((sb >> kk[i]) & 1u)) v[i] = -1;
      }
    } else if (tid < ROWS && nk > 0 && vb) {         // kvol == 1: the slot's own row
      v[0] = row_of_slot(p, slot);
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int j = j0 + JSTEP * i;
      if (j < nk) nbr_lds[j * ROWS + srow] = v[i] >= 0 ? (unsigned)v[i] : kNoRow;
      else if (j == kDummyJk) nbr_lds[j * ROWS + srow] = kNoRow;
      // which 16-row blocks have an input at this offset (the 64 lanes of the wavefront = 64 consecutive rows of one offset)
      const unsigned long long bal = __ballot(v[i] >= 0);
      static_assert(RB == 1, "block bits: one wavefront = the 64 rows of one offset");
      if (lane == 0 && j < kKCache) {
        const unsigned bits = ((bal & 0xFFFFull) ? 1u : 0u) | ((bal & 0xFFFF0000ull) ? 2u : 0u) |
                              ((bal & 0xFFFF00000000ull) ? 4u : 0u) | ((bal & 0xFFFF000000000000ull) ? 8u : 0u);
        if (j < nk) klist[j] = (klist[j] & 31) | (int)(bits << 8);
      }
    }
  }
  __syncthreads();
  // Sub-stage table.  Bits 20 .. 23 of an entry: the 16-row blocks of the tile that have an input at the sub-stage's offset
  // (round 6).  A wavefront whose block has none takes part in the barriers and the weight copy of the sub-stage but leaves
  // out its fragment reads, its split and its MFMAs -- they would add exact zeros, the sums are the same bit for bit.  In
  // slot = row order 96 % of the (block, offset) pairs are active; on the occupancy-sorted maps (csrc/rulebook_sort.hip) ~70 %.
  // Every wavefront writes the WHOLE table (lane = offset of the list): all four store the same words, so nobody needs a
  // barrier before reading it.
  if (lane < 32) {
    const bool on = lane < nk && n_sub > 0;
    const int kl = klist[on ? lane : 0];
    for (int cc = 0; cc < ncc; ++cc) {
      if (on) {
        const int ch0 = cc * 32;
        const bool second = CAT && ch0 >= p.c_a;
        const int cch = second ? (ch0 - p.c_a) >> 5 : cc;
        stab[lane * ncc + cc] = (unsigned)((kl & 31) * ncc + cc) | ((unsigned)lane << 9) | ((second ? 1u : 0u) << 14) | ((unsigned)cch << 15) |
                                ((p.nbr ? (unsigned)(kl >> 8) & 15u : 15u) << 20);
      }
    }
    if (lane < 8) stab[lane < 7 && n_sub + lane < kSubTab - 1 ? n_sub + lane : kSubTab - 1] = (unsigned)kDummyJk << 9;   // look-ahead entries
  }

  f32x4 acc[RB][CO_BLK];
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int cb = 0; cb < CO_BLK; ++cb) acc[b][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.w_packed), (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.in_a), (short)0, 0x7FFFF000, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(CAT ? p.in_b : p.in_a), (short)0, 0x7FFFF000, 0x00020000);
  const unsigned stride_a = (unsigned)p.c_a * 4u, stride_b = (unsigned)(CAT ? p.c_b : p.c_a) * 4u;
  const unsigned woff0 = (unsigned)tid * 16u;
  const int wave_last = W_PARTIAL ? (wave & 1) : wave;                              // wavefront whose piece the last weight DMA copies
  const unsigned woff_last = (unsigned)(wave_last * 64 + lane) * 16u;
  const unsigned wslab = (unsigned)((long long)y * p.kvol * ncc * SUB_F4 * 16);     // bytes (image < 2 GiB)
  // writer role of the lane in the row gather: row lane >> 2 of a 16-row block, piece (lane & 3) ^ f(row >> 2)
  const int row_w = lane >> 2;
  const unsigned piece_w = (unsigned)((lane & 3) ^ ((4 - (row_w >> 2)) & 3));
  const unsigned wr_byte = 16u * piece_w;
  const unsigned row_byte = (unsigned)(wave * 16 * RB + row_w) * 4u;   // block b of the wavefront: + 64 b bytes
  // reader role: MFMA A fragment, row r16, pieces q4 and 4 + q4
  const int rd_slot = 4 * r16 + (q4 ^ ((4 - (r16 >> 2)) & 3));

  struct Rows { unsigned r[RB]; };
#define IMF_READ_E(t) stab[(t) < kSubTab - 1 ? (t) : kSubTab - 1]
#define IMF_READ_ROWS(dst, e)                                                                                    \
  {                                                                                                              \
    const char *const base_ = reinterpret_cast<const char *>(nbr_lds) +                                          \
        (((((unsigned)__builtin_amdgcn_readfirstlane((int)(e))) >> 9) & 31u) * (unsigned)(ROWS * 4)) + row_byte; \
    _Pragma("unroll") for (int b_ = 0; b_ < RB; ++b_)                                                            \
        (dst).r[b_] = *reinterpret_cast<const unsigned *>(base_ + 64 * b_);                                      \
  }
  // LDS-DMA of one sub-stage into buffer `b`: weights verbatim, the wavefront's 16 RB rows as 1 KiB images
#define IMF_DMA_W(e, b)                                                                                          \
  {                                                                                                              \
    const unsigned ee = (unsigned)__builtin_amdgcn_readfirstlane((int)(e));                                      \
    const unsigned wso = wslab + (ee & 511u) * SUB_BYTES;                                                        \
    float4 *const wb = IMF_WBUF(b);                                                                              \
    if (!(ABL & 8)) {                                                                                            \
    _Pragma("unroll") for (int j = 0; j < QPS; ++j) {                                                            \
        const bool last_ = W_PARTIAL && j == QPS - 1;                                                            \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void *)(wb + j * 256 + (last_ ? wave_last : wave) * 64), 16, \
                                                 (last_ ? woff_last : woff0) + (unsigned)j * 4096u, wso, 0, 0);  \
    }                                                                                                            \
    }                                                                                                            \
  }
#define IMF_DMA_R(e, rows, b)                                                                                    \
  {                                                                                                              \
    const unsigned ee = (unsigned)__builtin_amdgcn_readfirstlane((int)(e));                                      \
    const bool second = CAT && ((ee >> 14) & 1u);                                                                \
    const unsigned soff = ((ee >> 15) & 31u) << 7;                                                                       \
    const __amdgpu_buffer_rsrc_t rs = second ? rs_b : rs_a;                                                      \
    float4 *const ab = IMF_ABUF(b);                                                                              \
    if (!(ABL & 4)) {                                                                                            \
    _Pragma("unroll") for (int b_ = 0; b_ < RB; ++b_) {                                                          \
      const unsigned rr_ = (ABL & 512) && (rows).r[b_] != kNoRow ? ((rows).r[b_] & 1023u) : (rows).r[b_];        \
      const unsigned voff = __umul24(rr_, second ? stride_b : stride_a) + wr_byte;                               \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(ab + 128 * b_), 16, voff, soff, 0, 0);           \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(ab + 128 * b_ + 64), 16, voff + 64u, soff, 0, 0); \
    }                                                                                                            \
    }                                                                                                            \
  }
#define IMF_DMA(e, rows, b) { IMF_DMA_W(e, b) IMF_DMA_R(e, rows, b) }

  // Ring of NB buffers, D = NB - 1 sub-stages in flight.  Bookkeeping runs ahead of the DMA: the table word of
  // sub-stage t + D + 2 and the input rows of t + D + 1 are read in iteration t, the DMA of t + D is issued in it.
  // Waits are counted by hand (the compiler does not order ds_reads after LDS-DMAs, and __syncthreads() would
  // drain the whole queue): vmcnt(PER (D - 1)) leaves the D - 1 younger sub-stages in flight across the barrier.
  // actq[i]: this wavefront's block has an input in sub-stage t + i (bit 20 + wave of its table entry; scalar)
#define IMF_E_ACTIVE(e) (((((unsigned)__builtin_amdgcn_readfirstlane((int)(e))) >> (20 + wave)) & 1u) != 0u)
  unsigned e_b, e_c = IMF_READ_E(D), e_d = IMF_READ_E(D + 1);
  Rows irow_b, irow_c;
  bool actq[D + 2];
#pragma unroll
  for (int d = 0; d < D; ++d) actq[d] = IMF_E_ACTIVE(IMF_READ_E(d));
  actq[D] = IMF_E_ACTIVE(e_c);
  actq[D + 1] = IMF_E_ACTIVE(e_d);
  IMF_READ_ROWS(irow_c, e_c)
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < n_sub) {
      const unsigned e0 = IMF_READ_E(d);
      Rows irow0;
      IMF_READ_ROWS(irow0, e0)
      if (!WS1 || d == 0) IMF_DMA_W(e0, d)          // (WS1: one weight block in flight, row sub-stages 0 .. D - 1)
      IMF_DMA_R(e0, irow0, d)
    }
  }
  unsigned e_hist = IMF_READ_E(1);                  // WS1, D 2: the table word of sub-stage t + 1 (the next weight block)
  int slot_rd = 0, slot_wr = D % NB;   // t % NB and (t + D) % NB
  IMF_GSTAMP(1);
#pragma unroll 1
  for (int t = 0; t < n_sub; ++t) {
    IMF_GSTAMP(8 + 4 * t);
    // sub-stage t has landed (this thread's part, then -- barrier -- everyone's); every wavefront is past its reads
    // of buffer (t - 1) % NB, which the DMA below refills
    if (ABL & 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (WS1) {
      // issue order W(0) R(0) [R(1)] | W(1) R(D) | W(2) R(D + 1) ...: W(t) and R(t) have landed once at most the ONE row
      // sub-stage requested after W(t) is outstanding (D 2: R(t + 1), if there is one)
      if (D == 2 && t + 1 < n_sub) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * RB) : "memory");
      else                         asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    else if (n_sub - 1 - t >= D - 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PER * (D - 1)) : "memory");
    else                        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    IMF_GSTAMP(9 + 4 * t);
    e_b = e_c; irow_b = irow_c; e_c = e_d;
    if (!WS1 && t + D < n_sub) IMF_DMA(e_b, irow_b, slot_wr)   // (WS1: after the fragment reads, below)
    IMF_READ_ROWS(irow_c, e_c)
    e_d = IMF_READ_E(t + D + 2);
    // (not in the 64-column bf16x3 kernel with single buffers: at its 96 VGPRs -- five workgroups per CU -- the branches cost
    // 13 spilled registers; its launches are the transposed maps', whose tiles are grouped by parity class and have no empty
    // blocks to speak of)
    constexpr bool SKIP = ABL == 0 && !(WS1 && CO_BLK == 4 && AR == kArBf16x3);
    const bool act = SKIP ? actq[0] : true;             // this wavefront has work in sub-stage t
#pragma unroll
    for (int i = 0; i <= D; ++i) actq[i] = actq[i + 1];
    actq[D + 1] = IMF_E_ACTIVE(e_d);
    IMF_GSTAMP(10 + 4 * t);
    const float4 *const wbuf = IMF_WBUF(slot_rd);
    const float4 *const abuf = IMF_ABUF(slot_rd);
    const int slot_dma = slot_wr;
    slot_rd = slot_rd + 1 >= NB ? 0 : slot_rd + 1;
    slot_wr = slot_wr + 1 >= NB ? 0 : slot_wr + 1;
    if (ABL & 32) {
#pragma unroll
      for (int cb = 0; cb < CO_BLK; ++cb) acc[0][cb][0] += (float)t + (float)e_b;
      continue;
    }
    // WS1, once the sub-stage's fragments are requested: every wavefront holds them -> the single weight buffer and the row
    // buffer of sub-stage t + D may be refilled
#define IMF_WS_MID                                                                                     \
    if constexpr (WS1) {                                                                               \
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                  \
      const unsigned e_w = D == 1 ? e_b : e_hist;   /* table word of sub-stage t + 1 */                \
      e_hist = e_b;                                                                                    \
      if (t + 1 < n_sub) IMF_DMA_W(e_w, 0)                                                             \
      if (t + D < n_sub) IMF_DMA_R(e_b, irow_b, slot_dma)                                              \
    }
    if constexpr (AR == kArF32) {
      float4 a0[RB], a1[RB], b0[CO_BLK], b1[CO_BLK];
      if (act) {
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          a0[b] = lds_read16(&abuf[128 * b + rd_slot]);
          a1[b] = lds_read16(&abuf[128 * b + 64 + rd_slot]);
        }
#pragma unroll
        for (int cb = 0; cb < CO_BLK; ++cb) {    // quad (j, cb) of the fp32 image: W[16 j + 4 q4 + t][16 cb + r16]
          b0[cb] = lds_read16(&wbuf[cb * 64 + lane]);
          b1[cb] = lds_read16(&wbuf[(CO_BLK + cb) * 64 + lane]);
        }
      }
      IMF_WS_MID
      if (!act) continue;
#define IMF_G_STEP(AV, BV, C)                                                                          \
  _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                       \
      _Pragma("unroll") for (int cb = 0; cb < CO_BLK; ++cb)                                            \
          acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[b].C, BV[cb].C, acc[b][cb], 0, 0, 0);
      IMF_G_STEP(a0, b0, x) IMF_G_STEP(a0, b0, y) IMF_G_STEP(a0, b0, z) IMF_G_STEP(a0, b0, w)
      IMF_G_STEP(a1, b1, x) IMF_G_STEP(a1, b1, y) IMF_G_STEP(a1, b1, z) IMF_G_STEP(a1, b1, w)
#undef IMF_G_STEP
      continue;
    }
    if constexpr (AR == kArBf16x3) {
      // fp32 rows -> three bf16 parts in registers; the image's quads (3 cb + part) are the matching B parts
      // (the fragments are READ before the mid-iteration barrier and SPLIT after it: the requests of sub-stage t + 1 go out as
      // early as possible and land under the split and the MFMAs)
      float4 ar0[RB], ar1[RB], bp[CO_BLK][3];
      if (act) {
#pragma unroll
        for (int b = 0; b < RB; ++b) {
          ar0[b] = lds_read16(&abuf[128 * b + rd_slot]);
          ar1[b] = lds_read16(&abuf[128 * b + 64 + rd_slot]);
        }
#pragma unroll
        for (int cb = 0; cb < CO_BLK; ++cb)
#pragma unroll
          for (int h = 0; h < 3; ++h) bp[cb][h] = lds_read16(&wbuf[(3 * cb + h) * 64 + lane]);
      }
      IMF_WS_MID
      if (act) {
        bf16x8 ap[RB][3];
#pragma unroll
        for (int b = 0; b < RB; ++b) split_b3(ar0[b], ar1[b], ap[b][0], ap[b][1], ap[b][2]);
#define IMF_G_TERM(I, J)                                                                               \
  _Pragma("unroll") for (int b = 0; b < RB; ++b)                                                       \
      _Pragma("unroll") for (int cb = 0; cb < CO_BLK; ++cb)                                            \
          acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[b][I], __builtin_bit_cast(bf16x8, bp[cb][J]), acc[b][cb], 0, 0, 0);
        IMF_B3_TERMS(IMF_G_TERM)
#undef IMF_G_TERM
      }
      continue;
    }
    float4 ar0[RB], ar1[RB];
    f16x8 bh[CO_BLK], bl[CO_BLK];
    if (act) {
#pragma unroll
      for (int b = 0; b < RB; ++b) {
        ar0[b] = lds_read16(&abuf[128 * b + rd_slot]);
        ar1[b] = lds_read16(&abuf[128 * b + 64 + rd_slot]);
      }
#pragma unroll
      for (int cb = 0; cb < CO_BLK; ++cb) {
        bh[cb] = lds_read_f16x8(&wbuf[(2 * cb) * 64 + lane]);
        bl[cb] = lds_read_f16x8(&wbuf[(2 * cb + 1) * 64 + lane]);
      }
    }
    IMF_WS_MID
    if (!act) continue;
    f16x8 ah[RB], al[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      if ((ABL & 2) || PRE) {
        ah[b] = __builtin_bit_cast(f16x8, ar0[b]);
        al[b] = __builtin_bit_cast(f16x8, ar1[b]);
      } else {
        split8(ar0[b], ar1[b], ah[b], al[b]);
      }
    }
#ifdef IMF_G_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments in registers (perturbs the schedule a little)
    IMF_GSTAMP(11 + 4 * t);
#endif
    if (ABL & 1) {   // keep the operands alive
#pragma unroll
      for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int cb = 0; cb < CO_BLK; ++cb)
          acc[b][cb][0] += (float)bh[cb][0] + (float)bl[cb][1] + (float)ah[b][0] + (float)al[b][1];
      continue;
    }
    // per accumulator the order of k_spconv_h3: lo*hi, hi*lo, hi*hi; consecutive MFMAs on different accumulators
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int cb = 0; cb < CO_BLK; ++cb)
        acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[b], bh[cb], acc[b][cb], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int cb = 0; cb < CO_BLK; ++cb)
        acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b], bl[cb], acc[b][cb], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int cb = 0; cb < CO_BLK; ++cb)
        acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b], bh[cb], acc[b][cb], 0, 0, 0);
  }
#undef IMF_DMA
#undef IMF_WS_MID
#undef IMF_DMA_W
#undef IMF_DMA_R
#undef IMF_READ_ROWS
#undef IMF_E_ACTIVE
#undef IMF_READ_E
#undef IMF_WBUF
#undef IMF_ABUF

  IMF_GSTAMP(2);
  if ((IMF_G_ABL & 256) && acc[0][0][0] != 12345.f) return;
  const bool staged = S == 1 && !p.l2norm && !p.geglu;
  if (staged) __syncthreads();                       // every wavefront is past its last fragment reads: LDS is free
  // block b of wavefront w holds rows 16 (RB w + b) .. + 15 of the workgroup's ROWS: tile and 16-row block inside it
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    const int blk = RB * wave + b;                   // 16-row block of the workgroup
    const int tb = blk >> 2;                         // tile of the block (0 .. RB - 1)
    const int tile = tile0 + tb, wv = blk & 3;
    const bool vt = RB == 1 ? valid[0] : (tb ? valid[RB - 1] : valid[0]);
    const int tt = RB == 1 ? total[0] : (tb ? total[RB - 1] : total[0]);
    if (!vt) continue;
    if (S == 1) {
      if (staged) {   // through LDS: whole 8-channel pieces per lane (16-byte accesses)
        if (tt > 0)
          conv_epilogue_staged<CO_BLK>(p, acc[b], reinterpret_cast<float *>(smem) + blk * (16 * (16 * CO_BLK + 4)), tile, y,
                                       wv, lane, p.w_unscale ? *p.w_unscale : 1.f);
      } else if (tt > 0) {
        conv_epilogue<CO_BLK>(p, acc[b], tile, y, wv, r16, q4, p.w_unscale ? *p.w_unscale : 1.f);
      }
    } else {   // raw partial sums, slot-major (k_spconv_reduce finishes)
      const int CW = 16 * CO_BLK;
#pragma unroll
      for (int cb = 0; cb < CO_BLK; ++cb) {
        const int col = y * CW + cb * 16 + r16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long slot = (long long)tile * IMF_TILE_ROWS + wv * 16 + q4 * 4 + r;
          p.partial[((long long)z * p.n_slots + slot) * p.cout + col] = acc[b][cb][r];
        }
      }
    }
  }
  IMF_GSTAMP(3);
}

void launch_spconv_g(const ConvParams &p, dim3 grid, int co_blk, hipStream_t st, int use) {
  // use: profiling label (bit 0)
  use &= 1;
  // deep ring (NB 4, two workgroups per CU) when the whole launch is resident at once that way (<= 512 workgroups);
  // measured (tools/layer_times.py): 438 unsplit workgroups of a pair's stride-2 level 43 -> 33 us, but 544 workgroups
  // 28 -> 33 us (a second round of 32), and every launch that fills the chip is faster with four workgroups per CU
  const int nb_env = 0, nb_wgs = 512;   // ring depth: by size (the deep ring when the whole launch is resident, <= 512 workgroups)
  long long wgs = (long long)grid.x * grid.y * grid.z;
  if (p.n_out_dev) {
    // capacity mode: the grid covers a capacity and the largest split, the working workgroups are decided on the
    // device.  Estimate them the way the buckets are sized (rows ~ capacity / 1.2; model/graph.py) -- a wrong guess
    // costs time only, both ring depths form the same sums.
    const long long tiles = grid.x > 1 ? (long long)(grid.x / 1.2) : 1;
    const int s_est = p.dyn_split_kvol ? auto_split_rule(tiles * IMF_TILE_ROWS, p.cout, p.dyn_split_kvol,
                                                         p.split_min_blocks, p.split_target)
                                       : (int)grid.z;
    wgs = tiles * grid.y * (s_est < (int)grid.z ? s_est : (int)grid.z);
  }
  const bool deep = nb_env ? nb_env >= 4 : wgs <= nb_wgs;
  // launches that fill the chip.  bf16x3: single buffers (WS1: 5 / 7 workgroups per CU instead of 3 / 4).  Split-f16 and fp32
  // (8 / 4 KiB weight blocks, four per CU with the ring of two already): single buffers for the 32-column slabs only -- A/B
  // on one box, 32 -> 32 at 103 k rows 38.5 -> 36.8 us (f16x2), 67.8 -> 65.3 us (fp32; in situ 52.2 -> 48.7), but 64 -> 64
  // 91.6 -> 92.9 us and 219 -> 242 us: six workgroups of the 64-column kernel per CU lose more to the second barrier per
  // sub-stage than they gain
#define IMF_G_LAUNCH(CB, USE, CAT)                                                         \
  do {                                                                                     \
    if (p.arith == kArF32) {   /* variant 0 */                                             \
      if (deep) k_spconv_g<CB, USE, CAT, 4, 1, kArF32><<<grid, 256, 0, st>>>(p);           \
      else      k_spconv_g<CB, USE, CAT, (CB == 2 ? 1 : 2), 1, kArF32, CB == 2><<<grid, 256, 0, st>>>(p); \
    } else if (p.arith == kArBf16x3) {   /* variant 3: 12 / 6 KiB of weights per sub-stage -- ring of 3 where the f16 kernels take 4 */ \
      /* (single buffers, 5 / 7 workgroups per CU, for the launches that fill the chip; the resident ones keep the ring) */ \
      if (deep) k_spconv_g<CB, USE, CAT, 3, 1, kArBf16x3><<<grid, 256, 0, st>>>(p);        \
      else      k_spconv_g<CB, USE, CAT, 1, 1, kArBf16x3, true><<<grid, 256, 0, st>>>(p);  \
    } else if (p.a_split) {   /* operand images: the ResUNet's own layers (label 0) */     \
      if (deep) k_spconv_g<CB, 0, CAT, 4, 1, kArF16x2Pre><<<grid, 256, 0, st>>>(p);        \
      else      k_spconv_g<CB, 0, CAT, (CB == 2 ? 1 : 2), 1, kArF16x2Pre, CB == 2><<<grid, 256, 0, st>>>(p); \
    } else if (deep) k_spconv_g<CB, USE, CAT, 4, 1><<<grid, 256, 0, st>>>(p);              \
    else             k_spconv_g<CB, USE, CAT, (CB == 2 ? 1 : 2), 1, kArF16x2, CB == 2><<<grid, 256, 0, st>>>(p); \
  } while (0)
  if (p.c_b > 0) {        // two-source input (decoder skip connections)
    if (co_blk == 4) IMF_G_LAUNCH(4, 0, true); else IMF_G_LAUNCH(2, 0, true);
  } else if (use == 1) {
    if (co_blk == 4) IMF_G_LAUNCH(4, 1, false); else IMF_G_LAUNCH(2, 1, false);
  } else {
    if (co_blk == 4) IMF_G_LAUNCH(4, 0, false); else IMF_G_LAUNCH(2, 0, false);
  }
#undef IMF_G_LAUNCH
}

}  // namespace imf
