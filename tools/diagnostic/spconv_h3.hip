// Sparse convolution, variant 6: fp32 arithmetic emulated on the f16 matrix pipe ("split-f16").
//
// The fp32 MFMA (v_mfma_f32_16x16x4_f32) issues once per 32 cycles for 2 048 FLOP; the f16 MFMA
// (v_mfma_f32_16x16x32_f16) once per ~17 cycles for 16 384 FLOP.  Every fp32 operand is written as
// x = hi + lo with hi = f16(x), lo = f16(x - hi) (22-23 significant bits; f16 subnormals are kept by
// the converts and by the MFMA on gfx950, tools/ubench/mfma_f16_denorm.hip), and the product is
// formed as lo_a*hi_w + hi_a*lo_w + hi_a*hi_w with fp32 accumulation inside the MFMA: three f16
// MFMAs per 32 input channels instead of eight fp32 ones (~5x fewer matrix-pipe cycles) at fp32-class
// accuracy (the dropped lo*lo term is 2^-22 relative; measured end to end: max |dF| 3e-7 against an
// fp64-accumulated network, plain fp32 is 2e-7).  Inputs must stay below the f16 range (65 504) --
// true for BatchNorm'ed activations; the weights are split once at pack time.
//
// Work decomposition is the one of variant 0 (spconv.hip): 64-row rulebook tile x 32/64-wide output
// slab x a partition of the tile's active offsets; 16 KiB weight stages double-buffered in LDS with
// register prefetch; A gathered straight from the input rows (lane l: 8 consecutive channels
// 32cc + 4(l>>4).. and 32cc + 16 + 4(l>>4).. of row nbr[k][l&15], split to hi/lo in registers).
//
// DIAGNOSTIC BUILDS ONLY since round 3 (`make h3`, `make stamps`: -DIMF_WITH_H3): the product library runs variant 6 on
// k_spconv_g (LDS-DMA staging, spconv_g.hip) and k_spconv_w (spconv_w.hip).  This register-staged kernel -- round 1's
// default, bit-identical to k_spconv_g -- carries the in-kernel s_memtime stamps, the in-launch split-K combine
// (tickets) and the balanced tail, all measured and recorded in DESIGN.md 4 / 4c.
#ifdef IMF_WITH_H3
#include "spconv_shared.h"

#ifndef IMF_H3_ABL
#define IMF_H3_ABL 0   // timing experiments only (tools/h3_ablations.sh): 1 no MFMA, 2 no split, 4 no A loads, 8 no W loads, 16 no LDS write, 32 no LDS read, 64 no barrier
#endif

namespace imf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const float4 &x0, const float4 &x1, f16x8 &hi, f16x8 &lo) {
  const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const _Float16 h = (_Float16)v[t];
    hi[t] = h;
    lo[t] = (_Float16)(v[t] - (float)h);
  }
}

#ifdef IMF_H3_STAMPS   // diagnostic build only (make stamps; tools/conv_stamps.py): s_memtime stamps -> p.partial
#define IMF_STAMP(i)                                                                             \
  do {                                                                                           \
    if (lane == 0 && (i) < 128)                                                                  \
      reinterpret_cast<long long *>(p.partial)[((long long)blockIdx.x * 4 + wave) * 128 + (i)] = \
          (long long)__builtin_readcyclecounter();                                               \
  } while (0)
#else
#define IMF_STAMP(i) do { } while (0)
#endif

// Sub-stage table entry (one per (offset ordinal jk, input-channel chunk cc) of the partition, built once per
// workgroup): weight sub-stage index k * ncc + cc | jk << 9 | source B? << 14 | chunk index inside its source << 15.
// The main loop reads ONE word per sub-stage and derives every address from it with a handful of scalar ops -- the
// previous formulation advanced (jk, cc) with compare/select chains and re-derived the source, the stride and the
// row offset per sub-stage (~100 bookkeeping instructions per 24 MFMAs; counters and ablations in
// profiles/r02_pmc_conv_counters.txt showed the kernel issue-bound on exactly those).
constexpr int kDummyJk = kKCache - 1;        // neighbour-table row that is always "no input" (kvol < kKCache)
constexpr unsigned kNoRow = 0x00FFFFFFu;     // 24-bit row index of a missing input: kNoRow * stride lands beyond the
                                             // buffer window for every stride that is a multiple of 128 bytes

// USE tags the launch for profilers only (same code): 0 = the 23 sparse convolutions of the ResUNet, 1 = the dense
// image-branch convolutions that run on this kernel over static pixel tables (csrc/image.hip) -- so that a kernel
// trace's per-symbol averages can be compared with bench.py's roofline block, which counts the sparse launches.
// CAT: the input is the channel concatenation of two matrices (in_a | in_b), else in_a alone.
template <int CO_BLK, int USE, bool CAT>
__global__ void __launch_bounds__(256, 4)   // 4 workgroups per CU: <= 128 registers per lane
k_spconv_h3(const ConvParams p) {
  constexpr int SUB_F4 = 2 * CO_BLK * 64;            // float4 per (k, cc) sub-stage: 512 or 256
  constexpr int SUB_SHIFT = CO_BLK == 4 ? 13 : 12;   // log2(bytes per sub-stage)
  constexpr int KG = 1024 / SUB_F4;                  // sub-stages per 16 KiB macro stage: 2 or 4
  constexpr int QPS = SUB_F4 / 256;                  // float4 per thread per sub-stage: 2 or 1
  __shared__ float4 wlds[2][1024];
  __shared__ unsigned nbr_lds[kKCache][IMF_TILE_ROWS];
  __shared__ unsigned stab[kSubTab];
  __shared__ int klist[kKCache];

  // XCD-contiguous tile order (opt-in, IMF_H3_XCD=1): workgroup b runs on XCD b % 8 and every XCD has its own 4 MiB L2,
  // so handing XCD x the x-th contiguous eighth of the tiles should let its L2 hold just an eighth of the feature
  // matrix.  Measured on the pair: 1.53 vs 1.38 ms per step -- SLOWER (neighbouring tiles then run at the same time
  // on one XCD and collide on its L2 channels; the round-robin order spreads them), hence off by default.
  int tile = blockIdx.x, z = blockIdx.z, S = gridDim.z;
  if (!p.tail_split && !p.no_xcd_swizzle) {
    // the XCD follows the LINEAR workgroup id; inside one (y, z) plane block b sits on XCD (b + off) % 8.  XCD x owns
    // the contiguous tiles [start(x), start(x) + count(x)), count(x) = blocks of the plane that land on it.
    const int nx = gridDim.x;
    const int off = (int)(((long long)nx * (blockIdx.y + (long long)gridDim.y * blockIdx.z)) & 7);
    const int x = (tile + off) & 7;
    int start = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int first = (q - off) & 7;                       // first block of the plane on XCD q
      const int cnt = first < nx ? (nx - first + 7) >> 3 : 0;
      if (q < x) start += cnt;
    }
    tile = start + ((tile - ((x - off) & 7)) >> 3);
  }
  long long part_slot0 = 0, part_slots = p.n_slots;
  if (p.tail_split > 1 && tile >= p.tail_begin) {   // balanced tail: see ConvParams
    const int r = tile - p.tail_begin;
    tile = p.tail_begin + r / p.tail_split;
    z = r % p.tail_split;
    S = p.tail_split;
    part_slot0 = (long long)p.tail_begin * IMF_TILE_ROWS;
    part_slots = p.n_slots - part_slot0;
  }
  if (p.n_out_dev) {   // capacity mode: padding tiles leave; the split is the rule applied to the actual rows
    const long long slots_act = conv_slots(p, conv_rows(p));
    if ((long long)tile * IMF_TILE_ROWS >= slots_act) return;
    if (p.dyn_split_kvol) {
      S = auto_split_rule(slots_act, p.cout, p.dyn_split_kvol, p.split_min_blocks, p.split_target);
      if (S > (int)gridDim.z) {   // far fewer rows than the capacity was chosen for: flagged, result to be discarded
        if (p.err && blockIdx.x == 0 && blockIdx.y == 0 && z == 0 && threadIdx.x == 0) atomicOr(p.err, 16);
        S = gridDim.z;
      }
      if (z >= S) return;
    }
  }
  const int y = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, q4 = lane >> 4;
  const int cin = p.c_a + (CAT ? p.c_b : 0);
  const int ncc = cin / 32;

  IMF_STAMP(0);
  uint32_t mask[IMF_MASK_WORDS] = {1u, 0u, 0u, 0u};
  if (p.tile_mask) {
#pragma unroll
    for (int w = 0; w < IMF_MASK_WORDS; ++w) mask[w] = p.tile_mask[tile * IMF_MASK_WORDS + w];
  }
  const int total = __builtin_popcount(mask[0]) + __builtin_popcount(mask[1]) +
                    __builtin_popcount(mask[2]) + __builtin_popcount(mask[3]);
  if (total == 0 && S == 1) return;                  // padding tile
  const int lo = (int)((long long)z * total / S), hi = (int)((long long)(z + 1) * total / S);
  const int nk = hi - lo;

  // offset list of this partition, built by 128 lanes in parallel: lane k owns offset k, its rank
  // among the active offsets is a popcount of the mask bits below it
  if (tid < 32 * IMF_MASK_WORDS) {
    const int w = tid >> 5, b = tid & 31;
    const uint32_t mw = w == 0 ? mask[0] : (w == 1 ? mask[1] : (w == 2 ? mask[2] : mask[3]));
    if ((mw >> b) & 1u) {
      int ord = __builtin_popcount(mw & ((1u << b) - 1u));
#pragma unroll
      for (int v = 0; v < IMF_MASK_WORDS; ++v)
        if (v < w) ord += __builtin_popcount(mask[v]);
      if (ord >= lo && ord < hi) klist[ord - lo] = tid;
    }
  }
  __syncthreads();
  IMF_STAMP(1);
  const int n_sub = nk * ncc;
  const int n_macro = (n_sub + KG - 1) / KG;
  // the tile's slice of the neighbour table (all loads in flight together, then the LDS stores) as 24-bit row
  // indices, and the sub-stage table; both padded so that the main loop needs no bounds checks: entries past the
  // partition point at the all-missing row kDummyJk and at weight sub-stage 0, i.e. they add exact zeros
  const long long tile_slot0 = (long long)tile * IMF_TILE_ROWS;
  {
    // (the loads are unconditional -- clamped offset index -- and outside any per-element branch: with a
    // "load or constant" select per element hipcc 7.2 branched around every load and waited for each one, seven
    // dependent memory round trips per workgroup instead of one; found in the ISA in round 2)
    constexpr int kPer = (kKCache * IMF_TILE_ROWS + 255) / 256;   // 7
    int v[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) v[i] = -1;
    if (p.nbr) {
      if (nk > 0) {
        const int32_t *const src = p.nbr + tile_slot0 + lane;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
          const int j = wave + 4 * i;
          v[i] = src[(long long)klist[j < nk ? j : 0] * p.n_slots];
        }
#pragma unroll
        for (int i = 0; i < kPer; ++i)
          if (wave + 4 * i >= nk) v[i] = -1;
      }
    } else if (tid < IMF_TILE_ROWS && nk > 0) {   // kvol == 1: the slot's own row
      v[0] = row_of_slot(p, tile_slot0 + tid);
    }
    if (tid < kSubTab) {
      unsigned e = (unsigned)kDummyJk << 9;
      if (tid < n_sub) {
        const int jk = tid / ncc, cc = tid - jk * ncc;
        const int ch0 = cc * 32;
        const bool second = CAT && ch0 >= p.c_a;       // a chunk never straddles the sources: c_a % 32 == 0 (host-checked)
        const int cch = second ? (ch0 - p.c_a) >> 5 : cc;
        e = (unsigned)(klist[jk] * ncc + cc) | ((unsigned)jk << 9) | ((second ? 1u : 0u) << 14) | ((unsigned)cch << 15);
      }
      stab[tid] = e;
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int j = wave + 4 * i;
      if (j < nk) nbr_lds[j][lane] = v[i] >= 0 ? (unsigned)v[i] : kNoRow;
      else if (j == kDummyJk) nbr_lds[j][lane] = kNoRow;
    }
  }
  __syncthreads();
  IMF_STAMP(2);

  f32x4 acc[CO_BLK];
#pragma unroll
  for (int cb = 0; cb < CO_BLK; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // prefetch registers: named scalars for the weight quads (an array would land in scratch)
  float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0, w2 = w0, w3 = w0;
  float4 a_next[KG][2] = {};

  // Raw buffer loads (SGPR base + 32-bit offsets): no 64-bit address arithmetic, and a row without an input at this
  // offset (kNoRow) gets an offset beyond the buffer window, which the hardware reads as zeros -- no branch, no zero
  // fill.  Window = 2 GiB - 4 KiB: kNoRow * stride mod 2^32 is >= 0x7FFFFD80 for every stride = 128 m, m <= 8.
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.w_packed), (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.in_a), (short)0, 0x7FFFF000, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(CAT ? p.in_b : p.in_a), (short)0, 0x7FFFF000, 0x00020000);
  const unsigned stride_a = (unsigned)p.c_a * 4u, stride_b = (unsigned)(CAT ? p.c_b : p.c_a) * 4u;
  const unsigned lane_base = 16u * q4;
  const unsigned row_byte = (unsigned)(wave * 16 + r16) * 4u;
  const unsigned woff0 = (unsigned)tid * 16u;
  const unsigned wslab = (unsigned)((long long)y * p.kvol * ncc * SUB_F4 * 16);     // bytes (image < 2 GiB)

  // Software pipeline of the bookkeeping: the table word of a sub-stage is read three macro stages ahead, its input
  // row two ahead, the global loads are issued one ahead -- nothing in the loop waits on an LDS round trip.
  unsigned e_b[KG], e_c[KG], e_d[KG], irow_b[KG], irow_c[KG];
#define IMF_READ_E(dst, stage)                                                                     \
  {                                                                                                \
    _Pragma("unroll") for (int g = 0; g < KG; ++g) {                                              \
      const int t = (stage) * KG + g;                                                              \
      dst[g] = stab[t < kSubTab - 1 ? t : kSubTab - 1];                                            \
    }                                                                                              \
  }
#define IMF_READ_ROW(dst, e)                                                                       \
  {                                                                                                \
    _Pragma("unroll") for (int g = 0; g < KG; ++g) {                                              \
      const unsigned jk256 = ((unsigned)__builtin_amdgcn_readfirstlane((int)e[g]) >> 1) & (31u << 8); \
      dst[g] = *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(&nbr_lds[0][0]) + jk256 + row_byte); \
    }                                                                                              \
  }
  // global loads of one macro stage: weights -> w0..w3, A fragments -> a_next
#define IMF_PREFETCH(e, irow)                                                                      \
  {                                                                                                \
    unsigned wso[KG];                                                                              \
    _Pragma("unroll") for (int g = 0; g < KG; ++g) {                                              \
      const unsigned ee = (unsigned)__builtin_amdgcn_readfirstlane((int)e[g]);                     \
      wso[g] = wslab + ((ee & 511u) << SUB_SHIFT);                                                 \
      const bool second = CAT && ((ee >> 14) & 1u);                                                \
      const unsigned soff = (ee >> 15) << 7;                                                       \
      const __amdgpu_buffer_rsrc_t rs = second ? rs_b : rs_a;                                      \
      const unsigned voff = __umul24(irow[g], second ? stride_b : stride_a) + lane_base;           \
      if (!(IMF_H3_ABL & 4)) {                                                                     \
      a_next[g][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));       \
      a_next[g][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 64u, soff, 0)); \
      } else { a_next[g][0].x += (float)voff + (float)soff; }                                      \
    }                                                                                              \
    if (IMF_H3_ABL & 8) { w0.x += (float)wso[0]; w1.x += (float)wso[KG - 1]; } else {             \
    w0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff0 + (0 % QPS) * 4096u, wso[0 / QPS], 0)); \
    w1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff0 + (1 % QPS) * 4096u, wso[1 / QPS], 0)); \
    w2 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff0 + (2 % QPS) * 4096u, wso[2 / QPS], 0)); \
    w3 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff0 + (3 % QPS) * 4096u, wso[3 / QPS], 0)); \
    }                                                                                              \
  }

  IMF_READ_E(e_b, 0)
  IMF_READ_E(e_c, 1)
  IMF_READ_E(e_d, 2)
  IMF_READ_ROW(irow_b, e_b)
  IMF_READ_ROW(irow_c, e_c)
  if (n_macro > 0) IMF_PREFETCH(e_b, irow_b)
  IMF_STAMP(3);
#pragma unroll 1
  for (int n = 0; n < n_macro; ++n) {
    IMF_STAMP(8 + 4 * n);
    float4 *wbuf = wlds[n & 1];
    if (!(IMF_H3_ABL & 16)) {
      wbuf[0 * 256 + tid] = w0;
      wbuf[1 * 256 + tid] = w1;
      wbuf[2 * 256 + tid] = w2;
      wbuf[3 * 256 + tid] = w3;
    }
    f16x8 ah[KG], al[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      if (IMF_H3_ABL & 2) {
        ah[g] = __builtin_bit_cast(f16x8, a_next[g][0]);
        al[g] = __builtin_bit_cast(f16x8, a_next[g][1]);
      } else {
        split8(a_next[g][0], a_next[g][1], ah[g], al[g]);
      }
    }
    IMF_STAMP(9 + 4 * n);
    if (!(IMF_H3_ABL & 64)) __syncthreads();   // stage n visible; every wave is past its reads of this buffer (stage n-2)
    IMF_STAMP(10 + 4 * n);
    // stage n+1: loads (rows looked up last iteration); stage n+2: rows; stage n+3: table words
#pragma unroll
    for (int g = 0; g < KG; ++g) { e_b[g] = e_c[g]; irow_b[g] = irow_c[g]; e_c[g] = e_d[g]; }
    if (n + 1 < n_macro) IMF_PREFETCH(e_b, irow_b)
    IMF_READ_ROW(irow_c, e_c)
    IMF_READ_E(e_d, n + 3)
    IMF_STAMP(11 + 4 * n);
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      f16x8 bh[CO_BLK], bl[CO_BLK];
#pragma unroll
      for (int cb = 0; cb < CO_BLK; ++cb) {
        if (IMF_H3_ABL & 32) {
          bh[cb] = __builtin_bit_cast(f16x8, w0);
          bl[cb] = __builtin_bit_cast(f16x8, w1);
        } else {
          bh[cb] = *reinterpret_cast<const f16x8 *>(&wbuf[g * SUB_F4 + (2 * cb) * 64 + lane]);
          bl[cb] = *reinterpret_cast<const f16x8 *>(&wbuf[g * SUB_F4 + (2 * cb + 1) * 64 + lane]);
        }
      }
      if (IMF_H3_ABL & 1) {   // keep the operands alive
#pragma unroll
        for (int cb = 0; cb < CO_BLK; ++cb)
          acc[cb][0] += (float)bh[cb][0] + (float)bl[cb][1] + (float)ah[g][0] + (float)al[g][1];
        continue;
      }
      // small terms first; consecutive MFMAs use different accumulators
#pragma unroll
      for (int cb = 0; cb < CO_BLK; ++cb)
        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[g], bh[cb], acc[cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < CO_BLK; ++cb)
        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[g], bl[cb], acc[cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < CO_BLK; ++cb)
        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[g], bh[cb], acc[cb], 0, 0, 0);
    }
  }
#undef IMF_PREFETCH
#undef IMF_READ_ROW
#undef IMF_READ_E
  IMF_STAMP(4);

  if (S == 1) {
    conv_epilogue<CO_BLK>(p, acc, tile, y, wave, r16, q4, p.w_unscale ? *p.w_unscale : 1.f);
    IMF_STAMP(5);
  } else {   // raw partial sums, slot-major (k_spconv_reduce finishes, or the last arriver below)
    const int CW = 16 * CO_BLK;
#pragma unroll
    for (int cb = 0; cb < CO_BLK; ++cb) {
      const int col = y * CW + cb * 16 + r16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long slot = tile_slot0 + wave * 16 + q4 * 4 + r;
        p.partial[((long long)z * part_slots + (slot - part_slot0)) * p.cout + col] = acc[cb][r];
      }
    }
    if (p.tickets && part_slot0 == 0) {
      // In-launch split-K combine (same protocol as variant 0): every wave drains its stores, one lane
      // releases at agent scope and takes a ticket; the last arriver acquires and reduces the tile.
      __shared__ int s_last;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int *cnt = p.tickets + (long long)tile * gridDim.y + y;
        const int t = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == S - 1);
        if (last) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        s_last = last;
      }
      __syncthreads();
      if (s_last) fused_reduce_tile<16 * CO_BLK>(p, S, tile_slot0, y, tid);
    }
  }
}

void launch_spconv_h3(const ConvParams &p, dim3 grid, int co_blk, hipStream_t st, int use) {
  // Default implementation since round 2: k_spconv_g (spconv_g.hip), both operands by LDS-DMA -- same sums bit for
  // bit.  This register-staged kernel stays for the in-launch combine (tickets), the balanced tail and the stamps
  // build, and as the A/B partner: kernel_tag bit 1 per call, IMF_H3_GLDS=0 for the whole process.
#ifndef IMF_H3_STAMPS
  static const int dma_env = getenv("IMF_H3_GLDS") ? atoi(getenv("IMF_H3_GLDS")) : 1;
  if (dma_env && !(use & 2) && !p.tickets && p.tail_split <= 1) {
    launch_spconv_g(p, grid, co_blk, st, use & 1);
    return;
  }
#endif
  use &= 1;
  if (p.c_b > 0) {        // two-source input (decoder skip connections)
    if (co_blk == 4) k_spconv_h3<4, 0, true><<<grid, 256, 0, st>>>(p);
    else             k_spconv_h3<2, 0, true><<<grid, 256, 0, st>>>(p);
  } else if (use == 1) {
    if (co_blk == 4) k_spconv_h3<4, 1, false><<<grid, 256, 0, st>>>(p);
    else             k_spconv_h3<2, 1, false><<<grid, 256, 0, st>>>(p);
  } else {
    if (co_blk == 4) k_spconv_h3<4, 0, false><<<grid, 256, 0, st>>>(p);
    else             k_spconv_h3<2, 0, false><<<grid, 256, 0, st>>>(p);
  }
}

}  // namespace imf

#endif  // IMF_WITH_H3
