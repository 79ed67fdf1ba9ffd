// EXPERIMENT -- not part of the library build.  Conv kernel variants 2-5 of round 1 (fp32 MFMA), all of which lost to
// variant 0 (k_spconv_mfma) and, later, to variant 6 (split-f16, spconv_h3.hip):
//   2 k_spconv_wave  one wavefront per tile, row compaction per offset, LDS accumulator tile
//   3 k_spconv_c     512-thread workgroups over two tiles
//   4/5 k_spconv_reg register-blocked rows (RB = 2 / 1)
// They were removed from csrc/spconv.hip in round 2 (they cost build and test time and were never selected);
// the bodies are kept here verbatim for reference.  They compile inside namespace imf of spconv.hip (they use
// ConvParams, gather_a, conv_epilogue, kKCache from spconv_shared.h).
// ---- variant 2: wave-autonomous kernel with per-offset compaction ------------------------------
// One WAVEFRONT (one 64-thread workgroup, no barriers with other waves) owns a 64-row tile x one
// output slab.  For every active offset k it compacts the rows that actually have an input
// (ballot + popcount -> LDS list), so the MFMA blocks are full: ~2.6 16-row blocks per offset
// instead of 4 at 52 % occupancy (-35 % MFMA work).  Because compacted rows no longer line up with
// fixed accumulator registers, the per-offset products (MFMA with C = 0) are added into an
// LDS-resident [64 x CW] accumulator tile owned by the wave (distinct addresses per lane, fixed
// order k ascending => deterministic).  The weight chunk of (k, cc) is held in registers and reused
// by all row blocks of that offset; A fragments are gathered one block ahead.
template <int CO_BLK, int J>
__global__ void __launch_bounds__(64)
k_spconv_wave(const ConvParams p) {
  constexpr int CW = 16 * CO_BLK;
  constexpr int NB = J * CO_BLK;                     // float4 weight fragments per lane per (k, cc)
  constexpr int SUB_F4 = NB * 64;
  __shared__ __attribute__((aligned(16))) float acc_l[IMF_TILE_ROWS][CW];
  __shared__ int list_in[IMF_TILE_ROWS], list_row[IMF_TILE_ROWS];
  __shared__ int klist[kKCache];

  const int tile = blockIdx.x, y = blockIdx.y, z = blockIdx.z, S = gridDim.z;
  const int lane = threadIdx.x, r16 = lane & 15, q4 = lane >> 4;
  const int cin = p.c_a + p.c_b;
  const int ncc = cin / (16 * J);

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
  if (lane == 0) {
    int ord = 0, n = 0;
#pragma unroll
    for (int w = 0; w < IMF_MASK_WORDS; ++w) {
      uint32_t m = mask[w];
      while (m) {
        const int k = w * 32 + __builtin_ctz(m);
        m &= m - 1;
        if (ord >= lo && ord < hi) klist[n++] = k;
        ++ord;
      }
    }
  }
  {   // zero the accumulator tile: CW floats per lane
    float4 *a4 = reinterpret_cast<float4 *>(&acc_l[0][0]);
#pragma unroll
    for (int i = 0; i < CW / 4; ++i) a4[i * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();

  const long long slot0 = (long long)tile * IMF_TILE_ROWS;
  const float4 *wbase = reinterpret_cast<const float4 *>(p.w_packed) +
                        (long long)y * p.kvol * ncc * SUB_F4 + lane;

  int irow_next = -1;
  if (nk > 0)
    irow_next = p.nbr ? p.nbr[(long long)klist[0] * p.n_slots + slot0 + lane] : row_of_slot(p, slot0 + lane);

#pragma unroll 1
  for (int jk = 0; jk < nk; ++jk) {
    const int k = klist[jk];
    const int irow = irow_next;
    if (jk + 1 < nk)                                  // neighbour column of the next offset: in flight
      irow_next = p.nbr[(long long)klist[jk + 1] * p.n_slots + slot0 + lane];
    const bool valid = irow >= 0;
    const unsigned long long vm = __ballot(valid);
    const int cnt = __builtin_popcountll(vm);
    if (cnt == 0) continue;
    const int pos = __builtin_popcountll(vm & ((1ull << lane) - 1ull));
    __syncthreads();                                  // previous offset's list fully consumed
    if (valid) {
      list_in[pos] = irow;
      list_row[pos] = lane;
    }
    __syncthreads();
    const int ngroups = (cnt + 15) >> 4;
#pragma unroll 1
    for (int cc = 0; cc < ncc; ++cc) {
      float4 b[NB];
      const float4 *src = wbase + ((long long)k * ncc + cc) * SUB_F4;
      if (!(p.ablate & 8)) {
#pragma unroll
        for (int e = 0; e < NB; ++e) b[e] = src[e * 64];
      } else {
#pragma unroll
        for (int e = 0; e < NB; ++e) b[e] = make_float4(1.f, 2.f, 3.f, (float)e);
      }
      float4 a_next[J];
      {
        const int my_in = (r16 < cnt && !(p.ablate & 4)) ? list_in[r16] : -1;
#pragma unroll
        for (int j = 0; j < J; ++j) a_next[j] = gather_a(p, my_in, cc * 16 * J + 16 * j + 4 * q4);
      }
#pragma unroll 1
      for (int g = 0; g < ngroups; ++g) {
        float4 a[J];
#pragma unroll
        for (int j = 0; j < J; ++j) a[j] = a_next[j];
        if (g + 1 < ngroups) {
          const int s = (g + 1) * 16 + r16;
          const int my_in = (s < cnt && !(p.ablate & 4)) ? list_in[s] : -1;
#pragma unroll
          for (int j = 0; j < J; ++j) a_next[j] = gather_a(p, my_in, cc * 16 * J + 16 * j + 4 * q4);
        }
        f32x4 d[CO_BLK];
#pragma unroll
        for (int cb = 0; cb < CO_BLK; ++cb) d[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!(p.ablate & 1)) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
#pragma unroll
          for (int cb = 0; cb < CO_BLK; ++cb) {
            const float4 bb = b[j * CO_BLK + cb];
            d[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, bb.x, d[cb], 0, 0, 0);
            d[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, bb.y, d[cb], 0, 0, 0);
            d[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, bb.z, d[cb], 0, 0, 0);
            d[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, bb.w, d[cb], 0, 0, 0);
          }
        }
        } else {
#pragma unroll
          for (int cb = 0; cb < CO_BLK; ++cb) d[cb][0] = a[0].x + b[cb].x + a[J - 1].w + b[NB - 1].w;
        }
        // d[cb][r] = product for compacted row g*16 + 4*q4 + r, column 16*cb + r16
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int s2 = g * 16 + 4 * q4 + r;
          if (s2 < cnt && !(p.ablate & 2)) {
            const int rho = list_row[s2];
#pragma unroll
            for (int cb = 0; cb < CO_BLK; ++cb) atomicAdd(&acc_l[rho][cb * 16 + r16], d[cb][r]);
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- epilogue: whole rows out of LDS, float4 per lane -------------------------------------
  constexpr int LPR = CW / 4;                        // lanes per row
  constexpr int RPI = 64 / LPR;                      // rows per iteration
  const int c4 = lane % LPR, rsub = lane / LPR;
  const int col = y * CW + 4 * c4;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (S == 1) {
    if (p.scale) sc = *reinterpret_cast<const float4 *>(p.scale + col);
    if (p.shift) sh = *reinterpret_cast<const float4 *>(p.shift + col);
  }
#pragma unroll 1
  for (int it = 0; it < IMF_TILE_ROWS / RPI; ++it) {
    const int row = it * RPI + rsub;
    float4 v = *reinterpret_cast<const float4 *>(&acc_l[row][4 * c4]);
    if (S > 1) {
      *reinterpret_cast<float4 *>(p.partial + ((long long)z * p.n_slots + slot0 + row) * p.cout + col) = v;
      continue;
    }
    const int orow = row_of_slot(p, slot0 + row);
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    if (p.residual && orow >= 0) {
      const float4 rr = *reinterpret_cast<const float4 *>(p.residual + (long long)orow * p.cout + col);
      v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
    }
    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (p.l2norm) {
      float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
      for (int o = 1; o < LPR; o <<= 1) ss += __shfl_xor(ss, o, 64);
      const float nrm = sqrtf(ss);
      v.x /= nrm; v.y /= nrm; v.z /= nrm; v.w /= nrm;
    }
    if (orow >= 0) *reinterpret_cast<float4 *>(p.out + (long long)orow * p.cout + col) = v;
  }
}


// ---- variant 3: 128-row workgroup kernel with per-offset row compaction ------------------------
// 8 wavefronts (512 threads) own TWO consecutive rulebook tiles (128 output rows) x one output slab.
// The K walk (offset k, channel chunk cc) is staged through LDS exactly like variant 0 (double
// buffer, one barrier per stage), but the M dimension is COMPACTED per offset: only the rows that
// really have an input at offset k (~52 %) are gathered, so a stage issues ~4.6 full 16-row MFMA
// blocks instead of 8 half-empty ones.  Every wave derives the compaction itself (two coalesced
// neighbour-column loads, two ballots, popcounts) one stage ahead, so the A gather of stage t+1 is
// in flight under the MFMAs of stage t and no extra barrier is needed.  Since compacted rows do not
// line up with fixed accumulator registers, a wave's 16 x CW product block (MFMA with C = 0) is
// added into an LDS-resident [128 x CW] accumulator tile; rows are distinct within a stage and
// stages are separated by the barrier, so plain read-add-write is race-free and the sum order
// (k, cc ascending) is fixed => bit-reproducible.  The epilogue streams whole rows out of LDS.
constexpr int kRowsC = 128;

template <int CO_BLK, int J>
__global__ void __launch_bounds__(512)
k_spconv_c(const ConvParams p) {
  constexpr int CW = 16 * CO_BLK;
  constexpr int SUB_F4 = J * CO_BLK * 64;            // float4 per (k, cc) weight stage
  constexpr int QPT = (SUB_F4 + 511) / 512;          // float4 per thread per stage (1 or 2)
  __shared__ float4 wlds[2][SUB_F4];
  __shared__ __attribute__((aligned(16))) float acc_l[kRowsC][CW];
  __shared__ int scratch[8][2][32];
  __shared__ int klist[kKCache];

  const int y = blockIdx.y, z = blockIdx.z, S = gridDim.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, q4 = lane >> 4;
  const int cin = p.c_a + p.c_b;
  const int ncc = cin / (16 * J);
  const long long slot0 = (long long)blockIdx.x * kRowsC;
  const bool has2 = slot0 + IMF_TILE_ROWS < p.n_slots;      // second tile exists

  uint32_t mask[IMF_MASK_WORDS] = {1u, 0u, 0u, 0u};
  if (p.tile_mask) {
    const long long t0 = blockIdx.x * 2ll;
#pragma unroll
    for (int w = 0; w < IMF_MASK_WORDS; ++w)
      mask[w] = p.tile_mask[t0 * IMF_MASK_WORDS + w] | (has2 ? p.tile_mask[(t0 + 1) * IMF_MASK_WORDS + w] : 0u);
  }
  const int total = __builtin_popcount(mask[0]) + __builtin_popcount(mask[1]) +
                    __builtin_popcount(mask[2]) + __builtin_popcount(mask[3]);
  if (total == 0 && S == 1) return;                  // padding tiles
  const int lo = (int)((long long)z * total / S), hi = (int)((long long)(z + 1) * total / S);
  const int nk = hi - lo;
  if (tid == 0) {
    int ord = 0, n = 0;
#pragma unroll
    for (int w = 0; w < IMF_MASK_WORDS; ++w) {
      uint32_t m = mask[w];
      while (m) {
        const int k = w * 32 + __builtin_ctz(m);
        m &= m - 1;
        if (ord >= lo && ord < hi) klist[n++] = k;
        ++ord;
      }
    }
  }
  {   // zero the accumulator tile
    float4 *a4 = reinterpret_cast<float4 *>(&acc_l[0][0]);
    for (int i = tid; i < kRowsC * CW / 4; i += 512) a4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();

  const float4 *wbase = reinterpret_cast<const float4 *>(p.w_packed) +
                        (long long)y * p.kvol * ncc * SUB_F4;
  const int n_st = nk * ncc;

  float4 w0, w1;                                     // named on purpose (see variant 0)
  float4 a_next[J];
  int rho_next[4];
  bool act_next = false;

  int ir0_pf = -1, ir1_pf = -1;                      // neighbour columns of the next stage to compact
#define IMF_INDEX_C(t)                                                                             \
  {                                                                                                \
    const int kq_ = klist[(t) / ncc];                                                              \
    if (p.nbr) {                                                                                   \
      const int *col_ = p.nbr + (long long)kq_ * p.n_slots + slot0;                                \
      ir0_pf = col_[lane];                                                                         \
      ir1_pf = has2 ? col_[64 + lane] : -1;                                                        \
    } else {                                                                                       \
      ir0_pf = row_of_slot(p, slot0 + lane);                                                       \
      ir1_pf = has2 ? row_of_slot(p, slot0 + 64 + lane) : -1;                                      \
    }                                                                                              \
  }
  // prefetch of stage t: weights -> w0/w1, compaction of offset k, A fragments of this wave's block
#define IMF_PREFETCH_C(t)                                                                          \
  {                                                                                                \
    const int jk_ = (t) / ncc, cc_ = (t) - jk_ * ncc, k_ = klist[jk_];                             \
    const float4 *src_ = wbase + ((long long)k_ * ncc + cc_) * SUB_F4;                             \
    if (QPT == 2 || tid < SUB_F4) w0 = src_[tid];                                                  \
    if (QPT == 2) w1 = src_[512 + tid];                                                            \
    const int ir0_ = ir0_pf, ir1_ = ir1_pf;       /* fetched one stage ago: latency hidden */       \
    if ((t) + 1 < n_st) IMF_INDEX_C((t) + 1)                                                       \
    const unsigned long long m0_ = __ballot(ir0_ >= 0), m1_ = __ballot(ir1_ >= 0);                 \
    const unsigned long long lt_ = (1ull << lane) - 1ull;                                          \
    const int c0_ = __builtin_popcountll(m0_), cnt_ = c0_ + __builtin_popcountll(m1_);            \
    const int p0_ = __builtin_popcountll(m0_ & lt_), p1_ = c0_ + __builtin_popcountll(m1_ & lt_); \
    int *sc_ = scratch[wave][(t) & 1];                                                             \
    if (ir0_ >= 0 && (p0_ >> 4) == wave) { sc_[p0_ & 15] = ir0_; sc_[16 + (p0_ & 15)] = lane; }    \
    if (ir1_ >= 0 && (p1_ >> 4) == wave) { sc_[p1_ & 15] = ir1_; sc_[16 + (p1_ & 15)] = 64 + lane; } \
    __builtin_amdgcn_wave_barrier();                                                               \
    const int base_ = 16 * wave;                                                                   \
    act_next = base_ < cnt_;                                                                       \
    const int my_in_ = (base_ + r16 < cnt_) ? sc_[r16] : -1;                                       \
    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                  \
        rho_next[r] = (base_ + 4 * q4 + r < cnt_) ? sc_[16 + 4 * q4 + r] : -1;                     \
    _Pragma("unroll") for (int j = 0; j < J; ++j)                                                  \
        a_next[j] = gather_a(p, my_in_, cc_ * 16 * J + 16 * j + 4 * q4);                           \
  }

  if (n_st > 0) {
    IMF_INDEX_C(0)
    IMF_PREFETCH_C(0)
  }
#pragma unroll 1
  for (int t = 0; t < n_st; ++t) {
    float4 *wbuf = wlds[t & 1];
    if (QPT == 2 || tid < SUB_F4) wbuf[tid] = w0;
    if (QPT == 2) wbuf[512 + tid] = w1;
    float4 a_cur[J];
    int rho[4];
#pragma unroll
    for (int j = 0; j < J; ++j) a_cur[j] = a_next[j];
#pragma unroll
    for (int r = 0; r < 4; ++r) rho[r] = rho_next[r];
    const bool act = act_next;
    __syncthreads();   // stage t's weights visible; previous stage's accumulator updates complete
    if (t + 1 < n_st) IMF_PREFETCH_C(t + 1)
    if (act) {
      f32x4 d[CO_BLK];
#pragma unroll
      for (int cb = 0; cb < CO_BLK; ++cb) d[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int cb = 0; cb < CO_BLK; ++cb) {
          const float4 b = wbuf[(j * CO_BLK + cb) * 64 + lane];
          d[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].x, b.x, d[cb], 0, 0, 0);
          d[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].y, b.y, d[cb], 0, 0, 0);
          d[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].z, b.z, d[cb], 0, 0, 0);
          d[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[j].w, b.w, d[cb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (rho[r] >= 0) {
          float *dst = &acc_l[rho[r]][r16];
#pragma unroll
          for (int cb = 0; cb < CO_BLK; ++cb) dst[cb * 16] += d[cb][r];
        }
      }
    }
  }
#undef IMF_PREFETCH_C
#undef IMF_INDEX_C
  __syncthreads();

  // ---- epilogue: whole rows out of LDS, float4 per thread -------------------------------------
  constexpr int LPR = CW / 4;                        // threads per row
  constexpr int RPI = 512 / LPR;                     // rows per iteration
  const int c4 = tid % LPR, rsub = tid / LPR;
  const int col = y * CW + 4 * c4;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (S == 1) {
    if (p.scale) sc = *reinterpret_cast<const float4 *>(p.scale + col);
    if (p.shift) sh = *reinterpret_cast<const float4 *>(p.shift + col);
  }
#pragma unroll 1
  for (int it = 0; it < kRowsC / RPI; ++it) {
    const int row = it * RPI + rsub;
    if (slot0 + row >= p.n_slots) continue;          // wave-uniform (rows of a tile stay together)
    float4 v = *reinterpret_cast<const float4 *>(&acc_l[row][4 * c4]);
    if (S > 1) {
      *reinterpret_cast<float4 *>(p.partial + ((long long)z * p.n_slots + slot0 + row) * p.cout + col) = v;
      continue;
    }
    const int orow = row_of_slot(p, slot0 + row);
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    if (p.residual && orow >= 0) {
      const float4 rr = *reinterpret_cast<const float4 *>(p.residual + (long long)orow * p.cout + col);
      v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
    }
    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (p.l2norm) {
      float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
      for (int o = 1; o < LPR; o <<= 1) ss += __shfl_xor(ss, o, 64);
      const float nrm = sqrtf(ss);
      v.x /= nrm; v.y /= nrm; v.z /= nrm; v.w /= nrm;
    }
    if (orow >= 0) *reinterpret_cast<float4 *>(p.out + (long long)orow * p.cout + col) = v;
  }
}


// ---- variant 4: barrier-free register kernel ---------------------------------------------------
// Measured (tools/conv_tiles.py, conv_ablate.py): in the LDS-staged kernels the co-resident
// workgroups of a CU run in lock-step, so the per-stage barrier phase idles the matrix pipe on all
// four SIMDs at once (46-50 cycles per MFMA instead of the 32 the pipe sustains).  Here every
// wavefront is autonomous: it owns RB 16-row blocks x one output slab, keeps the accumulators in
// registers, and streams BOTH operands straight from L1/L2 -- A rows gathered as before, B
// fragments as coalesced 1 KiB loads of the fragment-major weight image (all waves of a CU walk
// the offsets at the same pace, so the 16 KiB of an offset's weights are L1 hits after the first
// touch).  The (offset, 16-channel step) space is walked as one flat software pipeline with a
// look-ahead of one step; no LDS, no barriers, no atomics; sum order fixed (k, channel ascending).
template <int CO_BLK, int RB>
__global__ void __launch_bounds__(256)
k_spconv_reg(const ConvParams p) {
  const int y = blockIdx.y, z = blockIdx.z, S = gridDim.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, q4 = lane >> 4;
  const int cin = p.c_a + p.c_b;
  const int U = cin / 16;                                     // 16-channel steps per offset
  const long long gw = (long long)blockIdx.x * 4 + wave;      // global wave = RB consecutive row blocks
  const long long slot_w = gw * 16 * RB;
  if (slot_w >= p.n_slots) return;
  const long long tile = slot_w / IMF_TILE_ROWS;

  uint32_t mask[IMF_MASK_WORDS] = {1u, 0u, 0u, 0u};
  if (p.tile_mask) {
#pragma unroll
    for (int w = 0; w < IMF_MASK_WORDS; ++w) mask[w] = p.tile_mask[tile * IMF_MASK_WORDS + w];
  }
  const int total = __builtin_popcount(mask[0]) + __builtin_popcount(mask[1]) +
                    __builtin_popcount(mask[2]) + __builtin_popcount(mask[3]);
  if (total == 0 && S == 1) return;
  const int lo = (int)((long long)z * total / S), hi = (int)((long long)(z + 1) * total / S);
  const int nk = hi - lo;
  unsigned long long kp0 = 0ull, kp1 = 0ull, kp2 = 0ull;      // offsets of this partition, 7 bits each
  {
    int ord = 0, n = 0;
#pragma unroll
    for (int w = 0; w < IMF_MASK_WORDS; ++w) {
      uint32_t m = mask[w];
      while (m) {
        const unsigned long long k = (unsigned long long)(w * 32 + __builtin_ctz(m));
        m &= m - 1;
        if (ord >= lo && ord < hi) {
          if (n < 9) kp0 |= k << (7 * n);
          else if (n < 18) kp1 |= k << (7 * (n - 9));
          else kp2 |= k << (7 * (n - 18));
          ++n;
        }
        ++ord;
      }
    }
  }
#define IMF_KGET(jk) ((int)(((jk) < 9 ? kp0 >> (7 * (jk)) : ((jk) < 18 ? kp1 >> (7 * ((jk)-9)) : kp2 >> (7 * ((jk)-18)))) & 127ull))

  f32x4 acc[RB][CO_BLK];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CO_BLK; ++cb) acc[rb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float4 *wslab = reinterpret_cast<const float4 *>(p.w_packed) +
                        (long long)y * p.kvol * U * (CO_BLK * 64) + lane;

  // rows of offset jk (lane l holds the input row of output row rb*16 + (l & 15))
#define IMF_LOAD_ROWS(dst, jk)                                                                     \
  {                                                                                                \
    _Pragma("unroll") for (int rb = 0; rb < RB; ++rb) {                                           \
      const long long sl_ = slot_w + rb * 16 + r16;                                                \
      dst[rb] = p.nbr ? p.nbr[(long long)IMF_KGET(jk) * p.n_slots + sl_] : row_of_slot(p, sl_);    \
    }                                                                                              \
  }
#define IMF_LOAD_STEP(A, B, rows, kk, u)                                                           \
  {                                                                                                \
    const float4 *bp_ = wslab + ((long long)(kk) * U + (u)) * (CO_BLK * 64);                       \
    _Pragma("unroll") for (int cb = 0; cb < CO_BLK; ++cb) B[cb] = bp_[cb * 64];                    \
    _Pragma("unroll") for (int rb = 0; rb < RB; ++rb) A[rb] = gather_a(p, rows[rb], 16 * (u) + 4 * q4); \
  }
#define IMF_MFMA_STEP(A, B)                                                                        \
  {                                                                                                \
    _Pragma("unroll") for (int rb = 0; rb < RB; ++rb)                                             \
    _Pragma("unroll") for (int cb = 0; cb < CO_BLK; ++cb) {                                       \
      acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rb].x, B[cb].x, acc[rb][cb], 0, 0, 0); \
      acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rb].y, B[cb].y, acc[rb][cb], 0, 0, 0); \
      acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rb].z, B[cb].z, acc[rb][cb], 0, 0, 0); \
      acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rb].w, B[cb].w, acc[rb][cb], 0, 0, 0); \
    }                                                                                              \
  }

  // flat pipeline over (jk, u): "cur" is being multiplied while "nxt" is in flight
  int rows_cur[RB], rows_nxt[RB];
  float4 A0[RB], B0[CO_BLK], A1[RB], B1[CO_BLK];
  int jk = 0;
  // find the first offset with any valid row in this wave
  bool have = false;
  while (jk < nk) {
    IMF_LOAD_ROWS(rows_cur, jk)
    bool any = false;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) any |= rows_cur[rb] >= 0;
    if (__any(any)) { have = true; break; }
    ++jk;
  }
  if (have) {
    int k = IMF_KGET(jk), u = 0;
    IMF_LOAD_STEP(A0, B0, rows_cur, k, 0)
    bool more = true;
    int njk = jk;                                              // offset the next step belongs to
    bool rows_nxt_valid = false;
#pragma unroll 1
    while (more) {
      // ---- issue the loads of the next step -------------------------------------------------
      int nu = u + 1, nk_ = k;
      bool next_ok = true;
      if (nu == U) {                                           // move on to the next non-empty offset
        nu = 0;
        next_ok = false;
        njk = jk + 1;
        while (njk < nk) {
          IMF_LOAD_ROWS(rows_nxt, njk)
          bool any = false;
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) any |= rows_nxt[rb] >= 0;
          if (__any(any)) { next_ok = true; break; }
          ++njk;
        }
        if (next_ok) nk_ = IMF_KGET(njk);
        rows_nxt_valid = next_ok;
      }
      if (next_ok) {
        if (nu == 0) { IMF_LOAD_STEP(A1, B1, rows_nxt, nk_, 0) }
        else { IMF_LOAD_STEP(A1, B1, rows_cur, nk_, nu) }
      }
      // ---- multiply the current step --------------------------------------------------------
      IMF_MFMA_STEP(A0, B0)
      // ---- rotate ----------------------------------------------------------------------------
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) A0[rb] = A1[rb];
#pragma unroll
      for (int cb = 0; cb < CO_BLK; ++cb) B0[cb] = B1[cb];
      if (nu == 0 && rows_nxt_valid) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) rows_cur[rb] = rows_nxt[rb];
        jk = njk;
        rows_nxt_valid = false;
      }
      u = nu;
      k = nk_;
      more = next_ok;
    }
  }
#undef IMF_LOAD_ROWS
#undef IMF_LOAD_STEP
#undef IMF_MFMA_STEP
#undef IMF_KGET

#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const long long slot_b = slot_w + rb * 16;                 // this 16-row block
    const int tl = (int)(slot_b / IMF_TILE_ROWS), wv = (int)((slot_b % IMF_TILE_ROWS) / 16);
    if (S == 1) {
      conv_epilogue<CO_BLK>(p, acc[rb], tl, y, wv, r16, q4);
    } else {
      const int CW = 16 * CO_BLK;
#pragma unroll
      for (int cb = 0; cb < CO_BLK; ++cb) {
        const int col = y * CW + cb * 16 + r16;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          p.partial[((long long)z * p.n_slots + slot_b + q4 * 4 + r) * p.cout + col] = acc[rb][cb][r];
      }
    }
  }
}

