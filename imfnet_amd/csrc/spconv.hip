// Sparse convolution on gfx950: output-stationary gather -> fp32 MFMA small-GEMM per rulebook
// tile, fused BatchNorm / bias / residual / ReLU / L2-norm epilogue.
//
// Work decomposition (one workgroup = 4 wavefronts = one 64-row rulebook tile x one 32/64-wide
// slab of output channels x one partition of the tile's active kernel offsets):
//   wavefront w owns output rows [16w, 16w+16) of the tile and all CO_BLK 16-column blocks of the
//   slab: CO_BLK accumulators of v_mfma_f32_16x16x4_f32 (4 VGPRs each).
//   The K dimension (kernel offset k, input-channel chunk cc) is walked in 16 KiB "macro stages" of
//   packed weights.  Per macro stage:
//     - the weights (already in MFMA B-fragment order in HBM/L2) are prefetched global->VGPR one
//       stage ahead and written to one of two LDS buffers: ONE barrier per stage, loads of stage
//       n+1 in flight under the MFMAs of stage n;
//     - every lane gathers its A fragments straight from the input rows as float4 (lane l holds
//       channels 16j + 4(l>>4) + 0..3 of row nbr[k][l&15]) -- no LDS round trip for A; the tile's
//       slice of the neighbour table is cached in LDS once;
//     - a wavefront whose 16 rows have no input at offset k skips the MFMAs (wave-uniform).
//   Levels with few tiles are latency-bound, so their offsets are split over gridDim.z workgroups
//   that write raw partial sums; k_spconv_reduce adds them in a fixed order and applies the
//   epilogue.  The accumulation order per output element is fixed => bit-reproducible, no atomics.
#include <stdlib.h>
#include <string.h>

#include "spconv_shared.h"
#include "geometry_internal.h"
#include "rulebook_tile.h"

namespace imf {

__global__ void __launch_bounds__(256)
k_pack_weights(const float *__restrict__ w, int kvol, int cin, int cout, float *__restrict__ packed) {
  const long long total = (long long)kvol * cin * cout;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int CI = ci_chunk_of(cin), J = CI / 16, CB = co_blk_of(cout), CW = 16 * CB;
  const int ncc = cin / CI;
  long long r = idx;
  const int t = r & 3; r >>= 2;
  const int lane = r & 63; r >>= 6;
  const int cb = r % CB; r /= CB;
  const int j = r % J; r /= J;
  const int cc = r % ncc; r /= ncc;
  const int k = r % kvol; r /= kvol;
  const int y = (int)r;
  const int ci = cc * CI + 16 * j + 4 * (lane >> 4) + t;
  const int co = y * CW + 16 * cb + (lane & 15);
  packed[idx] = w[((long long)k * cin + ci) * cout + co];
}

// ---- variant 1: simple reference kernel (single LDS buffer, two barriers per stage) -----------
template <int CO_BLK, int J>
__global__ void __launch_bounds__(256)
k_spconv_mfma_simple(const ConvParams p) {
  constexpr int STAGE_F4 = J * CO_BLK * 64;          // float4 per weight stage (<= 1024 = 16 KiB)
  __shared__ float4 wlds[STAGE_F4];

  const int tile = blockIdx.x, y = blockIdx.y;
  if (p.n_out_dev && (long long)tile * IMF_TILE_ROWS >= conv_slots(p, conv_rows(p))) return;   // capacity mode: beyond the rows
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, q4 = lane >> 4;
  const int cin = p.c_a + p.c_b;
  const int ncc = cin / (16 * J);

  uint32_t mask[IMF_MASK_WORDS] = {1u, 0u, 0u, 0u};
  if (p.tile_mask) {
#pragma unroll
    for (int w = 0; w < IMF_MASK_WORDS; ++w) mask[w] = p.tile_mask[tile * IMF_MASK_WORDS + w];
  }
  if ((mask[0] | mask[1] | mask[2] | mask[3]) == 0u) return;   // padding tile

  const long long my_slot = (long long)tile * IMF_TILE_ROWS + wave * 16 + r16;

  f32x4 acc[CO_BLK];
#pragma unroll
  for (int cb = 0; cb < CO_BLK; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float4 *wbase = reinterpret_cast<const float4 *>(p.w_packed) +
                        (long long)y * p.kvol * ncc * STAGE_F4;

#pragma unroll 1
  for (int w = 0; w < IMF_MASK_WORDS; ++w) {
    uint32_t m = mask[w];
#pragma unroll 1
    while (m) {
      const int k = w * 32 + __builtin_ctz(m);
      m &= m - 1;
      const int irow = p.nbr ? p.nbr[(long long)k * p.n_slots + my_slot] : row_of_slot(p, my_slot);
      const bool wave_active = __any(irow >= 0);
#pragma unroll 1
      for (int cc = 0; cc < ncc; ++cc) {
        __syncthreads();                      // previous stage fully consumed
        const float4 *src = wbase + ((long long)k * ncc + cc) * STAGE_F4;
#pragma unroll
        for (int q = 0; q < STAGE_F4 / 256; ++q) wlds[q * 256 + tid] = src[q * 256 + tid];

        float4 a[J];
        if (wave_active) {
#pragma unroll
          for (int j = 0; j < J; ++j) a[j] = gather_a(p, irow, cc * 16 * J + 16 * j + 4 * q4);
        }
        __syncthreads();                      // stage visible
        if (wave_active) {
#pragma unroll
          for (int j = 0; j < J; ++j) {
#pragma unroll
            for (int cb = 0; cb < CO_BLK; ++cb) {
              const float4 b = wlds[(j * CO_BLK + cb) * 64 + lane];
              acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b.x, acc[cb], 0, 0, 0);
              acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b.y, acc[cb], 0, 0, 0);
              acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b.z, acc[cb], 0, 0, 0);
              acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b.w, acc[cb], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  conv_epilogue<CO_BLK>(p, acc, tile, y, wave, r16, q4);
}

// ---- variant 0: pipelined kernel ------------------------------------------------------------

template <int CO_BLK, int J>
__global__ void __launch_bounds__(256)
k_spconv_mfma(const ConvParams p) {
  constexpr int SUB_F4 = J * CO_BLK * 64;            // float4 per (k, cc) sub-stage
  constexpr int KG = 1024 / SUB_F4;                  // sub-stages per 16 KiB macro stage: 1, 2 or 4
  constexpr int QPS = SUB_F4 / 256;                  // float4 per thread per sub-stage: 4, 2 or 1
  __shared__ float4 wlds[2][1024];
  __shared__ int nbr_lds[kKCache][IMF_TILE_ROWS];
  __shared__ int klist[kKCache];

  const int tile = blockIdx.x, y = blockIdx.y, z = blockIdx.z, S = gridDim.z;
  if (p.n_out_dev && (long long)tile * IMF_TILE_ROWS >= conv_slots(p, conv_rows(p))) return;   // capacity mode: beyond the rows
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, q4 = lane >> 4;
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
  __syncthreads();
  const long long tile_slot0 = (long long)tile * IMF_TILE_ROWS;
  for (int j = wave; j < nk; j += 4)
    nbr_lds[j][lane] = p.nbr ? p.nbr[(long long)klist[j] * p.n_slots + tile_slot0 + lane]
                             : row_of_slot(p, tile_slot0 + lane);
  __syncthreads();

  f32x4 acc[CO_BLK];
#pragma unroll
  for (int cb = 0; cb < CO_BLK; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float4 *wbase = reinterpret_cast<const float4 *>(p.w_packed) +
                        (long long)y * p.kvol * ncc * SUB_F4;
  const int n_sub = nk * ncc;
  const int n_macro = (n_sub + KG - 1) / KG;

  // Prefetch registers.  The four weight quads are NAMED scalars on purpose: as an array they are
  // left in scratch memory by hipcc (ROCm 7.2), which serialises the prefetch behind vmcnt waits.
  float4 w0, w1, w2, w3;
  float4 a_next[KG][J];
  const float4 *sp[KG];

  // prefetch of macro stage n: weights -> w0..w3, A fragments -> a_next
#define IMF_PREFETCH(n)                                                                           \
  {                                                                                                \
    _Pragma("unroll") for (int g = 0; g < KG; ++g) {                                              \
      const int t = (n) * KG + g;                                                                  \
      const int tc = t < n_sub ? t : n_sub - 1;   /* tail: reload the last sub-stage, unused */    \
      const int jk = tc / ncc, cc = tc - jk * ncc;                                                 \
      sp[g] = wbase + ((long long)klist[jk] * ncc + cc) * SUB_F4 + tid;                            \
      const int irow = (t < n_sub) ? nbr_lds[jk][wave * 16 + r16] : -1;                            \
      /* a (k, cc) chunk never straddles the two cat sources: c_a % (16 J) == 0 (host-checked) */   \
      const int ch0 = cc * 16 * J;                                                                 \
      const float *rowp = (ch0 < p.c_a) ? p.in_a + (long long)irow * p.c_a + ch0                   \
                                        : p.in_b + (long long)irow * p.c_b + (ch0 - p.c_a);        \
      if (irow >= 0) {                                                                             \
        _Pragma("unroll") for (int j = 0; j < J; ++j)                                              \
            a_next[g][j] = *reinterpret_cast<const float4 *>(rowp + 16 * j + 4 * q4);              \
      } else {                                                                                     \
        _Pragma("unroll") for (int j = 0; j < J; ++j) a_next[g][j] = make_float4(0.f, 0.f, 0.f, 0.f); \
      }                                                                                            \
    }                                                                                              \
    w0 = sp[0 / QPS][(0 % QPS) * 256];                                                             \
    w1 = sp[1 / QPS][(1 % QPS) * 256];                                                             \
    w2 = sp[2 / QPS][(2 % QPS) * 256];                                                             \
    w3 = sp[3 / QPS][(3 % QPS) * 256];                                                             \
  }

  if (n_macro > 0) IMF_PREFETCH(0)
#pragma unroll 1
  for (int n = 0; n < n_macro; ++n) {
    float4 *wbuf = wlds[n & 1];
    wbuf[0 * 256 + tid] = w0;
    wbuf[1 * 256 + tid] = w1;
    wbuf[2 * 256 + tid] = w2;
    wbuf[3 * 256 + tid] = w3;
    float4 a_cur[KG][J];
#pragma unroll
    for (int g = 0; g < KG; ++g) {
#pragma unroll
      for (int j = 0; j < J; ++j) a_cur[g][j] = a_next[g][j];
    }
    __syncthreads();   // stage n visible; every wave is past its reads of this buffer (stage n-2)
    if (n + 1 < n_macro) IMF_PREFETCH(n + 1)
#pragma unroll
    for (int g = 0; g < KG; ++g) {     // unconditional: empty rows carry zeros (a skipped tail adds 0)
#pragma unroll
      for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int cb = 0; cb < CO_BLK; ++cb) {
          const float4 b = wbuf[g * SUB_F4 + (j * CO_BLK + cb) * 64 + lane];
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[g][j].x, b.x, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[g][j].y, b.y, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[g][j].z, b.z, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[g][j].w, b.w, acc[cb], 0, 0, 0);
        }
      }
    }
  }
#undef IMF_PREFETCH

  if (S == 1) {
    conv_epilogue<CO_BLK>(p, acc, tile, y, wave, r16, q4);
  } else {   // raw partial sums, slot-major
    const int CW = 16 * CO_BLK;
#pragma unroll
    for (int cb = 0; cb < CO_BLK; ++cb) {
      const int col = y * CW + cb * 16 + r16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long slot = tile_slot0 + wave * 16 + q4 * 4 + r;
        p.partial[((long long)z * p.n_slots + slot) * p.cout + col] = acc[cb][r];
      }
    }
    if (p.tickets) {
      // In-launch split-K combine, placement-independent (cdna_hip_programming.md G16, counter form):
      // every storing wave drains, ONE lane releases at agent scope and takes a ticket; the last
      // arriver acquires at agent scope, then all its waves read the S slabs with plain loads.
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


// Adds the split-K partial sums in ascending partition order and applies the epilogue.
// One thread per (slot, 4 output channels).
__global__ void __launch_bounds__(256)
k_spconv_reduce(const ConvParams p, int S, long long slot0) {
  // slots [slot0, n_slots): the whole table for split-K, the balanced tail's tiles otherwise
  if (p.n_out_dev && p.dyn_split_kvol) {   // capacity mode: the main kernel chose the split from the actual rows
    const int cover = S;
    S = auto_split_rule(conv_slots(p, conv_rows(p)), p.cout, p.dyn_split_kvol, p.split_min_blocks, p.split_target);
    S = S > cover ? cover : S;
    if (S == 1) return;                    // unsplit: the main kernel already wrote the output
  }
  const int c4n = p.cout / 4;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long rel = idx / c4n, slot = slot0 + rel, nrel = p.n_slots - slot0;
  const int c4 = (int)(idx - rel * c4n);
  const bool in_range = slot < p.n_slots;
  const int orow = in_range ? row_of_slot(p, slot) : -1;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (orow >= 0) {
    for (int zz = 0; zz < S; ++zz) {
      const float4 v = *reinterpret_cast<const float4 *>(
          p.partial + ((long long)zz * nrel + rel) * p.cout + 4 * c4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float x[4] = {s.x, s.y, s.z, s.w};
    const float un = p.w_unscale ? *p.w_unscale : 1.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int col = 4 * c4 + e;
      float v = (x[e] * un) * (p.scale ? p.scale[col] : 1.f) + (p.shift ? p.shift[col] : 0.f);
      if (p.residual) v += p.residual[(long long)orow * p.cout + col];
      if (p.relu) v = fmaxf(v, 0.f);
      x[e] = v;
    }
    s = make_float4(x[0], x[1], x[2], x[3]);
    if (range_guard(p) && (out_of_f16_range(s.x) || out_of_f16_range(s.y) || out_of_f16_range(s.z) || out_of_f16_range(s.w)))
      atomicOr(p.err, 32);
  }
  if (p.l2norm) {   // cout in {32, 64}: a row = 8 or 16 consecutive lanes (all lanes take part)
    float ss = s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w;
    for (int o = 1; o < c4n; o <<= 1) ss += __shfl_xor(ss, o, 64);
    const float nrm = sqrtf(ss);
    s.x /= nrm; s.y /= nrm; s.z /= nrm; s.w /= nrm;
  }
  if (orow >= 0) *reinterpret_cast<float4 *>(p.out + (long long)orow * p.cout + 4 * c4) = s;
}

// ---- first layer: tiny Cin (all-ones occupancy feature), one thread per output row -------------
template <int COUT>
__global__ void __launch_bounds__(256)
k_spconv_small_cin(const float *__restrict__ in, int cin, const float *__restrict__ w, int kvol,
                   const int32_t *__restrict__ nbr, long long n_slots, long long n_out,
                   const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                   float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const int nw = kvol * cin * COUT;
  for (int i = threadIdx.x; i < nw; i += blockDim.x) wl[i] = w[i];
  __syncthreads();
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
  for (int k = 0; k < kvol; ++k) {
    const int i = (row < n_out) ? nbr[(long long)k * n_slots + row] : -1;
    if (i >= 0) {
      for (int ci = 0; ci < cin; ++ci) {
        const float x = in[(long long)i * cin + ci];
        const float4 *wk = reinterpret_cast<const float4 *>(wl + (k * cin + ci) * COUT);
#pragma unroll
        for (int c4 = 0; c4 < COUT / 4; ++c4) {
          const float4 ww = wk[c4];
          acc[4 * c4 + 0] = fmaf(x, ww.x, acc[4 * c4 + 0]);
          acc[4 * c4 + 1] = fmaf(x, ww.y, acc[4 * c4 + 1]);
          acc[4 * c4 + 2] = fmaf(x, ww.z, acc[4 * c4 + 2]);
          acc[4 * c4 + 3] = fmaf(x, ww.w, acc[4 * c4 + 3]);
        }
      }
    }
  }
  if (row >= n_out) return;
  float4 *o = reinterpret_cast<float4 *>(out + row * COUT);
#pragma unroll
  for (int c4 = 0; c4 < COUT / 4; ++c4) {
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * c4 + e;
      float x = acc[c] * (scale ? scale[c] : 1.f) + (shift ? shift[c] : 0.f);
      y[e] = relu ? fmaxf(x, 0.f) : x;
    }
    o[c4] = make_float4(y[0], y[1], y[2], y[3]);
  }
}


// ---- first layer, fused with its kernel map: no neighbour table is ever written to HBM ----------
// One workgroup = 32 output voxels.  Phase 1: the 256 threads probe the 32 x kvol kernel offsets in
// the input level's hash (16 independent probes per thread) into an LDS neighbour tile.  Phase 2:
// thread = (output channel, row group) walks the offsets in ascending k -- the same sum order as the
// table-driven kernel and the oracle -- with the weights in LDS.  in == nullptr: all-ones input.
constexpr int kFirstRows = 32;

template <int COUT>
__global__ void __launch_bounds__(256)
k_conv_first_fused(const imf_slot *__restrict__ tab, uint32_t capmask,
                   const int32_t *__restrict__ coords, long long n, int ts, int ksize, int kvol,
                   const float *__restrict__ in, int cin, const float *__restrict__ w,
                   const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                   float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float wl[];      // [kvol*cin*COUT] weights, then nbr tile
  const int nw = kvol * cin * COUT;
  int *nbr_l = reinterpret_cast<int *>(wl + nw);                   // [kFirstRows][128]
  const int tid = threadIdx.x;
  for (int i = tid; i < nw; i += 256) wl[i] = w[i];
  const long long row0 = (long long)blockIdx.x * kFirstRows;
  const int r = ksize >> 1;
  // 16 probes per thread, issued as one independent batch of 16-byte slot loads (key + row together);
  // only a collision (rare: the level-0 table is <= 25 % full) falls back to the probe loop.
  constexpr int NP = kFirstRows * 128 / 256;
  uint64_t want[NP];
  uint4 got[NP];
  uint32_t hs[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int idx = j * 256 + tid, lr = idx >> 7, k = idx & 127;
    const long long row = row0 + lr;
    want[j] = kEmptyKey;                               // "no probe": resolves to -1 below
    hs[j] = 0;
    if (k < kvol && row < n) {
      const int4 c = reinterpret_cast<const int4 *>(coords)[row];
      const int dx = k % ksize - r, dy = (k / ksize) % ksize - r, dz = k / (ksize * ksize) - r;
      const int x = c.y + dx * ts, y = c.z + dy * ts, z = c.w + dz * ts;
      if (coord_in_range(x, y, z)) {
        want[j] = pack_key(c.x, x, y, z);
        hs[j] = hash_slot(want[j], __builtin_ctz((unsigned)ts), capmask);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NP; ++j) got[j] = *reinterpret_cast<const uint4 *>(tab + hs[j]);
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    int found = -1;
    if (want[j] != kEmptyKey) {
      const uint64_t k0 = ((uint64_t)got[j].y << 32) | got[j].x;
      if (k0 == want[j]) found = (int)got[j].z;
      else if (k0 != kEmptyKey) found = hash_find(tab, capmask, want[j], __builtin_ctz((unsigned)ts));   // collision: slow path
    }
    nbr_l[j * 256 + tid] = found;
  }
  __syncthreads();
  constexpr int RPT = kFirstRows * COUT / 256;                     // rows per thread: 4 (cout 32) / 8 (64)
  constexpr int RG = 256 / COUT;                                   // row groups
  const int co = tid % COUT, rg = tid / COUT;
  float acc[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) acc[q] = 0.f;
  for (int k = 0; k < kvol; ++k) {
    int idx[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) idx[q] = nbr_l[(rg + q * RG) * 128 + k];
    for (int ci = 0; ci < cin; ++ci) {
      const float wv = wl[(k * cin + ci) * COUT + co];
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        if (in) {
          if (idx[q] >= 0) acc[q] = fmaf(in[(long long)idx[q] * cin + ci], wv, acc[q]);
        } else {
          acc[q] += idx[q] >= 0 ? wv : 0.f;                        // x == 1: exact, branch-free
        }
      }
    }
  }
  const float sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const long long row = row0 + rg + q * RG;
    if (row < n) {
      float v = acc[q] * sc + sh;
      if (relu) v = fmaxf(v, 0.f);
      out[row * COUT + co] = v;
    }
  }
}


// ---- first layer on an occupancy bit grid (all-ones input feature) -----------------------------
// util/misc.py:76-79 feeds the network a column of ones, so conv1 is "sum of the weight rows of the
// occupied offsets".  Occupancy of a 5x5x5 neighbourhood is 25 five-bit windows of a dense bit grid
// over the fragment's bounding box (0.7 MB for a 3DMatch fragment, L2-resident) instead of 125
// dependent probes into a multi-MB hash table.  Per workgroup (64 voxels): the windows are expanded
// into 128-bit masks and out = A . W runs on the f16 matrix pipe (the 0 / 1 operand is exact in f16, the weights
// are split hi + lo) with the folded BatchNorm epilogue.
// GridDesc, grid_desc_from_bbox, DynGrid, dyn_grid, grid_row: geometry_internal.h (the level-0 compaction kernel fills the grid too)

__global__ void __launch_bounds__(256)
k_bitgrid_fill(const int32_t *__restrict__ coords, long long n, uint32_t *grid, GridDesc g, int ksize,
               const DynGrid dg) {
  if (!dyn_grid(dg, ksize, g, n)) return;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = reinterpret_cast<const int4 *>(coords)[i];
  const int bit = c.y - g.x0;
  atomicOr(grid + grid_row(g, c.x, c.z, c.w) + (bit >> 5), 1u << (bit & 31));
}

constexpr int kBitsRows = 64;

typedef _Float16 f16x8_b __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2_b __attribute__((ext_vector_type(2)));

// conv1's weights as f16 B fragments: max |w| (block reduce over 256 threads) -> power-of-two scale -> THREE f16 parts per weight,
// w = p0 + p1 + p2 exactly (3 x 11 significant bits >= fp32's 24; round 3 kept two parts = 22 bits).  conv1's left operand is
// the 0 / 1 occupancy, exact in f16, and the matrix pipe accumulates in fp32: with exact weights conv1 IS fp32 arithmetic --
// in every mode, for 8 more MFMAs per wavefront.  Writes [nkc][CBN][part][64 lanes] float4 (8 halves each) to `W_l` (LDS or
// global) and returns the factor that undoes the scale.
constexpr int kFirstParts = 3;
template <int COUT>
__device__ __forceinline__ float first_kernel_split(const float *__restrict__ w, int kvol, int nkc, float4 *W_l, unsigned *red,
                                                    int tid) {
  constexpr int CBN = COUT / 16;
  const int wave = tid >> 6, lane = tid & 63;
  unsigned amax = 0u;
  for (int i = tid; i < kvol * COUT; i += 256) {
    const unsigned bits = __float_as_uint(fabsf(w[i]));
    if (bits < 0x7F800000u) amax = bits > amax ? bits : amax;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned t = __shfl_xor(amax, o, 64);
    amax = t > amax ? t : amax;
  }
  if (lane == 0) red[wave] = amax;
  __syncthreads();
  amax = max(max(red[0], red[1]), max(red[2], red[3]));
  int wshift = 0;
  if (amax != 0u) {
    wshift = 13 - ((int)(amax >> 23) - 127);             // the scaled kernel peaks in [2^13, 2^14): lo halves stay normal
    wshift = wshift < -40 ? -40 : (wshift > 100 ? 100 : wshift);
  }
  for (int i = tid; i < nkc * CBN * 64; i += 256) {
    const int ln = i & 63, cb = (i >> 6) % CBN, kc = i / (64 * CBN);
    f16x8_b p0, p1, p2;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int k = 32 * kc + 16 * (t >> 2) + 4 * (ln >> 4) + (t & 3);
      const float x = k < kvol ? ldexpf(w[k * COUT + 16 * cb + (ln & 15)], wshift) : 0.f;
      const _Float16 h0 = (_Float16)x;
      const float r1 = x - (float)h0;                  // exact
      const _Float16 h1 = (_Float16)r1;
      p0[t] = h0;
      p1[t] = h1;
      p2[t] = (_Float16)(r1 - (float)h1);              // exact while normal (>= 2^-24 after the scale: |w| >= 2^-17 max |w|)
    }
    W_l[((kc * CBN + cb) * kFirstParts + 0) * 64 + ln] = __builtin_bit_cast(float4, p0);
    W_l[((kc * CBN + cb) * kFirstParts + 1) * 64 + ln] = __builtin_bit_cast(float4, p1);
    W_l[((kc * CBN + cb) * kFirstParts + 2) * 64 + ln] = __builtin_bit_cast(float4, p2);
  }
  return ldexpf(1.f, -wshift);
}

template <int COUT>
__global__ void __launch_bounds__(256) k_pack_first_kernel(const float *__restrict__ w, int kvol, float *__restrict__ image) {
  __shared__ unsigned red[4];
  const int nkc = (kvol + 31) >> 5;
  const float un = first_kernel_split<COUT>(w, kvol, nkc, reinterpret_cast<float4 *>(image), red, threadIdx.x);
  if (threadIdx.x == 0) image[(size_t)nkc * (COUT / 16) * kFirstParts * 64 * 4] = un;
}

// conv1 for the all-ones occupancy feature: out[v] = sum_k occ(v + off_k) * W[k], a [64, kvol] x [kvol, COUT] product per
// workgroup whose left operand is BINARY.  Round 3: the occupancy window of a voxel is kept as a 128-bit mask (one thread
// per (voxel, 32-offset word): no LDS atomics, no 33 KiB float matrix, no bank conflicts) and expanded to f16 0 / 1
// A fragments in registers; 0 and 1 are exact in f16, so only the WEIGHTS are split (hi + lo halves, pre-scaled by a
// power of two like imf_pack_weights_split16): 2 x v_mfma_f32_16x16x32_f16 per 32 offsets and column block instead of
// 8 x v_mfma_f32_16x16x4_f32 -- 16 matrix instructions of 16 cycles per wavefront instead of 64 of 32.  The weight split
// is redone by every workgroup (4 000 values from L2): no packed image, no change to the C ABI.
template <int COUT, int KS>
__device__ __forceinline__ void conv_first_bits_body(const int32_t *__restrict__ coords, long long n,
                                                     const uint32_t *__restrict__ grid, GridDesc g, int ksize_rt, int kvol,
                                                     const float *__restrict__ w, const float *__restrict__ scale,
                                                     const float *__restrict__ shift, int relu, float *__restrict__ out,
                                                     const DynGrid dg, int out_split, const long long blk) {
  constexpr int CBN = COUT / 16;                         // column blocks; wave w owns row block w
  constexpr int ksize = KS;
  constexpr int kMaxWin = KS == 3 ? 11 : 8;              // windows of KS bits that can touch one 32-offset word
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  (void)ksize_rt;
  if (!dyn_grid(dg, ksize, g, n)) return;
  if (blk * kBitsRows >= n) return;
  float4 *W_l = reinterpret_cast<float4 *>(lds_f);                          // [4 kc][CBN][3 parts][64 lanes] x 8 halves
  uint32_t *M_l = reinterpret_cast<uint32_t *>(W_l + 4 * CBN * kFirstParts * 64);     // [64 rows][4 words]: occupancy masks
  __shared__ unsigned red[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long long v0 = blk * kBitsRows;
  const int nkc = (kvol + 31) >> 5;

  // ---- occupancy masks: thread (v, wd) gathers the windows (dy, dz) whose ksize bits fall into offsets 32 wd .. 32 wd + 31
  const int v = tid >> 2, wd = tid & 3;
  const int r = ksize >> 1;
  const long long row = v0 + v < n ? v0 + v : n - 1;     // clamped: the loads are unconditional
  const int4 c = reinterpret_cast<const int4 *>(coords)[row];
  const int k_lo = 32 * wd, k_hi = min(32 * wd + 31, kvol - 1);
  const int yz_lo = k_lo / ksize;
  const int n_win = k_hi >= k_lo ? k_hi / ksize - yz_lo + 1 : 0;          // <= kMaxWin
  const int bx = c.y - r - g.x0;                         // first bit of every window of this voxel, >= 0 by construction
  const int wi = bx >> 5, sh = bx & 31;
  const __amdgpu_buffer_rsrc_t rs_grid = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(grid), (short)0, 0x7FFFFFFF, 0x00020000);
  uint32_t w0[kMaxWin], w1[kMaxWin];
#pragma unroll
  for (int j = 0; j < kMaxWin; ++j) {
    const int yz = j < n_win ? yz_lo + j : (n_win ? yz_lo : 0);   // unused slots repeat a valid window (the load is unconditional)
    const int dy = yz % ksize - r, dz = yz / ksize - r;
    // both words of the window in ONE 8-byte buffer load (dword-aligned is enough for buffer addressing): a row has
    // nx / 32 + 2 words and a window starts at most ksize bits before its last occupied bit, so word wi + 1 is in the row
    const long long off = (grid_row(g, c.x, c.z + dy, c.w + dz) + wi) * 4;
    const u32x2_b pr = __builtin_amdgcn_raw_buffer_load_b64(rs_grid, (int)off, 0, 0);
    w0[j] = pr[0];
    w1[j] = pr[1];
  }
  // ---- weights: the hi / lo f16 B fragments in the MFMA's lane order, [kc][cb][h][lane][t]: offset k = 32 kc + 16 (t >> 2) +
  //      4 (lane >> 4) + (t & 3), column 16 cb + (lane & 15).  From the image imf_pack_first_kernel wrote once per model
  //      (16 KiB verbatim: round 4 -- every one of the ~1 600 workgroups of a launch used to redo the 4 096 splits, 1.5 k of its
  //      1.7 k VALU instructions per wavefront), or, without one, split here: max |w| -> power-of-two scale -> hi / lo.
  float un;
  if (dg.w_image) {
    const float4 *img = reinterpret_cast<const float4 *>(dg.w_image);
    for (int i = tid; i < nkc * CBN * kFirstParts * 64; i += 256) W_l[i] = img[i];
    un = dg.w_image[(size_t)nkc * CBN * kFirstParts * 64 * 4];
  } else {
    un = first_kernel_split<COUT>(w, kvol, nkc, W_l, red, tid);
  }
  // ---- combine the windows into this thread's mask word
  {
    const uint32_t wmask = (1u << ksize) - 1u;
    uint32_t m = 0u;
#pragma unroll
    for (int j = 0; j < kMaxWin; ++j) {
      if (j >= n_win) continue;
      uint32_t bits = w0[j] >> sh;
      if (sh + ksize > 32) bits |= w1[j] << (32 - sh);
      bits &= wmask;
      const int rel = (yz_lo + j) * ksize - k_lo;        // where the window's offset 0 sits in this word (may be < 0)
      m |= rel >= 0 ? bits << rel : bits >> (-rel);
    }
    if (k_hi - k_lo < 31) m &= (1u << (k_hi - k_lo + 1)) - 1u;            // offsets >= kvol do not exist
    M_l[v * 4 + wd] = v0 + v < n ? m : 0u;
  }
  __syncthreads();

  const int r16 = lane & 15, q4 = lane >> 4;
  f32x4 acc[CBN];
#pragma unroll
  for (int cb = 0; cb < CBN; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int kc = 0; kc < nkc; ++kc) {
    // A fragment of lane (r16, q4): offsets 32 kc + {4 q4 .. 4 q4 + 3, 16 + 4 q4 .. 16 + 4 q4 + 3} of row 16 wave + r16
    const uint32_t word = M_l[(wave * 16 + r16) * 4 + kc];
    const uint32_t b8 = ((word >> (4 * q4)) & 0xFu) | (((word >> (16 + 4 * q4)) & 0xFu) << 4);
    uint32_t aw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      aw[j] = ((b8 >> (2 * j)) & 1u ? 0x3C00u : 0u) | ((b8 >> (2 * j + 1)) & 1u ? 0x3C000000u : 0u);   // f16 1.0 = 0x3C00
    const f16x8_b a = __builtin_bit_cast(f16x8_b, make_uint4(aw[0], aw[1], aw[2], aw[3]));
#pragma unroll
    for (int cb = 0; cb < CBN; ++cb) {
      const f16x8_b b0 = __builtin_bit_cast(f16x8_b, W_l[((kc * CBN + cb) * kFirstParts + 0) * 64 + lane]);
      const f16x8_b b1 = __builtin_bit_cast(f16x8_b, W_l[((kc * CBN + cb) * kFirstParts + 1) * 64 + lane]);
      const f16x8_b b2 = __builtin_bit_cast(f16x8_b, W_l[((kc * CBN + cb) * kFirstParts + 2) * 64 + lane]);
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b2, acc[cb], 0, 0, 0);      // smallest parts first
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, acc[cb], 0, 0, 0);
    }
  }
#pragma unroll
  for (int cb = 0; cb < CBN; ++cb) {
    const int col = cb * 16 + r16;
    const float sc = scale ? scale[col] : 1.f, shf = shift ? shift[col] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long orow = v0 + wave * 16 + q4 * 4 + e;
      if (orow < n) {
        float y = (acc[cb][e] * un) * sc + shf;
        if (relu) y = fmaxf(y, 0.f);
        if (dg.err && out_of_f16_range(y)) atomicOr(dg.err, 32);
        if (out_split) store_split(out, orow, COUT, col, y);   // operand image for block1 (ConvParams::a_split)
        else out[orow * COUT + col] = y;
      }
    }
  }
}

template <int COUT, int KS>
__global__ void __launch_bounds__(256)
k_conv_first_bits(const int32_t *__restrict__ coords, long long n, const uint32_t *__restrict__ grid,
                  GridDesc g, int ksize_rt, int kvol, const float *__restrict__ w,
                  const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                  float *__restrict__ out, const DynGrid dg, int out_split) {
  conv_first_bits_body<COUT, KS>(coords, n, grid, g, ksize_rt, kvol, w, scale, shift, relu, out, dg, out_split, blockIdx.x);
}

// conv1 AND the level-0 3x3x3 neighbour map in one launch (imf_fragment_forward): both need only the level-0 rows (and
// grid / table), block1 needs both, and as two launches one of them has to cross streams -- the hand-over (event record,
// stream wait) costs ~15 us on the critical path.  Even workgroups run conv1's 64-row blocks, odd ones the map's tiles.
struct MapArgs {
  const imf_slot *tab;
  uint32_t capmask;
  const int32_t *n_out_dev;
  int32_t *tile_rows, *nbr;
  uint32_t *tile_mask;
  long long n_slots;
};
template <int COUT, int KS>
__global__ void __launch_bounds__(256)
k_conv_first_and_map(const int32_t *__restrict__ coords, long long n, const uint32_t *__restrict__ grid,
                     GridDesc g, int kvol, const float *__restrict__ w, const float *__restrict__ scale,
                     const float *__restrict__ shift, int relu, float *__restrict__ out, const DynGrid dg, int out_split,
                     const MapArgs m) {
  const long long idx = blockIdx.x >> 1;
  if (blockIdx.x & 1) {
    if (idx < m.n_slots / IMF_TILE_ROWS)
      rulebook_tile<+1, false>(m.tab, m.capmask, coords, n, m.n_out_dev, 1, 3, 27, m.tile_rows, m.nbr, m.tile_mask, m.n_slots, idx);
  } else {
    conv_first_bits_body<COUT, KS>(coords, n, grid, g, KS, kvol, w, scale, shift, relu, out, dg, out_split, idx);
  }
}

}  // namespace imf

using namespace imf;

extern "C" {

int64_t imf_packed_weight_floats(int kvol, int cin, int cout) { return (int64_t)kvol * cin * cout; }
int64_t imf_packed_weight_floats_split16(int kvol, int cin, int cout) { return (int64_t)kvol * cin * cout + 64; }

int imf_pack_weights(const float *w, int kvol, int cin, int cout, float *packed, void *stream) {
  IMF_REQUIRE(w && packed, "imf_pack_weights: null pointer");
  IMF_REQUIRE(kvol >= 1 && kvol <= IMF_MAX_KVOL, "imf_pack_weights: kvol=%d", kvol);
  IMF_REQUIRE(cin > 0 && cin % 32 == 0 && cout > 0 && cout % 32 == 0,
              "imf_pack_weights: cin=%d cout=%d must be multiples of 32", cin, cout);
  const long long total = (long long)kvol * cin * cout;
  k_pack_weights<<<(unsigned)div_up(total, 256), 256, 0, (hipStream_t)stream>>>(w, kvol, cin, cout,
                                                                                packed);
  IMF_CHECK_LAUNCH("k_pack_weights");
  return IMF_OK;
}

/* Resident workgroups per CU the runtime reports for a sparse-conv kernel instantiation (tuning aid). */
int imf_spconv_occupancy(int variant, int co_blk, int j) {
  int n = -1;
  const void *f = nullptr;
  int threads = 256;
#define IMF_PICK(K, T)                                                                    \
  do {                                                                                    \
    threads = T;                                                                          \
    if (co_blk == 4 && j == 4) f = (const void *)K<4, 4>;                                 \
    else if (co_blk == 4 && j == 2) f = (const void *)K<4, 2>;                            \
    else if (co_blk == 2 && j == 4) f = (const void *)K<2, 4>;                            \
    else f = (const void *)K<2, 2>;                                                       \
  } while (0)
  if (variant == 0) IMF_PICK(k_spconv_mfma, 256);
  else if (variant == 1) IMF_PICK(k_spconv_mfma_simple, 256);
  else return -1;
#undef IMF_PICK
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, threads, 0) != hipSuccess) return -1;
  return n;
}

static int split_min_blocks() {
  const int v = 400;   // measured: 438 unsplit workgroups (a pair's stride-2 level) beat split 2 + reduce by 1.5 % per step
  return v;
}
static int split_target() {
  const int v = 768;   // (round-1 values; 512 ... 1536 measured in round 2)
  return v;
}

int imf_spconv_auto_split(int64_t n_slots, int cout, int kvol) {
  return auto_split_rule(n_slots, cout, kvol, split_min_blocks(), split_target());
}

/* Largest split the rule can return for any row count up to the capacity (fewer rows -> more partitions). */
int imf_spconv_max_split(int cout, int kvol) {
  return auto_split_rule(IMF_TILE_ROWS, cout, kvol, split_min_blocks(), split_target());
}

size_t imf_spconv_workspace_bytes(int64_t n_slots, int cout, int split) {
  if (split > 1) return (size_t)split * (size_t)n_slots * (size_t)cout * sizeof(float);
  // unsplit launches of >= 512 tiles may balance their last partial round of workgroups (variant 6)
  const int64_t n_tiles = n_slots / IMF_TILE_ROWS;
  if (n_tiles < 512 || n_tiles % 256 == 0) return 0;
  return (size_t)8 * (size_t)(n_tiles % 256) * IMF_TILE_ROWS * (size_t)cout * sizeof(float);
}

int imf_spconv_fwd(const imf_conv_args *a, void *stream) {
  IMF_REQUIRE(a, "imf_spconv_fwd: null args");
  IMF_REQUIRE(a->in_a && a->w_packed && a->out, "imf_spconv_fwd: null pointer");
  IMF_REQUIRE(a->kvol >= 1 && a->kvol <= IMF_MAX_KVOL, "imf_spconv_fwd: kvol=%d", a->kvol);
  IMF_REQUIRE((a->nbr && a->tile_mask) || a->kvol == 1,
              "imf_spconv_fwd: nbr / tile_mask may be NULL only when kvol == 1");
  IMF_REQUIRE(a->c_a > 0 && a->c_a % 32 == 0 && a->c_b >= 0 && a->c_b % 32 == 0,
              "imf_spconv_fwd: c_a=%d c_b=%d must be multiples of 32", a->c_a, a->c_b);
  IMF_REQUIRE((a->c_b == 0) == (a->in_b == nullptr), "imf_spconv_fwd: in_b / c_b mismatch");
  IMF_REQUIRE(a->cout > 0 && a->cout % 32 == 0, "imf_spconv_fwd: cout=%d", a->cout);
  IMF_REQUIRE(a->n_slots > 0 && a->n_slots % IMF_TILE_ROWS == 0, "imf_spconv_fwd: n_slots");
  IMF_REQUIRE(a->n_out > 0 && (a->tile_rows || a->n_out <= a->n_slots), "imf_spconv_fwd: n_out");
  const int cin = a->c_a + a->c_b;
  const int J = ci_chunk_of(cin) / 16, CB = co_blk_of(a->cout);
  IMF_REQUIRE(!a->l2norm || a->cout == 16 * CB, "imf_spconv_fwd: l2norm needs cout in {32, 64}");
  IMF_REQUIRE(a->variant == 0 || a->variant == 1 || a->variant == 3 || a->variant == 6,
              "imf_spconv_fwd: variant=%d (0 = fp32 MFMA, 1 = fp32 MFMA without the pipeline, 3 = bf16x3 MFMA, 6 = split-f16 MFMA)", a->variant);
  const bool v16 = a->variant == 6 || a->variant == 3;   // the 16-bit matrix pipe: LDS-DMA kernels only
  const bool simple = !v16 && (a->variant == 1 || a->kvol >= kKCache || (a->c_b > 0 && a->c_a % (16 * J) != 0));
  IMF_REQUIRE(!v16 || a->kvol < kKCache,
              "imf_spconv_fwd: variants 6 / 3 (split-f16 / bf16x3 weights) need kvol <= %d", kKCache - 1);
  IMF_REQUIRE(!v16 || (a->kvol * (cin / 32) < kSubTab && a->c_a <= 1024 && a->c_b <= 1024),
              "imf_spconv_fwd: variants 6 / 3 need kvol * cin / 32 < %d and <= 1024 channels per source (kvol=%d cin=%d): use variant 0",
              kSubTab, a->kvol, cin);
  IMF_REQUIRE(a->variant != 3 || (!(a->kernel_tag & 2) && !a->tickets), "imf_spconv_fwd: variant 3 has no register-staged kernel / tickets");
  // Variant 0 (fp32 MFMA) runs on the LDS-DMA kernels too since round 5 (k_spconv_g / k_spconv_w with AR = kArF32: the
  // fp32 weight image has the split-f16 image's sub-stage addressing) wherever their tables cover the shape; kernel_tag
  // bit 1 or `tickets` keep the register-staged round-1 kernel k_spconv_mfma (A/B, the in-launch split-K combine).
  const bool dma0 = a->variant == 0 && !simple && !(a->kernel_tag & 2) && !a->tickets && a->kvol < kKCache &&
                    a->kvol * (cin / 32) < kSubTab && a->c_a <= 1024 && a->c_b <= 1024;
  const bool dma = v16 || dma0;
  // kernel_tag bits 2 / 3 (variant 6, variant 0 on the DMA kernels): the wave-split kernel (spconv_w.hip) with 8 / 4
  // wavefronts per workgroup -- the whole tile in one workgroup, no split-K partitions, no reduce launch
  const int wsplit = dma ? ((a->kernel_tag & 4) ? 8 : ((a->kernel_tag & 8) ? 4 : 0)) : 0;
  if (wsplit) {
    IMF_REQUIRE(a->cout % 64 == 0 && (a->kvol > 1 || cin >= 256),
                "imf_spconv_fwd: the wave-split kernel needs cout %% 64 == 0 and kvol > 1 or cin >= 256 (kvol=%d cin=%d cout=%d)",
                a->kvol, cin, a->cout);
    IMF_REQUIRE(a->split_k <= 1 && !a->tickets, "imf_spconv_fwd: the wave-split kernel takes no split_k / tickets");
  }
  int split = (simple || wsplit) ? 1 : (a->split_k > 0 ? a->split_k : imf_spconv_auto_split(a->n_slots, a->cout, a->kvol));
  IMF_REQUIRE(split >= 1 && split <= 32, "imf_spconv_fwd: split_k=%d", split);
  if (split > 1)
    IMF_REQUIRE(a->workspace && a->workspace_bytes >= imf_spconv_workspace_bytes(a->n_slots, a->cout, split),
                "imf_spconv_fwd: split_k=%d needs %zu workspace bytes", split,
                imf_spconv_workspace_bytes(a->n_slots, a->cout, split));
  ConvParams p{a->in_a, a->in_b, a->c_a, a->c_b, a->w_packed, a->kvol, a->cout, a->tile_rows,
               a->nbr, a->tile_mask, (long long)a->n_slots, (long long)a->n_out, a->scale, a->shift,
               a->residual, a->relu, a->l2norm, a->out, (float *)a->workspace,
               ((a->variant == 0 && !simple) || a->variant == 6) ? a->tickets : nullptr, 0};
  p.tail_begin = p.tail_split = 0;
  p.w_unscale = a->variant == 6 ? a->w_packed + (long long)a->kvol * cin * a->cout + 1 : nullptr;
  p.arith = a->variant == 6 ? kArF16x2 : (a->variant == 3 ? kArBf16x3 : kArF32);
  p.n_out_dev = a->n_out_dev;
  p.dyn_split_kvol = (a->n_out_dev && !wsplit) ? a->dyn_split_kvol : 0;
  p.slots_extra = a->slots_extra;
  p.split_min_blocks = split_min_blocks();
  p.split_target = split_target();
  IMF_REQUIRE(!a->n_out_dev || dma || split == 1,
              "imf_spconv_fwd: n_out_dev (capacity mode) on the fp32-MFMA kernels needs an unsplit launch (split_k = 1)");
  p.err = a->dyn_err;
  p.geglu = a->geglu;
  p.a_split = (a->operand_format & IMF_FMT_A_SPLIT) ? 1 : 0;
  p.res_split = (a->operand_format & IMF_FMT_RES_SPLIT) ? 1 : 0;
  p.out_split = (a->operand_format & IMF_FMT_OUT_SPLIT) ? 1 : 0;
  IMF_REQUIRE(!a->operand_format || (a->variant == 6 && split == 1 && !a->tickets && !(a->kernel_tag & 2)),
              "imf_spconv_fwd: operand_format needs variant 6 and an unsplit launch (split_k=%d)", split);
  IMF_REQUIRE(!p.out_split || !a->l2norm, "imf_spconv_fwd: IMF_FMT_OUT_SPLIT not with l2norm");
  IMF_REQUIRE(!p.res_split || a->residual, "imf_spconv_fwd: IMF_FMT_RES_SPLIT without a residual");
  IMF_REQUIRE(!a->geglu || ((v16 || (a->variant == 0 && !simple)) && !wsplit && a->kvol == 1 && a->cout % 64 == 0 &&
                            split == 1 && !a->scale && !a->residual && !a->relu && !a->l2norm && (dma0 || !(a->kernel_tag & 2))),
              "imf_spconv_fwd: geglu needs variant 6 (k_spconv_g) or 0, kvol 1, cout %% 64 == 0, an unsplit launch and no other epilogue");
  // XCD-contiguous tile order of k_spconv_g (IMF_G_XCD: bit 0 = the 64-column launches, bit 1 = the 32-column ones; default
  // both): workgroup b runs on XCD b % 8 and every XCD has its own 4 MiB L2.  In launch order each XCD gathers from ALL input
  // rows (26 MB for 64 channels at 103 k rows); when XCD x instead walks ONE range of consecutive tiles -- rows are in scan
  // order, a tile's neighbours sit in nearby tiles -- its L2 serves ~1/8 of the rows.  The ranges are cut from the ACTUAL
  // tiles on the device (capacity mode; cut from the capacity they left the last XCDs idle: round 2 measured that form
  // slower), the grid's x extent is padded to a multiple of 8 so that a workgroup's XCD is blockIdx.x & 7.  Same sums.
  // Measured (round 3, pair step, same box): 0.979 -> 0.955 ms together with the same order in k_spconv_w.
  const int xcd = 3;
  // (not for parity-grouped transposed maps: a range of consecutive tiles there is one parity class spread over the whole
  // level -- no locality to win, measured 43 -> 54 us for conv2_tr)
  const bool g_xcd = dma && !wsplit && ((CB == 4 && (xcd & 1)) || (CB == 2 && (xcd & 2))) &&
                     a->n_slots == imf_rulebook_slots(a->n_out);
  p.no_xcd_swizzle = !g_xcd;
  IMF_REQUIRE(!p.dyn_split_kvol || (!p.tickets && a->split_k >= 1), "imf_spconv_fwd: dyn_split_kvol needs an explicit split_k cover and no tickets");
#ifndef IMF_WITH_H3
  if (a->variant == 6 && ((a->kernel_tag & 2) || a->tickets)) {
    set_error("imf_spconv_fwd: the register-staged variant-6 kernel (kernel_tag bit 1, tickets) is compiled into diagnostic "
              "builds only (make -C imfnet_amd/csrc h3)");
    return IMF_EUNSUPPORTED;
  }
#endif
  dim3 grid((unsigned)(a->n_slots / IMF_TILE_ROWS), (unsigned)(a->cout / (16 * CB)), (unsigned)split);
  if (g_xcd) grid.x = (grid.x + 7u) / 8u * 8u;
  hipStream_t st = (hipStream_t)stream;
  if (a->ev_begin) IMF_CHECK_HIP(hipEventRecord((hipEvent_t)a->ev_begin, st));
  if (wsplit) {
    launch_spconv_w(p, grid.x, wsplit, st, (a->kernel_tag & 1) | ((a->kernel_tag & 64) ? 2 : 0) | ((a->kernel_tag & 128) ? 4 : 0) |
                                             ((a->kernel_tag & 256) ? 8 : 0));
  } else if (dma0 || a->variant == 3) {
    launch_spconv_g(p, grid, CB, st, a->kernel_tag & 1);
  } else if (a->variant == 6) {
    // Balanced tail: with >= 2 full rounds of workgroups per CU and a partial last round (801 tiles on
    // 256 CUs: 33 CUs get a 4th tile and set the kernel time), the tail tiles are split over their
    // offsets so every CU receives the same work.  Needs a little workspace; skipped without it.
    // Measured (S50k): the two 64->64 layers drop 61 -> 56 us, but the extra reduce launch takes the
    // step-level gain back (0.886 vs 0.884 ms), so it is opt-in: IMF_CONV_TAIL=1.
#ifdef IMF_WITH_H3
    static const int tail_env = getenv("IMF_CONV_TAIL") ? atoi(getenv("IMF_CONV_TAIL")) : 0;
#else
    const int tail_env = 0;                                    // the balanced tail lives in k_spconv_h3 (diagnostic builds)
#endif
    const long long n_tiles = grid.x;
    const int tail_tiles = (int)(n_tiles % 256);
    const int ts = a->kvol >= 16 ? 8 : 4;
    // (not for parity-grouped transposed rulebooks: their slot order comes from atomics, and splitting
    // only SOME tiles would make a row's rounding depend on where it landed -- bit-reproducibility)
    const bool fixed_order = a->n_slots == imf_rulebook_slots(a->n_out);
    if (tail_env && fixed_order && split == 1 && grid.y == 1 && a->kvol >= 8 && n_tiles >= 512 && tail_tiles > 0 &&
        a->workspace && a->workspace_bytes >= (size_t)ts * tail_tiles * IMF_TILE_ROWS * a->cout * sizeof(float)) {
      p.tail_begin = (int)(n_tiles - tail_tiles);
      p.tail_split = ts;
      grid.x = (unsigned)(p.tail_begin + tail_tiles * ts);
    }
#ifdef IMF_WITH_H3
    launch_spconv_h3(p, grid, CB, st, a->kernel_tag);          // diagnostic build: register-staged twin / stamps / tickets
#else
    launch_spconv_g(p, grid, CB, st, a->kernel_tag & 1);
#endif
  } else if (simple) {
    if (CB == 4 && J == 4)      k_spconv_mfma_simple<4, 4><<<grid, 256, 0, st>>>(p);
    else if (CB == 4 && J == 2) k_spconv_mfma_simple<4, 2><<<grid, 256, 0, st>>>(p);
    else if (CB == 2 && J == 4) k_spconv_mfma_simple<2, 4><<<grid, 256, 0, st>>>(p);
    else                        k_spconv_mfma_simple<2, 2><<<grid, 256, 0, st>>>(p);
  } else {
    if (CB == 4 && J == 4)      k_spconv_mfma<4, 4><<<grid, 256, 0, st>>>(p);
    else if (CB == 4 && J == 2) k_spconv_mfma<4, 2><<<grid, 256, 0, st>>>(p);
    else if (CB == 2 && J == 4) k_spconv_mfma<2, 4><<<grid, 256, 0, st>>>(p);
    else                        k_spconv_mfma<2, 2><<<grid, 256, 0, st>>>(p);
  }
  IMF_CHECK_LAUNCH("k_spconv_mfma");
  if (a->ev_end) IMF_CHECK_HIP(hipEventRecord((hipEvent_t)a->ev_end, st));
  if (p.tail_split > 1) {
    const long long slot0 = (long long)p.tail_begin * IMF_TILE_ROWS;
    const long long total = (a->n_slots - slot0) * (a->cout / 4);
    k_spconv_reduce<<<(unsigned)div_up(total, 256), 256, 0, st>>>(p, p.tail_split, slot0);
    IMF_CHECK_LAUNCH("k_spconv_reduce");
  }
  if (split > 1 && !p.tickets) {
    const long long total = (long long)a->n_slots * (a->cout / 4);
    k_spconv_reduce<<<(unsigned)div_up(total, 256), 256, 0, st>>>(p, split, 0);
    IMF_CHECK_LAUNCH("k_spconv_reduce");
  }
  return IMF_OK;
}

int imf_spconv_small_cin(const float *in, int cin, const float *w, int kvol, int cout,
                         const int32_t *nbr, int64_t n_slots, int64_t n_out, const float *scale,
                         const float *shift, int relu, float *out, void *stream) {
  IMF_REQUIRE(in && w && nbr && out, "imf_spconv_small_cin: null pointer");
  IMF_REQUIRE(cin >= 1 && cin <= 4, "imf_spconv_small_cin: cin=%d not in [1,4]", cin);
  IMF_REQUIRE(cout == 32 || cout == 64, "imf_spconv_small_cin: cout=%d not in {32,64}", cout);
  IMF_REQUIRE(kvol >= 1 && kvol <= IMF_MAX_KVOL, "imf_spconv_small_cin: kvol=%d", kvol);
  const size_t lds = (size_t)kvol * cin * cout * sizeof(float);
  IMF_REQUIRE(lds <= 64 * 1024, "imf_spconv_small_cin: kernel does not fit 64 KiB of LDS");
  IMF_REQUIRE(n_out > 0 && n_slots >= n_out, "imf_spconv_small_cin: n_out / n_slots");
  hipStream_t st = (hipStream_t)stream;
  const unsigned nb = (unsigned)div_up(n_out, 256);
  if (cout == 32)
    k_spconv_small_cin<32><<<nb, 256, lds, st>>>(in, cin, w, kvol, nbr, n_slots, n_out, scale, shift, relu, out);
  else
    k_spconv_small_cin<64><<<nb, 256, lds, st>>>(in, cin, w, kvol, nbr, n_slots, n_out, scale, shift, relu, out);
  IMF_CHECK_LAUNCH("k_spconv_small_cin");
  return IMF_OK;
}

int imf_conv_first_fused(const imf_slot *table, int64_t capacity,
                         const int32_t *coords, int64_t n, int ts, int ksize, const float *in, int cin,
                         const float *w, int cout, const float *scale, const float *shift, int relu,
                         float *out, void *stream) {
  IMF_REQUIRE(table && coords && w && out, "imf_conv_first_fused: null pointer");
  IMF_REQUIRE(ksize == 3 || ksize == 5, "imf_conv_first_fused: ksize must be 3 or 5");
  IMF_REQUIRE(cin >= 1 && cin <= 4, "imf_conv_first_fused: cin=%d not in [1,4]", cin);
  IMF_REQUIRE(cout == 32 || cout == 64, "imf_conv_first_fused: cout=%d not in {32,64}", cout);
  IMF_REQUIRE(n > 0 && ts >= 1, "imf_conv_first_fused: bad n / ts");
  IMF_REQUIRE((capacity & (capacity - 1)) == 0, "imf_conv_first_fused: capacity not a power of 2");
  const int kvol = ksize * ksize * ksize;
  const size_t lds = (size_t)kvol * cin * cout * sizeof(float) + (size_t)kFirstRows * 128 * sizeof(int);
  IMF_REQUIRE(lds <= 64 * 1024, "imf_conv_first_fused: kernel does not fit 64 KiB of LDS");
  hipStream_t st = (hipStream_t)stream;
  const long long nb = div_up(n, kFirstRows);
  if (cout == 32)
    k_conv_first_fused<32><<<(unsigned)nb, 256, lds, st>>>(table, (uint32_t)(capacity - 1), coords, n, ts,
                                                          ksize, kvol, in, cin, w, scale, shift, relu, out);
  else
    k_conv_first_fused<64><<<(unsigned)nb, 256, lds, st>>>(table, (uint32_t)(capacity - 1), coords, n, ts,
                                                          ksize, kvol, in, cin, w, scale, shift, relu, out);
  IMF_CHECK_LAUNCH("k_conv_first_fused");
  return IMF_OK;
}

size_t imf_bitgrid_words(const int32_t *bbox, int ksize) {
  GridDesc g;
  size_t words = 0;
  if (!bbox || (ksize != 3 && ksize != 5)) return 0;
  return grid_desc_from_bbox(bbox, ksize, g, words) ? words : 0;
}

static int conv_first_bitgrid_impl(const int32_t *coords, int64_t n, const int32_t *bbox, const DynGrid &dg, int ksize,
                                   uint32_t *grid, size_t grid_words, const float *w, int cout,
                                   const float *scale, const float *shift, int relu, float *out, void *stream,
                                   bool grid_is_clear = false, int out_split = 0) {
  IMF_REQUIRE(coords && grid && w && out, "imf_conv_first_bitgrid: null pointer");
  IMF_REQUIRE(ksize == 3 || ksize == 5, "imf_conv_first_bitgrid: ksize must be 3 or 5");
  IMF_REQUIRE(cout == 32 || cout == 64, "imf_conv_first_bitgrid: cout=%d not in {32,64}", cout);
  IMF_REQUIRE(n > 0, "imf_conv_first_bitgrid: n");
  GridDesc g;
  memset(&g, 0, sizeof(g));
  size_t words = grid_words;
  if (!dg.bbox_dev)
    IMF_REQUIRE(bbox && grid_desc_from_bbox(bbox, ksize, g, words) && words <= grid_words,
                "imf_conv_first_bitgrid: bounding box too large for the provided grid");
  hipStream_t st = (hipStream_t)stream;
  if (!grid_is_clear) {   // grid_is_clear: imf_fragment_forward zeroed it and the level-0 compaction kernel set the bits
    IMF_CHECK_HIP(hipMemsetAsync(grid, 0, words * sizeof(uint32_t), st));
    k_bitgrid_fill<<<(unsigned)div_up(n, 256), 256, 0, st>>>(coords, n, grid, g, ksize, dg);
  }
  const int kvol = ksize * ksize * ksize;
  const size_t lds = (size_t)4 * (cout / 16) * kFirstParts * 64 * 16 + (size_t)kBitsRows * 4 * sizeof(uint32_t);   // B fragments + masks
  const unsigned nb = (unsigned)div_up(n, kBitsRows);
  if (cout == 32 && ksize == 5)      k_conv_first_bits<32, 5><<<nb, 256, lds, st>>>(coords, n, grid, g, ksize, kvol, w, scale, shift, relu, out, dg, out_split);
  else if (cout == 32)               k_conv_first_bits<32, 3><<<nb, 256, lds, st>>>(coords, n, grid, g, ksize, kvol, w, scale, shift, relu, out, dg, out_split);
  else if (ksize == 5)               k_conv_first_bits<64, 5><<<nb, 256, lds, st>>>(coords, n, grid, g, ksize, kvol, w, scale, shift, relu, out, dg, out_split);
  else                               k_conv_first_bits<64, 3><<<nb, 256, lds, st>>>(coords, n, grid, g, ksize, kvol, w, scale, shift, relu, out, dg, out_split);
  IMF_CHECK_LAUNCH("k_conv_first_bits");
  return IMF_OK;
}

int imf_conv_first_bitgrid_flags(const int32_t *coords, int64_t n, const int32_t *bbox, int ksize,
                                 uint32_t *grid, size_t grid_words, const float *w, int cout, const float *scale,
                                 const float *shift, int relu, float *out, int32_t *flags, void *stream) {
  IMF_REQUIRE(bbox, "imf_conv_first_bitgrid: null pointer");
  DynGrid dg;
  memset(&dg, 0, sizeof(dg));
  dg.err = flags;
  return conv_first_bitgrid_impl(coords, n, bbox, dg, ksize, grid, grid_words, w, cout, scale, shift, relu, out, stream);
}

int imf_conv_first_bitgrid(const int32_t *coords, int64_t n, const int32_t *bbox, int ksize,
                           uint32_t *grid, size_t grid_words, const float *w, int cout,
                           const float *scale, const float *shift, int relu, float *out, void *stream) {
  IMF_REQUIRE(bbox, "imf_conv_first_bitgrid: null pointer");
  DynGrid dg;
  memset(&dg, 0, sizeof(dg));
  return conv_first_bitgrid_impl(coords, n, bbox, dg, ksize, grid, grid_words, w, cout, scale, shift, relu, out, stream);
}

int imf_conv_first_bitgrid_dyn(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, const int32_t *bbox_dev,
                               int32_t *err, int ksize, uint32_t *grid, size_t grid_words, const float *w, int cout,
                               const float *scale, const float *shift, int relu, float *out, void *stream) {
  IMF_REQUIRE(n_dev && bbox_dev && err && grid_words > 0, "imf_conv_first_bitgrid_dyn: null pointer");
  DynGrid dg{n_dev, bbox_dev, err, (unsigned long long)grid_words};
  return conv_first_bitgrid_impl(coords, n_cap, nullptr, dg, ksize, grid, grid_words, w, cout, scale, shift, relu, out,
                                 stream);
}

}  // extern "C"

namespace imf {
// the executor's entry points: as the public ones, with the output optionally written as a split-f16 operand image
int conv_first_bitgrid_flags_fmt(const int32_t *coords, int64_t n, const int32_t *bbox, int ksize, uint32_t *grid,
                                 size_t grid_words, const float *w, int cout, const float *scale, const float *shift,
                                 int relu, float *out, int32_t *flags, hipStream_t stream, int out_split) {
  IMF_REQUIRE(bbox, "imf_conv_first_bitgrid: null pointer");
  DynGrid dg;
  memset(&dg, 0, sizeof(dg));
  dg.err = flags;
  return conv_first_bitgrid_impl(coords, n, bbox, dg, ksize, grid, grid_words, w, cout, scale, shift, relu, out, stream,
                                 false, out_split);
}
int conv_first_bitgrid_dyn_fmt(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, const int32_t *bbox_dev,
                               int32_t *err, int ksize, uint32_t *grid, size_t grid_words, const float *w, int cout,
                               const float *scale, const float *shift, int relu, float *out, hipStream_t stream,
                               int out_split) {
  IMF_REQUIRE(n_dev && bbox_dev && err && grid_words > 0, "imf_conv_first_bitgrid_dyn: null pointer");
  DynGrid dg{n_dev, bbox_dev, err, (unsigned long long)grid_words};
  return conv_first_bitgrid_impl(coords, n_cap, nullptr, dg, ksize, grid, grid_words, w, cout, scale, shift, relu, out,
                                 stream, false, out_split);
}
// conv1 on a grid the caller has zeroed and filled (as conv_first_bitgrid_dyn_cleared) TOGETHER with the level-0 3x3x3
// neighbour map (as imf_rulebook_conv_dyn, tensor stride 1) in one launch: k_conv_first_and_map
int conv_first_and_map_dyn(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, const int32_t *bbox_dev, int32_t *err,
                           int ksize, uint32_t *grid, size_t grid_words, const float *w, int cout, const float *scale,
                           const float *shift, int relu, float *out, int out_split, const imf_slot *table, int64_t capacity,
                           int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask, hipStream_t st, const float *w_image) {
  IMF_REQUIRE(coords && n_dev && bbox_dev && err && grid && grid_words > 0 && w && out && table && tile_rows && nbr && tile_mask,
              "conv_first_and_map_dyn: null pointer");
  IMF_REQUIRE((ksize == 3 || ksize == 5) && (cout == 32 || cout == 64) && n_cap > 0, "conv_first_and_map_dyn: ksize / cout / n");
  IMF_REQUIRE((capacity & (capacity - 1)) == 0, "conv_first_and_map_dyn: capacity not a power of 2");
  DynGrid dg{n_dev, bbox_dev, err, (unsigned long long)grid_words, w_image};
  GridDesc g;
  memset(&g, 0, sizeof(g));
  const int kvol = ksize * ksize * ksize;
  const size_t lds = (size_t)4 * (cout / 16) * kFirstParts * 64 * 16 + (size_t)kBitsRows * 4 * sizeof(uint32_t);
  const int64_t n_slots = imf_rulebook_slots(n_cap);
  MapArgs m{table, (uint32_t)(capacity - 1), n_dev, tile_rows, nbr, tile_mask, (long long)n_slots};
  const unsigned nb = 2u * (unsigned)(n_slots / IMF_TILE_ROWS);      // conv1's 64-row blocks == the map's tiles
  if (cout == 32 && ksize == 5)      k_conv_first_and_map<32, 5><<<nb, 256, lds, st>>>(coords, n_cap, grid, g, kvol, w, scale, shift, relu, out, dg, out_split, m);
  else if (cout == 32)               k_conv_first_and_map<32, 3><<<nb, 256, lds, st>>>(coords, n_cap, grid, g, kvol, w, scale, shift, relu, out, dg, out_split, m);
  else if (ksize == 5)               k_conv_first_and_map<64, 5><<<nb, 256, lds, st>>>(coords, n_cap, grid, g, kvol, w, scale, shift, relu, out, dg, out_split, m);
  else                               k_conv_first_and_map<64, 3><<<nb, 256, lds, st>>>(coords, n_cap, grid, g, kvol, w, scale, shift, relu, out, dg, out_split, m);
  IMF_CHECK_LAUNCH("k_conv_first_and_map");
  return IMF_OK;
}
// imf_conv_first_bitgrid_dyn for a grid the caller has already zeroed AND filled (imf_fragment_forward clears it before
// the level-0 pyramid, whose compaction kernel sets the bits: two launches fewer between the pyramid and conv1)
int conv_first_bitgrid_dyn_cleared(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, const int32_t *bbox_dev,
                                   int32_t *err, int ksize, uint32_t *grid, size_t grid_words, const float *w, int cout,
                                   const float *scale, const float *shift, int relu, float *out, hipStream_t stream,
                                   int out_split, const float *w_image) {
  IMF_REQUIRE(n_dev && bbox_dev && err && grid_words > 0, "imf_conv_first_bitgrid_dyn: null pointer");
  DynGrid dg{n_dev, bbox_dev, err, (unsigned long long)grid_words, w_image};
  return conv_first_bitgrid_impl(coords, n_cap, nullptr, dg, ksize, grid, grid_words, w, cout, scale, shift, relu, out,
                                 stream, true, out_split);
}
}  // namespace imf

extern "C" {
/* conv1's hi / lo f16 weight image (see first_kernel_split): [ceil(kvol / 32)][cout / 16][2][64][8 halves] + the unscale factor. */
int64_t imf_first_kernel_image_floats(int kvol, int cout) { return (int64_t)((kvol + 31) / 32) * (cout / 16) * imf::kFirstParts * 64 * 4 + 4; }

int imf_pack_first_kernel(const float *w, int kvol, int cout, float *image, void *stream) {
  IMF_REQUIRE(w && image, "imf_pack_first_kernel: null pointer");
  IMF_REQUIRE((kvol == 27 || kvol == 125) && (cout == 32 || cout == 64), "imf_pack_first_kernel: kvol=%d cout=%d", kvol, cout);
  IMF_REQUIRE(((uintptr_t)image & 15) == 0, "imf_pack_first_kernel: image must be 16-byte aligned");
  if (cout == 32) imf::k_pack_first_kernel<32><<<1, 256, 0, (hipStream_t)stream>>>(w, kvol, image);
  else            imf::k_pack_first_kernel<64><<<1, 256, 0, (hipStream_t)stream>>>(w, kvol, image);
  IMF_CHECK_LAUNCH("k_pack_first_kernel");
  return IMF_OK;
}
}
