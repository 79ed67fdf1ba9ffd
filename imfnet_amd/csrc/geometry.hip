// Geometry kernels: voxel hash build, first-occurrence unique, pyramid levels, rulebooks.
// Integer / byte work, HBM- and L2-latency bound: coalesced streams over points and slots,
// random probes into an L2-resident open-addressing table.  No MFMA here by design.
#include <string.h>

#include "common.h"
#include "geometry_internal.h"
#include "rulebook_tile.h"

namespace imf {


constexpr int kScanThreads = 256;
constexpr int kScanItems = 4;
constexpr int kScanTile = kScanThreads * kScanItems;  // 1024 rows per block

__device__ __forceinline__ int floor_div(int a, int s) {
  return a >= 0 ? a / s : -((-a + s - 1) / s);
}

// ---- K1: quantise + insert (atomicCAS on the key, atomicMin on the row index) -----------------
// Point ranges of the items of a batched build: item b = points [start[b], start[b+1]).
struct BatchStarts {
  long long start[IMF_MAX_BATCH];
  int nb;
};

// first row of every batch item at up to four levels in one launch (rows are grouped by batch index, ascending); level j
// owns the blocks [blk0[j], blk0[j+1])
struct ItemJobs {
  const int32_t *coords[4], *n_dev[4];
  int32_t *starts[4];
  int blk0[5];
};
__global__ void __launch_bounds__(256) k_item_starts(const ItemJobs js) {
  int l = 0;
  while (l < 3 && (int)blockIdx.x >= js.blk0[l + 1]) ++l;
  const int n = *js.n_dev[l];
  const int i = ((int)blockIdx.x - js.blk0[l]) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t *coords = js.coords[l];
  const int b = coords[4 * i];
  if ((i == 0 || coords[4 * (i - 1)] != b) && b >= 0 && b < IMF_MAX_BATCH) js.starts[l][b] = i;
}

// Bounding box (b,x,y,z min / max) of a workgroup's voxels -> wg_bbox[blockIdx.x][0..7] (plain stores, no atomics:
// 2 385 workgroups improving 8 shared words with atomics made k_insert_points 32 -> 108 us; same-address atomics
// serialise in L2).  Block 0 of k_flag_first folds the per-workgroup boxes into the level's box.  lo / hi of lanes
// without a voxel must be +INT_MAX / INT_MIN.
__device__ __forceinline__ void block_bbox_store(int4 lo, int4 hi, int32_t *wg_bbox) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int o = 32; o > 0; o >>= 1) {
    lo.x = min(lo.x, __shfl_down(lo.x, o, 64)); lo.y = min(lo.y, __shfl_down(lo.y, o, 64));
    lo.z = min(lo.z, __shfl_down(lo.z, o, 64)); lo.w = min(lo.w, __shfl_down(lo.w, o, 64));
    hi.x = max(hi.x, __shfl_down(hi.x, o, 64)); hi.y = max(hi.y, __shfl_down(hi.y, o, 64));
    hi.z = max(hi.z, __shfl_down(hi.z, o, 64)); hi.w = max(hi.w, __shfl_down(hi.w, o, 64));
  }
  __shared__ int bb[4][8];
  if (lane == 0) {
    bb[w][0] = lo.x; bb[w][1] = lo.y; bb[w][2] = lo.z; bb[w][3] = lo.w;
    bb[w][4] = hi.x; bb[w][5] = hi.y; bb[w][6] = hi.z; bb[w][7] = hi.w;
  }
  __syncthreads();
  if (t < 8) {
    const bool is_min = t < 4;
    int v = bb[0][t];
    for (int q = 1; q < 4; ++q) v = is_min ? min(v, bb[q][t]) : max(v, bb[q][t]);
    wg_bbox[(long long)blockIdx.x * 8 + t] = v;
  }
}

// dyn (optional, device): [0] = number of points, [1] = number of items, [2 + b] = first point of item b --
// the per-fragment scalars of a captured launch sequence (IMF_DYN_WORDS ints).
//
// Voxel insert of the raw points (util/misc.py:82-87) with the keys of a WORKGROUP's 1 024 points deduplicated in LDS first
// (round 4).  A 2.5 cm voxel holds ~5 points, but only runs of neighbours in point order share a wavefront: electing one
// leader per run (rounds 2-3) still sent ~151 k of a 258 k-point fragment's keys to the table for 51 k voxels -- and every
// device-scope atomic is a 64-byte line written back beyond the XCD's L2 (PMC, round 3: 62 MB per launch for 16 MB of
// algorithmic bytes).  Here every point goes into a 2 048-slot LDS table (ds_cmpst_b64 on the key, ds_min on the point
// index), the table's distinct keys (83 k per fragment) are inserted into the level's table once each with the workgroup's
// smallest point index, and the points pick their global slot up from LDS.  The level's table ends up with the same keys
// and the same minimum point index per key as one insert per point (which SLOT a key lands in may differ; nothing reads
// that).  Points per workgroup, measured on the pair (imf_voxelize / step): 512: 49.4 us / 0.931 ms, 1 024: 49.6 / 0.931,
// 2 048: 56.8 / 0.943, 4 096: 78.2 / 0.953 (fewer, longer workgroups: the global inserts of a workgroup are a dependent
// chain per thread); the run-leader kernel: 62.3 / 0.943.
#ifndef IMF_INS_PPT
#define IMF_INS_PPT 4
#endif
constexpr int kInsThreads = 256, kInsPerThread = IMF_INS_PPT, kInsPoints = kInsThreads * kInsPerThread, kInsSlots = 2 * kInsPoints;
static_assert(kInsPerThread % 2 == 0 && kInsSlots <= 65536, "k_insert_points_wg packs two 16-bit LDS slot numbers per word (mine[kInsPerThread / 2])");

template <typename T>
__global__ void __launch_bounds__(kInsThreads)
k_insert_points_wg(const T *__restrict__ xyz, int64_t n, double voxel, int batch0, const BatchStarts bs,
                   const int32_t *__restrict__ dyn, imf_slot *tab, uint32_t capmask, int32_t *slot_of, int32_t *err,
                   int32_t *wg_bbox) {
  __shared__ unsigned long long lkey[kInsSlots];
  __shared__ int32_t lval[kInsSlots];                 // smallest point index of the key, later its global slot
  const int t = threadIdx.x;
  for (int s = t; s < kInsSlots; s += kInsThreads) {
    lkey[s] = kEmptyKey;
    lval[s] = 0x7FFFFFFF;
  }
  int nb = bs.nb;
  if (dyn) {
    n = min((int64_t)dyn[0], n);
    nb = min(max(dyn[1], 1), IMF_MAX_BATCH);
  }
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kInsPoints;
  const int big = 0x7FFFFFFF;
  int4 lo = make_int4(big, big, big, big), hi = make_int4(-big - 1, -big - 1, -big - 1, -big - 1);
  uint32_t mine[kInsPerThread / 2];                   // LDS slot of each of this thread's points, 16 bits each
  bool bad = false;
#pragma unroll
  for (int j = 0; j < kInsPerThread; ++j) {
    const int64_t i = base + (int64_t)j * kInsThreads + t;
    uint32_t l = 0;
    if (i < n) {
      int batch = batch0;
      if (dyn) {
        for (int b = 1; b < nb; ++b) batch += (i >= dyn[2 + b]) ? 1 : 0;
      } else {
        for (int b = 1; b < nb; ++b) batch += (i >= bs.start[b]) ? 1 : 0;   // items are contiguous point ranges
      }
      // util/misc.py:82 -- np.floor(xyz / voxel_size) in float64 (IEEE division, exact floor)
      double fx = floor((double)xyz[3 * i + 0] / voxel);
      double fy = floor((double)xyz[3 * i + 1] / voxel);
      double fz = floor((double)xyz[3 * i + 2] / voxel);
      const bool ok = fx >= -kCoordLim && fx < kCoordLim && fy >= -kCoordLim && fy < kCoordLim &&
                      fz >= -kCoordLim && fz < kCoordLim;   // also false for NaN
      if (!ok) {
        bad = true;
        fx = fy = fz = 0.0;
      }
      const int x = (int)fx, y = (int)fy, z = (int)fz;
      lo.x = min(lo.x, batch); lo.y = min(lo.y, x); lo.z = min(lo.z, y); lo.w = min(lo.w, z);
      hi.x = max(hi.x, batch); hi.y = max(hi.y, x); hi.z = max(hi.z, y); hi.w = max(hi.w, z);
      const unsigned long long key = pack_key(batch, x, y, z);
      l = hash64(key) & (kInsSlots - 1);
      while (true) {                                  // <= 2 048 keys in 4 096 slots: always terminates
        const unsigned long long prev = atomicCAS(&lkey[l], (unsigned long long)kEmptyKey, key);
        if (prev == kEmptyKey || prev == key) break;
        l = (l + 1) & (kInsSlots - 1);
      }
      atomicMin(&lval[l], (int32_t)i);
    }
    if (j & 1) mine[j >> 1] |= l << 16;
    else mine[j >> 1] = l;
  }
  if (bad) atomicOr(err, 1);
  __syncthreads();
  for (int s = t; s < kInsSlots; s += kInsThreads) {
    const unsigned long long key = lkey[s];
    if (key == kEmptyKey) continue;
    const int32_t first = lval[s];
    const uint32_t g = hash_insert(tab, capmask, key, 0);
    // the slot's row index only ever decreases: a stale read is larger than the truth, i.e. errs towards doing the atomic
    if (__hip_atomic_load(&tab[g].val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > first) atomicMin(&tab[g].val, first);
    lval[s] = (int32_t)g;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kInsPerThread; ++j) {
    const int64_t i = base + (int64_t)j * kInsThreads + t;
    if (i < n) slot_of[i] = lval[(mine[j >> 1] >> ((j & 1) * 16)) & 0xFFFF];
  }
  // the level's bounding box (every voxel has a point here): known BEFORE the compaction, which can then fill conv1's
  // occupancy bit grid itself (its origin is the box's minimum)
  if (wg_bbox) block_bbox_store(lo, hi, wg_bbox);
}

__global__ void __launch_bounds__(256)
k_insert_coords(const int32_t *__restrict__ cin, const int32_t *__restrict__ n_dev, int stride,
                imf_slot *tab, uint32_t capmask, int32_t *slot_of) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *n_dev) return;
  int4 c = reinterpret_cast<const int4 *>(cin)[i];
  int x = floor_div(c.y, stride) * stride;
  int y = floor_div(c.z, stride) * stride;
  int z = floor_div(c.w, stride) * stride;
  uint32_t s = hash_insert(tab, capmask, pack_key(c.x, x, y, z), __builtin_ctz((unsigned)stride));
  if (__hip_atomic_load(&tab[s].val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (int32_t)i)   // (see k_insert_points)
    atomicMin(&tab[s].val, (int32_t)i);
  slot_of[i] = (int32_t)s;
}

// ---- K2: first-occurrence flag (sign of slot_of) + per-block count -----------------------------
__device__ __forceinline__ int block_sum_256(int v, int *lds4) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) lds4[w] = v;
  __syncthreads();
  return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

__global__ void __launch_bounds__(kScanThreads)
k_flag_first(int32_t *slot_of, const imf_slot *__restrict__ tab, int64_t n_static,
             const int32_t *__restrict__ n_dev, int32_t *block_sums,
             const int32_t *__restrict__ wg_bbox = nullptr, int n_wg = 0, int32_t *bbox_out = nullptr) {
  __shared__ int lds4[4];
  if (wg_bbox && blockIdx.x == 0) {   // level 0: fold k_insert_points' per-workgroup boxes into the level's box
    __shared__ int part[4][8];
    const int big = 0x7FFFFFFF;
    int4 lo = make_int4(big, big, big, big), hi = make_int4(-big - 1, -big - 1, -big - 1, -big - 1);
    for (int g = threadIdx.x; g < n_wg; g += kScanThreads) {       // two 16-byte loads per workgroup box, all in flight
      const int4 a = reinterpret_cast<const int4 *>(wg_bbox)[2 * g], c = reinterpret_cast<const int4 *>(wg_bbox)[2 * g + 1];
      lo.x = min(lo.x, a.x); lo.y = min(lo.y, a.y); lo.z = min(lo.z, a.z); lo.w = min(lo.w, a.w);
      hi.x = max(hi.x, c.x); hi.y = max(hi.y, c.y); hi.z = max(hi.z, c.z); hi.w = max(hi.w, c.w);
    }
    for (int o = 32; o > 0; o >>= 1) {
      lo.x = min(lo.x, __shfl_down(lo.x, o, 64)); lo.y = min(lo.y, __shfl_down(lo.y, o, 64));
      lo.z = min(lo.z, __shfl_down(lo.z, o, 64)); lo.w = min(lo.w, __shfl_down(lo.w, o, 64));
      hi.x = max(hi.x, __shfl_down(hi.x, o, 64)); hi.y = max(hi.y, __shfl_down(hi.y, o, 64));
      hi.z = max(hi.z, __shfl_down(hi.z, o, 64)); hi.w = max(hi.w, __shfl_down(hi.w, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      int *pw = part[threadIdx.x >> 6];
      pw[0] = lo.x; pw[1] = lo.y; pw[2] = lo.z; pw[3] = lo.w; pw[4] = hi.x; pw[5] = hi.y; pw[6] = hi.z; pw[7] = hi.w;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
      const bool is_min = threadIdx.x < 4;
      int r = part[0][threadIdx.x];
      for (int j = 1; j < 4; ++j) r = is_min ? min(r, part[j][threadIdx.x]) : max(r, part[j][threadIdx.x]);
      bbox_out[threadIdx.x] = r;
    }
    __syncthreads();
  }
  const int64_t n = n_dev ? min((int64_t)*n_dev, n_static) : n_static;
  int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  int cnt = 0;
#pragma unroll
  for (int e = 0; e < kScanItems; ++e) {
    int64_t i = base + e;
    if (i < n) {
      int s = slot_of[i];
      bool first = tab[s].val == (int32_t)i;
      slot_of[i] = first ? s : ~s;
      cnt += first;
    }
  }
  int tot = block_sum_256(cnt, lds4);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

struct GridFill {            // optional extra of the level-0 compaction: set conv1's occupancy bits (grid zeroed by the caller)
  uint32_t *grid;
  unsigned long long words_cap;
  const int32_t *bbox;       // device: min b,x,y,z, max b,x,y,z (complete: written by k_insert_points)
  int ksize;
  int32_t *err;
};

// ---- K3: order-preserving compaction: row r = rank of the first-occurrence point ---------------
// The exclusive scan of the per-block counts is done here, by every workgroup for itself (it sums the counts of the
// blocks before it: <= a few thousand ints from L2) -- a single-workgroup scan kernel between K2 and K3 cost a launch
// and a kernel boundary on the critical path of every level.  The last block writes the level's row count to *m_out.
// row_cap > 0: the level holds at most row_cap rows; a larger count is clamped and flagged (bit 1 of m_out[1]).
__global__ void __launch_bounds__(kScanThreads)
k_emit_unique(const int32_t *__restrict__ slot_of, imf_slot *tab,
              int64_t n_static, const int32_t *__restrict__ n_dev,
              const int32_t *__restrict__ block_sums, int32_t *m_out, int32_t *coords_out, int32_t *first_idx,
              const GridFill gf, int64_t row_cap = 0) {
  __shared__ int wsum[4], psum[4];
  const int64_t n = n_dev ? min((int64_t)*n_dev, n_static) : n_static;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  int block_off = 0;
  {
    int part = 0;
    for (int i = t; i < (int)blockIdx.x; i += kScanThreads) part += block_sums[i];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    if (lane == 0) psum[w] = part;
    __syncthreads();
    block_off = psum[0] + psum[1] + psum[2] + psum[3];
    if (blockIdx.x == gridDim.x - 1 && t == 0) {
      int m = block_off + block_sums[blockIdx.x];
      if (row_cap > 0 && m > row_cap) {
        m = (int)row_cap;
        atomicOr(m_out + 1, 2);
      }
      *m_out = m;
    }
  }
  int64_t base = (int64_t)blockIdx.x * kScanTile + t * kScanItems;
  int s[kScanItems];
  int cnt = 0;
#pragma unroll
  for (int e = 0; e < kScanItems; ++e) {
    int64_t i = base + e;
    s[e] = (i < n) ? slot_of[i] : -1;
    cnt += s[e] >= 0;
  }
  // wave inclusive scan
  int inc = cnt;
  for (int o = 1; o < 64; o <<= 1) {
    int v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int woff = 0;
  for (int q = 0; q < w; ++q) woff += wsum[q];
  int r = block_off + woff + inc - cnt;
  // conv1's occupancy bit grid (level 0 of imf_fragment_forward): one bit per voxel, origin = the box computed by
  // k_insert_points; a box larger than the grid raises IMF_FLAG_BITGRID and leaves the grid alone (conv1 does the same)
  GridDesc gd;
  bool fill = gf.grid != nullptr;
  if (fill) {
    int32_t bbv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) bbv[q] = gf.bbox[q];
    size_t words = 0;
    if (!grid_desc_from_bbox(bbv, gf.ksize, gd, words) || words > gf.words_cap) {
      if (blockIdx.x == 0 && t == 0) atomicOr(gf.err, 4);
      fill = false;
    }
  }
#pragma unroll
  for (int e = 0; e < kScanItems; ++e) {
    if (s[e] >= 0 && row_cap > 0 && r >= row_cap) {
      tab[s[e]].val = -1;         // over capacity (flagged by the last block): the voxel has no row
      ++r;
    } else if (s[e] >= 0) {
      uint64_t key = tab[s[e]].key;
      int4 c;
      c.x = (int)(key >> (3 * kCoordBits));
      c.y = ((int)((key >> (2 * kCoordBits)) & 0x3FFFF) << 14) >> 14;   // sign-extend 18 bits
      c.z = ((int)((key >> kCoordBits) & 0x3FFFF) << 14) >> 14;
      c.w = ((int)(key & 0x3FFFF) << 14) >> 14;
      reinterpret_cast<int4 *>(coords_out)[r] = c;
      if (first_idx) first_idx[r] = (int32_t)(base + e);
      tab[s[e]].val = r;   // the table now maps voxel -> row
      ++r;
      if (fill) {
        const int bit = c.y - gd.x0;
        atomicOr(gf.grid + grid_row(gd, c.x, c.z, c.w) + (bit >> 5), 1u << (bit & 31));
      }
    }
  }
}

static int run_unique_tail(int32_t *slot_of, int32_t *block_sums, imf_slot *tab,
                           int64_t n_max, const int32_t *n_dev, int32_t *coords_out,
                           int32_t *first_idx, int32_t *m_out, hipStream_t st, const GridFill *grid_fill = nullptr,
                           int64_t row_cap = 0, const int32_t *wg_bbox = nullptr, int n_wg = 0, int32_t *bbox_out = nullptr) {
  const int nb = (int)div_up(n_max, kScanTile);
  k_flag_first<<<nb, kScanThreads, 0, st>>>(slot_of, tab, n_max, n_dev, block_sums, wg_bbox, n_wg, bbox_out);
  GridFill gf;
  memset(&gf, 0, sizeof(gf));
  if (grid_fill) gf = *grid_fill;
  k_emit_unique<<<nb, kScanThreads, 0, st>>>(slot_of, tab, n_max, n_dev, block_sums, m_out,
                                             coords_out, first_idx, gf, row_cap);
  IMF_CHECK_LAUNCH("unique pipeline");
  return IMF_OK;
}

// ---- rulebooks (the per-tile body: rulebook_tile.h) -------------------------------------------------
template <int SIGN, bool INDIRECT>
__global__ void __launch_bounds__(256)
k_rulebook(const imf_slot *__restrict__ tab, uint32_t capmask,
           const int32_t *__restrict__ out_coords, int64_t n_out, const int32_t *__restrict__ n_out_dev, int ts,
           int ksize, int kvol, int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask, int64_t n_slots) {
  rulebook_tile<SIGN, INDIRECT>(tab, capmask, out_coords, n_out, n_out_dev, ts, ksize, kvol, tile_rows, nbr, tile_mask, n_slots,
                                blockIdx.x);
}

// Transposed conv: group fine rows by the parity of (coord / ts) per axis (8 classes).
__device__ __forceinline__ int parity_class(int4 c, int ts) {
  return (((c.y / ts) & 1)) | (((c.z / ts) & 1) << 1) | (((c.w / ts) & 1) << 2);
}

// Parity-class grouping of the transposed rulebook, DETERMINISTIC: a row's slot is
//   class base + (rows of its class in earlier blocks) + (its rank among its class inside the block),
// all three order-defined (no arrival-order atomics), so the slot table -- and with it the split-K
// partition a row falls into -- is the same on every run.  blockcnt: [n_blocks][8] scratch (the head of
// the not-yet-written neighbour table).
__global__ void __launch_bounds__(256)
k_class_count(const int32_t *__restrict__ coords, int64_t n, const int32_t *__restrict__ n_dev, int ts,
              int32_t *__restrict__ blockcnt) {
  __shared__ int cnt[8];
  if (n_dev) n = min((int64_t)*n_dev, n);
  if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
  __syncthreads();
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&cnt[parity_class(reinterpret_cast<const int4 *>(coords)[i], ts)], 1);   // a count: order-free
  __syncthreads();
  if (threadIdx.x < 8) blockcnt[blockIdx.x * 8 + threadIdx.x] = cnt[threadIdx.x];
}

// one workgroup: per class, exclusive scan of the block counts (in place) and the tile-aligned class bases
__global__ void __launch_bounds__(256)
k_class_bases(int32_t *__restrict__ blockcnt, int nb, int32_t *__restrict__ counters) {
  __shared__ int part[8][32], total[8];
  const int p = threadIdx.x >> 5, t = threadIdx.x & 31;       // 8 classes x 32 threads
  const int per = (nb + 31) / 32;
  const int lo = min(nb, t * per), hi = min(nb, lo + per);
  int s = 0;
  for (int b = lo; b < hi; ++b) s += blockcnt[b * 8 + p];
  part[p][t] = s;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int q = 0; q < 32; ++q) {
      const int v = part[p][q];
      part[p][q] = run;
      run += v;
    }
    total[p] = run;
  }
  __syncthreads();
  int run = part[p][t];
  for (int b = lo; b < hi; ++b) {
    const int v = blockcnt[b * 8 + p];
    blockcnt[b * 8 + p] = run;
    run += v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int base = 0;
    for (int q = 0; q < 8; ++q) {
      counters[q] = total[q];
      counters[8 + q] = base;
      base += (total[q] + IMF_TILE_ROWS - 1) / IMF_TILE_ROWS * IMF_TILE_ROWS;
    }
  }
}

__global__ void __launch_bounds__(256)
k_class_assign(const int32_t *__restrict__ coords, int64_t n, const int32_t *__restrict__ n_dev, int ts,
               const int32_t *__restrict__ counters, const int32_t *__restrict__ blockbase,
               int32_t *__restrict__ tile_rows) {
  __shared__ int wcnt[4][8];
  if (n_dev) n = min((int64_t)*n_dev, n);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int p = (i < n) ? parity_class(reinterpret_cast<const int4 *>(coords)[i], ts) : -1;
  int local = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {               // rank among the wave's rows of the same class, in lane order
    const unsigned long long bal = __ballot(p == q);
    if (p == q) local = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[w][q] = __popcll(bal);
  }
  __syncthreads();
  if (p >= 0) {
    int off = counters[8 + p] + blockbase[blockIdx.x * 8 + p] + local;
    for (int v = 0; v < w; ++v) off += wcnt[v][p];
    tile_rows[off] = (int32_t)i;
  }
}

__global__ void __launch_bounds__(256)
k_init_transpose(int32_t *counters, int32_t *tile_rows, int64_t n_slots, uint32_t *tile_mask,
                 int64_t n_mask) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 16) counters[i] = 0;
  if (i < n_slots) tile_rows[i] = -1;
  if (i < n_mask) tile_mask[i] = 0u;
}

__global__ void __launch_bounds__(256)
k_init_table(imf_slot *tab, int64_t capacity) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < capacity) reinterpret_cast<uint4 *>(tab)[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0x7FFFFFFFu, 0u);
}

// xyz_down = xyz[inds] on the device (util/misc.py:92 return_coords): one thread per output coordinate
template <typename T>
__global__ void __launch_bounds__(256)
k_gather_points(const T *__restrict__ xyz, const int32_t *__restrict__ first_idx, const int32_t *__restrict__ m_dev,
                int64_t m_cap, double *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t m = m_dev ? min((int64_t)*m_dev, m_cap) : m_cap;
  if (i >= 3 * m) return;
  const int64_t r = i / 3;
  out[i] = (double)xyz[3 * (int64_t)first_idx[r] + (i - 3 * r)];
}

static int init_table(imf_slot *tab, int64_t capacity, hipStream_t st) {
  k_init_table<<<(unsigned)div_up(capacity, 256), 256, 0, st>>>(tab, capacity);
  IMF_CHECK_LAUNCH("k_init_table");
  return IMF_OK;
}

}  // namespace imf

using namespace imf;

extern "C" {

int64_t imf_hash_capacity(int64_t n) {
  int64_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  return cap;
}

// slot_of[n] | block_sums[div_up(n, 1024) + 16] | per-workgroup bounding boxes of k_insert_points [div_up(n, 256)][8]
size_t imf_unique_workspace_bytes(int64_t n) {
  return (size_t)(n + div_up(n, kScanTile) + 16 + 4 + 8 * div_up(n, 256)) * sizeof(int32_t);   // + 4: the boxes start 16-byte aligned
}

int imf_voxelize(const void *xyz, int xyz_is_f64, int64_t n, double voxel_size, int batch_index,
                 int32_t *coords, int32_t *first_idx, int32_t *m_out, imf_slot *table,
                 int64_t capacity, void *workspace, int32_t *err_out, void *stream) {
  IMF_REQUIRE(xyz && coords && first_idx && m_out && table && workspace && err_out,
              "imf_voxelize: null pointer");
  IMF_REQUIRE(n > 0 && n < (1ll << 31) - 2048, "imf_voxelize: n=%lld out of range", (long long)n);
  IMF_REQUIRE(voxel_size > 0.0, "imf_voxelize: voxel_size must be > 0");
  IMF_REQUIRE(batch_index >= 0 && batch_index < 512, "imf_voxelize: batch_index out of [0,512)");
  IMF_REQUIRE(capacity >= 2 * n && (capacity & (capacity - 1)) == 0 && capacity <= (1ll << 32),
              "imf_voxelize: capacity must be a power of two >= 2n");
  hipStream_t st = (hipStream_t)stream;
  int32_t *slot_of = (int32_t *)workspace;
  int32_t *block_sums = slot_of + n;
  int rc = init_table(table, capacity, st);
  if (rc) return rc;
  const int nblk = (int)div_up(n, kInsPoints);
  BatchStarts one;
  memset(&one, 0, sizeof(one));
  one.nb = 1;
  if (xyz_is_f64)
    k_insert_points_wg<double><<<nblk, kInsThreads, 0, st>>>((const double *)xyz, n, voxel_size, batch_index, one, nullptr,
                                                             table, (uint32_t)(capacity - 1), slot_of, err_out, nullptr);
  else
    k_insert_points_wg<float><<<nblk, kInsThreads, 0, st>>>((const float *)xyz, n, voxel_size, batch_index, one, nullptr,
                                                            table, (uint32_t)(capacity - 1), slot_of, err_out, nullptr);
  IMF_CHECK_LAUNCH("k_insert_points_wg");
  return run_unique_tail(slot_of, block_sums, table, n, nullptr, coords, first_idx, m_out, st);
}

int imf_downsample(const int32_t *coords_in, const int32_t *n_in_dev, int64_t n_in_max,
                   int out_stride, int32_t *coords_out, int32_t *m_out, imf_slot *table,
                   int64_t capacity, void *workspace, void *stream) {
  IMF_REQUIRE(coords_in && n_in_dev && coords_out && m_out && table && workspace,
              "imf_downsample: null pointer");
  IMF_REQUIRE(n_in_max > 0 && n_in_max < (1ll << 31) - 2048, "imf_downsample: n out of range");
  IMF_REQUIRE(out_stride >= 1, "imf_downsample: out_stride must be >= 1");
  IMF_REQUIRE(capacity >= 2 * n_in_max && (capacity & (capacity - 1)) == 0 && capacity <= (1ll << 32),
              "imf_downsample: capacity must be a power of two >= 2n");
  hipStream_t st = (hipStream_t)stream;
  int32_t *slot_of = (int32_t *)workspace;
  int32_t *block_sums = slot_of + n_in_max;
  int rc = init_table(table, capacity, st);
  if (rc) return rc;
  k_insert_coords<<<(int)div_up(n_in_max, 256), 256, 0, st>>>(
      coords_in, n_in_dev, out_stride, table, (uint32_t)(capacity - 1), slot_of);
  IMF_CHECK_LAUNCH("k_insert_coords");
  return run_unique_tail(slot_of, block_sums, table, n_in_max, n_in_dev, coords_out, nullptr,
                         m_out, st);
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Arena layout shared by the sizing query and the build.  row_caps == NULL: every level is sized for n rows and
// an n-point table (the exact-size path: a level cannot have more rows than points).  row_caps given (capacity
// mode): level l holds at most row_caps[l] rows and its table is sized for the rows that can be INSERTED into
// it -- n points at level 0, row_caps[l-1] rows above -- so an over-full level can never fill a table.
struct PyramidLayout {
  size_t coords_off[8], table_off[8];
  int64_t cap[8], rows[8];
  size_t first_off, slot_off, total;
};

static PyramidLayout pyramid_layout(int64_t n, int n_levels, const int64_t *row_caps) {
  PyramidLayout L;
  size_t p = 0;
  for (int l = 0; l < n_levels; ++l) {
    L.rows[l] = row_caps ? row_caps[l] : n;
    L.cap[l] = imf_hash_capacity(row_caps && l > 0 ? row_caps[l - 1] : n);
    L.coords_off[l] = p;  p += align_up((size_t)L.rows[l] * 16, 256);
  }
  // the tables of all levels are contiguous so that one launch initialises them
  for (int l = 0; l < n_levels; ++l) { L.table_off[l] = p; p += align_up((size_t)L.cap[l] * sizeof(imf_slot), 256); }
  L.first_off = p;  p += align_up((size_t)L.rows[0] * 4, 256);
  L.slot_off = p;   p += align_up(imf_unique_workspace_bytes(n), 256);
  L.total = p;
  return L;
}

size_t imf_pyramid_arena_bytes(int64_t n, int n_levels) {
  if (n <= 0 || n_levels < 1 || n_levels > 8) return 0;
  return pyramid_layout(n, n_levels, nullptr).total;
}

size_t imf_pyramid_arena_bytes_caps(int64_t n_points_cap, int n_levels, const int64_t *row_caps) {
  if (n_points_cap <= 0 || n_levels < 1 || n_levels > 8 || !row_caps) return 0;
  return pyramid_layout(n_points_cap, n_levels, row_caps).total;
}

// the tables of all levels are one contiguous region: one grid-stride launch empties them and resets the meta block
__global__ void __launch_bounds__(256)
k_init_tables2(imf_slot *tab, int64_t n_slots, int n_levels, int32_t *meta, int n_meta, uint4 *zero, int64_t n_zero16) {
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (int64_t)gridDim.x * blockDim.x;
  if (i0 < n_meta) meta[i0] = (i0 >= 2 * n_levels && i0 < 2 * n_levels + 8)
                                  ? (i0 < 2 * n_levels + 4 ? 0x7FFFFFFF : -0x7FFFFFFF - 1)    // level-0 bounding box
                                  : (i0 >= 2 * n_levels + 8 ? -1 : 0);                       // item starts / counts
  for (int64_t i = i0; i < n_slots; i += step)
    reinterpret_cast<uint4 *>(tab)[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0x7FFFFFFFu, 0u);
  for (int64_t i = i0; i < n_zero16; i += step) zero[i] = make_uint4(0u, 0u, 0u, 0u);   // conv1's occupancy bit grid
}

}  // extern "C"

namespace imf {

int pyramid_prepare(PyramidBuild &b, const void *xyz, int xyz_is_f64, int64_t n, double voxel_size, int batch_index,
                    const int64_t *item_starts, int n_items, int n_levels, void *arena, size_t arena_bytes,
                    int32_t *meta, imf_level *levels_out, const int32_t *dyn, const int64_t *row_caps) {
  IMF_REQUIRE(xyz && arena && meta && levels_out, "imf_pyramid_build: null pointer");
  IMF_REQUIRE(n > 0 && n < (1ll << 31) - 2048, "imf_pyramid_build: n=%lld out of range", (long long)n);
  IMF_REQUIRE(n_levels >= 1 && n_levels <= 8, "imf_pyramid_build: n_levels=%d", n_levels);
  IMF_REQUIRE(voxel_size > 0.0, "imf_pyramid_build: voxel_size must be > 0");
  IMF_REQUIRE(batch_index >= 0 && batch_index < 512, "imf_pyramid_build: batch_index out of [0,512)");
  if (row_caps)
    for (int l = 0; l < n_levels; ++l)
      IMF_REQUIRE(row_caps[l] > 0 && row_caps[l] <= n && (l == 0 || row_caps[l] <= row_caps[l - 1]),
                  "imf_pyramid_build: row capacity of level %d", l);
  memset(&b, 0, sizeof(b));
  BatchStarts &bs = *reinterpret_cast<BatchStarts *>(b.batch_starts);
  static_assert(sizeof(BatchStarts) <= sizeof(b.batch_starts), "PyramidBuild::batch_starts too small");
  bs.nb = n_items;
  for (int i = 0; i < n_items && item_starts; ++i) bs.start[i] = item_starts[i];
  const PyramidLayout lay = pyramid_layout(n, n_levels, row_caps);
  IMF_REQUIRE(arena_bytes >= lay.total, "imf_pyramid_build: arena too small");
  IMF_REQUIRE(((uintptr_t)arena & 255) == 0, "imf_pyramid_build: arena must be 256-byte aligned");
  char *base = (char *)arena;
  for (int l = 0; l < n_levels; ++l) {
    imf_level &L = levels_out[l];
    L.coords = (int32_t *)(base + lay.coords_off[l]);
    L.table = (imf_slot *)(base + lay.table_off[l]);
    L.capacity = lay.cap[l];
    L.first_idx = nullptr;
    L.cap_rows = lay.rows[l];
    L.tensor_stride = 1 << l;
    b.row_cap[l] = row_caps ? row_caps[l] : 0;
  }
  levels_out[0].first_idx = (int32_t *)(base + lay.first_off);
  b.xyz = xyz; b.xyz_is_f64 = xyz_is_f64; b.n = n; b.voxel = voxel_size; b.batch_index = batch_index;
  b.n_levels = n_levels; b.meta = meta; b.levels = levels_out; b.dyn = dyn;
  b.slot_of = (int32_t *)(base + lay.slot_off);
  b.block_sums = b.slot_of + n;
  b.batched = n_items > 1 || dyn != nullptr;
  b.n_meta = 2 * n_levels + 8 + (b.batched ? IMF_MAX_BATCH * n_levels : 0);
  b.n_table_slots = (int64_t)((lay.first_off - lay.table_off[0]) / sizeof(imf_slot));
  return IMF_OK;
}

int pyramid_init(const PyramidBuild &b, hipStream_t st) {
  imf_level *lv = b.levels;
  int64_t nb = div_up(b.n_table_slots, 256 * 4);
  nb = nb < 1 ? 1 : (nb > 4096 ? 4096 : nb);
  // (+ conv1's bit grid, when the caller handed one over: 16-byte aligned, a whole number of 16-byte words)
  k_init_tables2<<<(unsigned)nb, 256, 0, st>>>(lv[0].table, b.n_table_slots, b.n_levels, b.meta, b.n_meta,
                                               reinterpret_cast<uint4 *>(b.grid), b.grid ? (int64_t)(b.grid_words / 4) : 0);
  IMF_CHECK_LAUNCH("k_init_tables2");
  return IMF_OK;
}

int pyramid_level0(const PyramidBuild &b, hipStream_t st, bool init) {
  const BatchStarts &bs = *reinterpret_cast<const BatchStarts *>(b.batch_starts);
  imf_level *lv = b.levels;
  if (init) {
    const int rc = pyramid_init(b, st);
    if (rc) return rc;
  }
  const int nblk = (int)div_up(b.n, kInsPoints);
  // [<= div_up(n, 256)][8] inside the unique workspace, 16-byte aligned (slot_of starts 256-byte aligned; k_flag_first reads int4)
  int32_t *const wg_bbox = b.slot_of + (b.n + div_up(b.n, kScanTile) + 16 + 3) / 4 * 4;
  if (b.xyz_is_f64)
    k_insert_points_wg<double><<<nblk, kInsThreads, 0, st>>>((const double *)b.xyz, b.n, b.voxel, b.batch_index, bs, b.dyn,
                                                             lv[0].table, (uint32_t)(lv[0].capacity - 1), b.slot_of,
                                                             b.meta + 1, wg_bbox);
  else
    k_insert_points_wg<float><<<nblk, kInsThreads, 0, st>>>((const float *)b.xyz, b.n, b.voxel, b.batch_index, bs, b.dyn,
                                                            lv[0].table, (uint32_t)(lv[0].capacity - 1), b.slot_of,
                                                            b.meta + 1, wg_bbox);
  IMF_CHECK_LAUNCH("k_insert_points_wg");
  GridFill gf;
  memset(&gf, 0, sizeof(gf));
  gf.grid = b.grid; gf.words_cap = b.grid_words; gf.bbox = b.meta + 2 * b.n_levels; gf.ksize = b.grid_ksize; gf.err = b.meta + 1;
  return run_unique_tail(b.slot_of, b.block_sums, lv[0].table, b.n, b.dyn, lv[0].coords, lv[0].first_idx,
                         b.meta, st, b.grid ? &gf : nullptr, b.row_cap[0], wg_bbox, nblk, b.meta + 2 * b.n_levels);
}

int pyramid_coarse_level(const PyramidBuild &b, int l, hipStream_t st) {
  imf_level *lv = b.levels;
  const int64_t n_in = lv[l - 1].cap_rows;
  k_insert_coords<<<(unsigned)div_up(n_in, 256), 256, 0, st>>>(lv[l - 1].coords, b.meta + 2 * (l - 1), 1 << l, lv[l].table,
                                                              (uint32_t)(lv[l].capacity - 1), b.slot_of);
  IMF_CHECK_LAUNCH("k_insert_coords");
  return run_unique_tail(b.slot_of, b.block_sums, lv[l].table, n_in, b.meta + 2 * (l - 1), lv[l].coords, nullptr,
                         b.meta + 2 * l, st, nullptr, b.row_cap[l]);
}

// where every item's rows begin at every level: meta[2L+8 + IMF_MAX_BATCH*l + b] (-1 = no row)
int pyramid_item_starts(const PyramidBuild &b, hipStream_t st, int l_begin, int l_end) {
  if (!b.batched || l_end <= l_begin) return IMF_OK;
  IMF_REQUIRE(l_end - l_begin <= 4, "pyramid_item_starts: at most 4 levels per launch");
  int32_t *starts = b.meta + 2 * b.n_levels + 8;
  ItemJobs js;
  memset(&js, 0, sizeof(js));
  int nblk = 0;
  for (int j = 0; j < 4; ++j) {
    const int l = l_begin + j < l_end ? l_begin + j : l_begin;      // (unused entries repeat the first: valid pointers)
    js.coords[j] = b.levels[l].coords;
    js.n_dev[j] = b.meta + 2 * l;
    js.starts[j] = starts + IMF_MAX_BATCH * l;
    js.blk0[j] = nblk;
    if (l_begin + j < l_end) nblk += (int)div_up(b.levels[l].cap_rows, 256);
  }
  js.blk0[4] = nblk;
  k_item_starts<<<(unsigned)nblk, 256, 0, st>>>(js);
  IMF_CHECK_LAUNCH("k_item_starts");
  return IMF_OK;
}

}  // namespace imf

static int pyramid_build_impl(const void *xyz, int xyz_is_f64, int64_t n, double voxel_size, int batch_index,
                              const int64_t *item_starts, int n_items, int n_levels, void *arena, size_t arena_bytes,
                              int32_t *meta, imf_level *levels_out, void *stream, const int32_t *dyn = nullptr,
                              const int64_t *row_caps = nullptr) {
  PyramidBuild b;
  int rc = pyramid_prepare(b, xyz, xyz_is_f64, n, voxel_size, batch_index, item_starts, n_items, n_levels, arena,
                           arena_bytes, meta, levels_out, dyn, row_caps);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if ((rc = pyramid_level0(b, st))) return rc;
  for (int l = 1; l < n_levels; ++l)
    if ((rc = pyramid_coarse_level(b, l, st))) return rc;
  return pyramid_item_starts(b, st, 0, n_levels);
}

extern "C" {

int imf_pyramid_build(const void *xyz, int xyz_is_f64, int64_t n, double voxel_size, int batch_index,
                      int n_levels, void *arena, size_t arena_bytes, int32_t *meta,
                      imf_level *levels_out, void *stream) {
  return pyramid_build_impl(xyz, xyz_is_f64, n, voxel_size, batch_index, nullptr, 1, n_levels, arena, arena_bytes, meta,
                            levels_out, stream);
}

int imf_pyramid_build_batched(const void *xyz, int xyz_is_f64, int64_t n, double voxel_size,
                              const int64_t *item_starts, int n_items, int n_levels, void *arena,
                              size_t arena_bytes, int32_t *meta, imf_level *levels_out, void *stream) {
  IMF_REQUIRE(item_starts && n_items >= 1 && n_items <= IMF_MAX_BATCH, "imf_pyramid_build_batched: n_items=%d (1..%d)",
              n_items, IMF_MAX_BATCH);
  for (int b = 0; b < n_items; ++b)
    IMF_REQUIRE(item_starts[b] >= 0 && item_starts[b] < n && (b == 0 ? item_starts[0] == 0 : item_starts[b] > item_starts[b - 1]),
                "imf_pyramid_build_batched: item_starts must start at 0 and ascend strictly below n");
  return pyramid_build_impl(xyz, xyz_is_f64, n, voxel_size, 0, item_starts, n_items, n_levels, arena, arena_bytes, meta,
                            levels_out, stream);
}

int imf_pyramid_build_dyn(const void *xyz, int xyz_is_f64, const int32_t *dyn, int64_t n_points_cap,
                          const int64_t *row_caps, double voxel_size, int n_levels, void *arena, size_t arena_bytes,
                          int32_t *meta, imf_level *levels_out, void *stream) {
  IMF_REQUIRE(dyn && row_caps, "imf_pyramid_build_dyn: null pointer");
  return pyramid_build_impl(xyz, xyz_is_f64, n_points_cap, voxel_size, 0, nullptr, 1, n_levels, arena, arena_bytes, meta,
                            levels_out, stream, dyn, row_caps);
}

int imf_gather_points(const void *xyz, int xyz_is_f64, const int32_t *first_idx, const int32_t *m_dev, int64_t m_cap,
                      double *out, void *stream) {
  IMF_REQUIRE(xyz && first_idx && out && m_cap > 0, "imf_gather_points: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const unsigned nb = (unsigned)div_up(3 * m_cap, 256);
  if (xyz_is_f64) k_gather_points<double><<<nb, 256, 0, st>>>((const double *)xyz, first_idx, m_dev, m_cap, out);
  else k_gather_points<float><<<nb, 256, 0, st>>>((const float *)xyz, first_idx, m_dev, m_cap, out);
  IMF_CHECK_LAUNCH("k_gather_points");
  return IMF_OK;
}

int64_t imf_rulebook_slots(int64_t n_out) { return div_up(n_out, IMF_TILE_ROWS) * IMF_TILE_ROWS; }

static int rulebook_conv_impl(const imf_slot *in_table, int64_t in_capacity,
                              const int32_t *out_coords, int64_t n_out, const int32_t *n_out_dev, int ts_in, int ksize,
                              int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask, void *stream) {
  IMF_REQUIRE(in_table && out_coords && tile_rows && nbr && tile_mask,
              "imf_rulebook_conv: null pointer");
  IMF_REQUIRE(ksize == 1 || ksize == 3 || ksize == 5, "imf_rulebook_conv: ksize must be 1, 3 or 5");
  IMF_REQUIRE(n_out > 0 && ts_in >= 1, "imf_rulebook_conv: bad n_out / ts_in");
  IMF_REQUIRE((in_capacity & (in_capacity - 1)) == 0, "imf_rulebook_conv: capacity not a power of 2");
  hipStream_t st = (hipStream_t)stream;
  const int kvol = ksize * ksize * ksize;
  const int64_t n_slots = imf_rulebook_slots(n_out);
  k_rulebook<+1, false><<<(unsigned)(n_slots / IMF_TILE_ROWS), 256, 0, st>>>(
      in_table, (uint32_t)(in_capacity - 1), out_coords, n_out, n_out_dev, ts_in, ksize, kvol,
      tile_rows, nbr, tile_mask, n_slots);
  IMF_CHECK_LAUNCH("k_rulebook");
  return IMF_OK;
}

int imf_rulebook_conv(const imf_slot *in_table, int64_t in_capacity,
                      const int32_t *out_coords, int64_t n_out, int ts_in, int ksize,
                      int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask, void *stream) {
  return rulebook_conv_impl(in_table, in_capacity, out_coords, n_out, nullptr, ts_in, ksize, tile_rows, nbr,
                            tile_mask, stream);
}

int imf_rulebook_conv_dyn(const imf_slot *in_table, int64_t in_capacity,
                          const int32_t *out_coords, int64_t n_out_cap, const int32_t *n_out_dev, int ts_in,
                          int ksize, int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask, void *stream) {
  IMF_REQUIRE(n_out_dev, "imf_rulebook_conv_dyn: null pointer");
  return rulebook_conv_impl(in_table, in_capacity, out_coords, n_out_cap, n_out_dev, ts_in, ksize, tile_rows,
                            nbr, tile_mask, stream);
}

int64_t imf_rulebook_transpose_slots(int64_t n_fine) {
  return (div_up(n_fine, IMF_TILE_ROWS) + 8) * IMF_TILE_ROWS;
}

static int rulebook_transpose_impl(const imf_slot *coarse_table, int64_t coarse_capacity,
                                   const int32_t *fine_coords, int64_t n_fine, const int32_t *n_fine_dev, int ts_fine,
                                   int ksize, int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask, int64_t n_slots,
                                   int32_t *counters, void *stream) {
  IMF_REQUIRE(coarse_table && fine_coords && tile_rows && nbr && tile_mask && counters,
              "imf_rulebook_transpose: null pointer");
  IMF_REQUIRE(ksize == 3, "imf_rulebook_transpose: only kernel_size 3 / stride 2 is supported");
  IMF_REQUIRE(n_fine > 0 && ts_fine >= 1, "imf_rulebook_transpose: bad n_fine / ts_fine");
  IMF_REQUIRE(n_slots == imf_rulebook_transpose_slots(n_fine), "imf_rulebook_transpose: n_slots");
  IMF_REQUIRE((coarse_capacity & (coarse_capacity - 1)) == 0, "capacity not a power of 2");
  hipStream_t st = (hipStream_t)stream;
  const int kvol = 27;
  k_init_transpose<<<(unsigned)div_up(n_slots, 256), 256, 0, st>>>(
      counters, tile_rows, n_slots, tile_mask, n_slots / IMF_TILE_ROWS * IMF_MASK_WORDS);
  const unsigned nb = (unsigned)div_up(n_fine, 256);
  int32_t *blockcnt = nbr;   // scratch: the neighbour table is written by k_rulebook below (nb*8 <= 27*n_slots)
  k_class_count<<<nb, 256, 0, st>>>(fine_coords, n_fine, n_fine_dev, ts_fine, blockcnt);
  k_class_bases<<<1, 256, 0, st>>>(blockcnt, (int)nb, counters);
  k_class_assign<<<nb, 256, 0, st>>>(fine_coords, n_fine, n_fine_dev, ts_fine, counters, blockcnt, tile_rows);
  k_rulebook<-1, true><<<(unsigned)(n_slots / IMF_TILE_ROWS), 256, 0, st>>>(
      coarse_table, (uint32_t)(coarse_capacity - 1), fine_coords, n_fine, n_fine_dev, ts_fine, ksize,
      kvol, tile_rows, nbr, tile_mask, n_slots);
  IMF_CHECK_LAUNCH("transpose rulebook");
  return IMF_OK;
}

int imf_rulebook_transpose(const imf_slot *coarse_table, int64_t coarse_capacity, const int32_t *fine_coords, int64_t n_fine,
                           int ts_fine, int ksize, int32_t *tile_rows, int32_t *nbr,
                           uint32_t *tile_mask, int64_t n_slots, int32_t *counters, void *stream) {
  return rulebook_transpose_impl(coarse_table, coarse_capacity, fine_coords, n_fine, nullptr, ts_fine, ksize,
                                 tile_rows, nbr, tile_mask, n_slots, counters, stream);
}

int imf_rulebook_transpose_dyn(const imf_slot *coarse_table, int64_t coarse_capacity,
                               const int32_t *fine_coords, int64_t n_fine_cap, const int32_t *n_fine_dev, int ts_fine,
                               int ksize, int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask, int64_t n_slots,
                               int32_t *counters, void *stream) {
  IMF_REQUIRE(n_fine_dev, "imf_rulebook_transpose_dyn: null pointer");
  return rulebook_transpose_impl(coarse_table, coarse_capacity, fine_coords, n_fine_cap, n_fine_dev, ts_fine,
                                 ksize, tile_rows, nbr, tile_mask, n_slots, counters, stream);
}

}  // extern "C"
