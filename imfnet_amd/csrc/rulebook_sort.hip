// Occupancy-sorted rulebooks: the SLOTS of a kernel map re-ordered so that the rows of a 16-row block have similar
// neighbour-occupancy patterns (round 5: tiles; round 6: blocks, every map, one launch per sort).
//
// Why.  The convolution kernels walk, per unit of rows, every kernel offset at which ANY of its rows has an input, and
// since round 6 leave out the 16-row blocks of a sub-stage that have none.  In slot = row order 96 % of the (block, offset)
// pairs of a stride-1 level are active although only 52 % of its (row, offset) pairs exist (a surface: every voxel misses
// the offsets off its sheet, but neighbours in first-occurrence order miss different ones): 1.86 issued multiply-adds per
// useful one.  Rows sorted by their occupancy pattern share their missing offsets: 1.32 (strided maps: 2.5 -> 1.5).
// Rows keep their numbers -- tile_rows[slot] = row, nbr[k][slot] = that row's input -- so features, skip connections and
// the descriptors' order are untouched; a row's sum is formed over its unit's offset list, so the partition (not the set) of
// its terms changes with the map, as between any two tile layouts.
//
// The order.  Inside windows of 16 384 consecutive slots (a window keeps a unit's gathers inside ~4 MB of the feature
// matrix; a global sort is 5 % better on paper and spreads a tile over the whole level), stable, by the 18-bit key
//   r    = the 12 edge offsets (two coordinates differ) in bits 17 .. 6, the 6 face offsets in bits 5 .. 0
//                                              -- the most evenly split offsets decide first
//   key  = gray^-1(r) = r ^ r >> 1 ^ r >> 2 ^ ...   -- neighbours in the order differ in ONE of the deciding offsets
//          (numeric order of r: 1.35; this order: 1.32; all 27 bits, four passes: 1.30); slots >= the row count keep their
//          place at the window's end.
// Measured on the S50k fragment (tools: LAB_NOTES round 6): issued / useful multiply-adds with 16-row blocks, identity
// order -> this order: stride-1 maps of the four levels 1.86 / 1.87 / 1.92 / 1.97 -> 1.32 / 1.33 / 1.48 / 1.67, strided maps
// 2.48 / 2.53 / 2.66 -> 1.46 / 1.58 / 1.88.
//
// The sort.  One 1024-thread workgroup per window, everything in LDS and registers (k_rbs_window_sort below): three stable
// LSD passes of 6 bits over 32-bit words key << 14 | slot; deterministic, and equal to a stable comparison sort (the CPU
// twin, oracle/imf_cpu_twins.c).  Round 5 used rocPRIM's device radix sort of 64-bit keys (eight launches per map); this is
// a key launch, one sort launch per map and the gather (first version of this round: keys and indices in separate arrays, two dependent LDS
// reads per element and sweep, 76-90 us per map; this one: ~25 us).
//   out: tile_rows[s] = perm[s] (or -1), nbr[k][s] = nbr_in[k][perm[s]], tile_mask = OR over each tile's 64 slots
// The input map is in identity slot order (imf_rulebook_conv); in capacity mode the row count is read from the device and
// every slot beyond it sorts last (its input slice is never read).
#include <hip/hip_runtime.h>
#include <cstring>

#include "common.h"

namespace imf {
namespace {

constexpr int kWindowShift = 14;                     // 16 384 slots per sort window
constexpr int kWindow = 1 << kWindowShift;
constexpr int kSortThreads = 1024, kSortWaves = kSortThreads / 64, kPerThread = kWindow / kSortThreads;   // 16, 16
constexpr int kDigitBits = 6, kDigits = 1 << kDigitBits, kPasses = 3, kKeyBits = kDigitBits * kPasses;    // 18-bit key
constexpr int kMaxGroups = kWindow / kSortWaves / 64;                                                      // 16 per wavefront

// bit of the 27-bit occupancy mask -> bit of r (-1: not part of the key)
__device__ __forceinline__ constexpr int key_bit_of_offset(int k) {
  constexpr int edges[12] = {1, 3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25};
  constexpr int faces[6] = {4, 10, 12, 14, 16, 22};
  for (int i = 0; i < 12; ++i)
    if (edges[i] == k) return 17 - i;
  for (int i = 0; i < 6; ++i)
    if (faces[i] == k) return 5 - i;
  return -1;
}

__device__ __forceinline__ unsigned gray_inverse18(unsigned r) {
  r ^= r >> 1;
  r ^= r >> 2;
  r ^= r >> 4;
  r ^= r >> 8;
  r ^= r >> 16;
  return r & 0x3FFFFu;
}

// Keys of all slots, by the whole chip: one CU reading 18 offsets x 16 k slots (1.2 MB) through its own L1 path took ~15 us of the
// window sort; 400 workgroups do the level-0 map's 7.4 MB in 3-4 us.  keys[s] = gray^-1(r) for s < n (rows), undefined beyond.
template <int KVOL>
__global__ void __launch_bounds__(256)
k_rbs_keys(const int32_t *__restrict__ nbr, int kvol_rt, long long n_slots, long long n_out, const int32_t *__restrict__ n_dev,
           unsigned *__restrict__ keys) {
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long n = n_out;
  if (n_dev) n = *n_dev < n ? *n_dev : n;
  if (s >= n || s >= n_slots) return;
  const int kvol = KVOL ? KVOL : kvol_rt;
  unsigned r = 0u;
  if (KVOL == 27) {
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const int bit = key_bit_of_offset(k);
      if (bit >= 0) r |= (nbr[(long long)k * n_slots + s] >= 0 ? 1u : 0u) << bit;
    }
  } else {
    for (int k = 0; k < kvol && k < kKeyBits; ++k) r |= (nbr[(long long)k * n_slots + s] >= 0 ? 1u : 0u) << k;
  }
  keys[s] = gray_inverse18(r);
}

// One workgroup = one window.  An element travels as ONE 32-bit word, key << 14 | window-local slot, through two LDS
// arrays (2 x 64 KiB); a pass is two sweeps of the wavefront's own <= 16 groups of 64 consecutive positions:
//   A  digit of every element, the lanes of the group with the same digit (6 ballots) -> rank inside the group and count;
//      the first lane of each digit adds the count to the wavefront's histogram; word, digit, rank, count stay in registers
//   -- digit-major, wavefront-minor exclusive scan of the 16 x 64 counters --
//   B  position = wavefront's running base of the digit + rank; the last lane of each digit advances the base
// so equal digits keep their order (stable, deterministic).  The valid slots of a window are a prefix of it (slots >= the row
// count sit at the level's tail), so only ceil(valid / 1024) groups per wavefront are sorted; the padding of the last group
// carries the largest key and, being behind every valid element, stays behind (stability) -- the first `valid` outputs are
// exactly the valid slots.
__global__ void __launch_bounds__(kSortThreads)
k_rbs_window_sort(const unsigned *__restrict__ keys, long long n_slots, long long n_out,
                  const int32_t *__restrict__ n_dev, int32_t *__restrict__ perm) {
  __shared__ unsigned val0[kWindow], val1[kWindow];  // 2 x 64 KiB
  __shared__ unsigned hist[kSortWaves * kDigits], dtot[kDigits], doff[kDigits];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long wbase = (long long)blockIdx.x * kWindow;
  long long n = n_out;
  if (n_dev) n = *n_dev < n ? *n_dev : n;
  const int n_win = (int)((n_slots - wbase) < kWindow ? (n_slots - wbase) : kWindow);            // slots of this window
  const int n_valid = (int)(n - wbase < 0 ? 0 : (n - wbase < n_win ? n - wbase : n_win));         // ... that hold a row
  const int groups = ((n_valid + kSortWaves - 1) / kSortWaves + 63) / 64;                         // per wavefront, <= 16
  const int per_wave = groups * 64, n_pos = per_wave * kSortWaves;

  // ---- the window's words: key << 14 | slot (k_rbs_keys computed the keys); the padding of the last group: the largest key
#pragma unroll 4
  for (int i = 0; i < kPerThread; ++i) {
    const int loc = tid + kSortThreads * i;
    if (loc < n_pos) val0[loc] = ((loc < n_valid ? keys[wbase + loc] : (1u << kKeyBits) - 1u) << kWindowShift) | (unsigned)loc;
  }
  __syncthreads();

  unsigned *src = val0, *dst = val1;
  unsigned *const myhist = hist + wave * kDigits;
  const int chunk = wave * per_wave;
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll 1
  for (int pass = 0; pass < kPasses; ++pass) {
    const int shift = kWindowShift + kDigitBits * pass;
    for (int d = tid; d < kSortWaves * kDigits; d += kSortThreads) hist[d] = 0u;
    __syncthreads();
    unsigned word[kMaxGroups], info[kMaxGroups];      // info = digit | rank << 6 | (count - 1) << 12
#pragma unroll
    for (int g = 0; g < kMaxGroups; ++g) {
      if (g < groups) {                                // (uniform)
        const unsigned v = src[chunk + 64 * g + lane];
        const unsigned d = (v >> shift) & (kDigits - 1);
        unsigned long long same = ~0ull;
#pragma unroll
        for (int b = 0; b < kDigitBits; ++b) {
          const bool bit = (d >> b) & 1u;
          const unsigned long long bal = __ballot(bit);
          same &= bit ? bal : ~bal;
        }
        const unsigned rank = (unsigned)__builtin_popcountll(same & below), cnt = (unsigned)__builtin_popcountll(same);
        if (rank == 0) atomicAdd(&myhist[d], cnt);     // one lane per digit of the group: no two lanes share an address
        word[g] = v;
        info[g] = d | (rank << 6) | ((cnt - 1u) << 12);
      }
    }
    __syncthreads();
    if (tid < kDigits) {                               // digit-major, wavefront-minor exclusive offsets
      unsigned run = 0u;
#pragma unroll
      for (int w = 0; w < kSortWaves; ++w) {
        const unsigned c = hist[w * kDigits + tid];
        hist[w * kDigits + tid] = run;
        run += c;
      }
      unsigned incl = run;                             // exclusive scan of the 64 digit totals: this IS wavefront 0
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(incl, o, 64);
        if (tid >= o) incl += up;
      }
      doff[tid] = incl - run;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < kMaxGroups; ++g) {
      if (g < groups) {
        const unsigned d = info[g] & 63u, rank = (info[g] >> 6) & 63u, last = info[g] >> 12;
        const unsigned base = myhist[d];
        if (rank == last) myhist[d] = base + last + 1u;               // (LDS operations of a wavefront execute in order)
        dst[base + doff[d] + rank] = word[g];
      }
    }
    __syncthreads();
    unsigned *const t = src;
    src = dst;
    dst = t;
  }

  // ---- new slot wbase + p takes old slot wbase + (src[p] & 16383); slots beyond the rows: -1
  for (int loc = tid; loc < n_win; loc += kSortThreads)
    perm[wbase + loc] = loc < n_valid ? (int32_t)(wbase + (src[loc] & (kWindow - 1))) : -1;
  (void)dtot;
}

// one wavefront = one tile of the OUTPUT map: 64 consecutive new slots
__global__ void __launch_bounds__(256)
k_rbs_gather(const int32_t *__restrict__ nbr, int kvol, long long n_slots, long long n_out, const int32_t *__restrict__ n_dev,
             const int32_t *__restrict__ perm, int32_t *__restrict__ tile_rows, int32_t *__restrict__ nbr_out,
             uint32_t *__restrict__ tile_mask) {
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;                                 // n_slots is a multiple of 64: whole wavefronts leave
  long long n = n_out;
  if (n_dev) n = *n_dev < n ? *n_dev : n;
  const int r = perm[s];
  const bool valid = r >= 0 && r < n;
  tile_rows[s] = valid ? r : -1;
  uint32_t m = 0u;
  for (int k = 0; k < kvol; ++k) {
    const int v = valid ? nbr[(long long)k * n_slots + r] : -1;
    nbr_out[(long long)k * n_slots + s] = v;
    if (__ballot(v >= 0) != 0ull) m |= 1u << k;
  }
  if ((threadIdx.x & 63) == 0) {
    uint32_t *t = tile_mask + (s >> 6) * IMF_MASK_WORDS;
    t[0] = m;
    for (int w = 1; w < IMF_MASK_WORDS; ++w) t[w] = 0u;
  }
}

}  // namespace
}  // namespace imf

using namespace imf;

extern "C" {

size_t imf_rulebook_sorted_workspace_bytes(int64_t n_slots) {
  if (n_slots <= 0) return 0;
  return 2 * ((((size_t)n_slots + 63) / 64 * 64) * 4 + 255) / 256 * 256;   // the permutation, the keys
}

int imf_rulebook_sort_by_occupancy(const int32_t *nbr_in, int kvol, int64_t n_slots, int64_t n_out, const int32_t *n_out_dev,
                                   int32_t *tile_rows, int32_t *nbr_out, uint32_t *tile_mask, void *workspace,
                                   size_t workspace_bytes, void *stream) {
  IMF_REQUIRE(nbr_in && tile_rows && nbr_out && tile_mask && workspace, "imf_rulebook_sort_by_occupancy: null pointer");
  IMF_REQUIRE(kvol >= 1 && kvol <= 27, "imf_rulebook_sort_by_occupancy: kvol=%d (1 .. 27)", kvol);
  IMF_REQUIRE(n_slots > 0 && n_slots % IMF_TILE_ROWS == 0 && n_out > 0 && n_out <= n_slots && n_slots < (1ll << 31),
              "imf_rulebook_sort_by_occupancy: n_slots=%lld n_out=%lld (whole tiles)", (long long)n_slots, (long long)n_out);
  IMF_REQUIRE(workspace_bytes >= imf_rulebook_sorted_workspace_bytes(n_slots), "imf_rulebook_sort_by_occupancy: workspace %zu < %zu bytes",
              workspace_bytes, imf_rulebook_sorted_workspace_bytes(n_slots));
  IMF_REQUIRE(nbr_in != nbr_out, "imf_rulebook_sort_by_occupancy: in place");
  hipStream_t st = (hipStream_t)stream;
  int32_t *perm = (int32_t *)workspace;
  const unsigned windows = (unsigned)((n_slots + kWindow - 1) / kWindow);
  const unsigned blocks = (unsigned)((n_slots + 255) / 256);
  unsigned *keys = (unsigned *)(perm + (((size_t)n_slots + 63) / 64 * 64));
  if (kvol == 27) k_rbs_keys<27><<<blocks, 256, 0, st>>>(nbr_in, kvol, n_slots, n_out, n_out_dev, keys);
  else            k_rbs_keys<0><<<blocks, 256, 0, st>>>(nbr_in, kvol, n_slots, n_out, n_out_dev, keys);
  IMF_CHECK_LAUNCH("k_rbs_keys");
  k_rbs_window_sort<<<windows, kSortThreads, 0, st>>>(keys, n_slots, n_out, n_out_dev, perm);
  IMF_CHECK_LAUNCH("k_rbs_window_sort");
  k_rbs_gather<<<blocks, 256, 0, st>>>(nbr_in, kvol, n_slots, n_out, n_out_dev, perm, tile_rows, nbr_out, tile_mask);
  IMF_CHECK_LAUNCH("k_rbs_gather");
  return IMF_OK;
}

}  // extern "C"
