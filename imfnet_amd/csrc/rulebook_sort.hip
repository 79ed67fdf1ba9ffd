// Occupancy-sorted rulebooks (round 5): the SLOTS of a stride-1 kernel map re-ordered so that the rows of a tile have
// similar neighbour-occupancy patterns.
//
// Why.  The convolution kernels walk, per 64-row tile, every kernel offset at which ANY of its rows has an input.  In
// slot = row order 99.7 % of the (tile, offset) pairs of the stride-1 level are active although only 52 % of its
// (row, offset) pairs exist (a surface: every voxel misses the offsets off its sheet, but neighbours in first-occurrence
// order miss different ones).  Rows sorted by their 27-bit occupancy mask share their missing offsets: 78 % of the
// (tile, offset) pairs stay active when the sort runs inside windows of 16 384 consecutive rows (the window keeps a tile's
// gathers inside ~4 MB of the feature matrix: a global sort reaches 70-74 % but spreads a tile over the whole level), and
// the two 64 -> 64 layers of the stride-1 decoder block take 142 -> 113 us each on the S50k pair with NO change to a kernel
// (tools/sorted_rulebook_probe.py).  Rows keep their numbers -- tile_rows[slot] = row, nbr[k][slot] = that row's input --
// so features, skip connections and the descriptors' order are untouched; a row's sum is formed over its tile's offset
// list, so the partition (not the set) of its terms changes with the map, as between any two tile layouts.
//
//   key[s]  = s < n ? (s >> 14) << 27 | mask(s) : ~0 (64 bits)        mask bit k <=> nbr[k][s] >= 0
//   perm    = stable sort of the slots by key (rocPRIM LSD radix sort: deterministic)
//   out: tile_rows[s] = perm[s] (or -1), nbr[k][s] = nbr_in[k][perm[s]], tile_mask = OR over each tile's 64 slots
// The input map is in identity slot order (imf_rulebook_conv, stride 1); in capacity mode the row count is read from the
// device and every slot beyond it sorts last (its input slice is never read).
#include <hip/hip_runtime.h>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace imf {
namespace {

#ifndef IMF_RBS_WINDOW_SHIFT
#define IMF_RBS_WINDOW_SHIFT 14
#endif
constexpr int kWindowShift = IMF_RBS_WINDOW_SHIFT;   // 16 384 rows per sort window

__global__ void __launch_bounds__(256)
k_rbs_keys(const int32_t *__restrict__ nbr, int kvol, long long n_slots, long long n_out, const int32_t *__restrict__ n_dev,
           uint64_t *__restrict__ keys, int32_t *__restrict__ vals) {
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  long long n = n_out;
  if (n_dev) n = *n_dev < n ? *n_dev : n;
  uint64_t key = ~0ull;
  if (s < n) {
    uint32_t m = 0u;
    for (int k = 0; k < kvol; ++k) m |= (nbr[(long long)k * n_slots + s] >= 0 ? 1u : 0u) << k;
    key = ((uint64_t)(s >> kWindowShift) << 27) | m;
  }
  keys[s] = key;
  vals[s] = (int32_t)s;
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

size_t sort_temp_bytes(long long n_slots) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (const int32_t *)nullptr,
                                  (int32_t *)nullptr, (size_t)n_slots, 0u, 64u, (hipStream_t)0);
  return (bytes + 255) / 256 * 256;
}

}  // namespace
}  // namespace imf

using namespace imf;

extern "C" {

size_t imf_rulebook_sorted_workspace_bytes(int64_t n_slots) {
  if (n_slots <= 0) return 0;
  return 6 * (((size_t)n_slots * 4 + 255) / 256 * 256) + sort_temp_bytes(n_slots);   // keys (8 B) and slots (4 B), in and out
}

int imf_rulebook_sort_by_occupancy(const int32_t *nbr_in, int kvol, int64_t n_slots, int64_t n_out, const int32_t *n_out_dev,
                                   int32_t *tile_rows, int32_t *nbr_out, uint32_t *tile_mask, void *workspace,
                                   size_t workspace_bytes, void *stream) {
  IMF_REQUIRE(nbr_in && tile_rows && nbr_out && tile_mask && workspace, "imf_rulebook_sort_by_occupancy: null pointer");
  IMF_REQUIRE(kvol >= 1 && kvol <= 27, "imf_rulebook_sort_by_occupancy: kvol=%d (1 .. 27: the mask takes the key's low 27 bits)", kvol);
  IMF_REQUIRE(n_slots > 0 && n_slots % IMF_TILE_ROWS == 0 && n_out > 0 && n_out <= n_slots && n_slots < (1ll << 31),
              "imf_rulebook_sort_by_occupancy: n_slots=%lld n_out=%lld (whole tiles)", (long long)n_slots, (long long)n_out);
  IMF_REQUIRE(workspace_bytes >= imf_rulebook_sorted_workspace_bytes(n_slots), "imf_rulebook_sort_by_occupancy: workspace %zu < %zu bytes",
              workspace_bytes, imf_rulebook_sorted_workspace_bytes(n_slots));
  IMF_REQUIRE(nbr_in != nbr_out, "imf_rulebook_sort_by_occupancy: in place");
  hipStream_t st = (hipStream_t)stream;
  const size_t arr = ((size_t)n_slots * 4 + 255) / 256 * 256;
  char *w = (char *)workspace;
  uint64_t *k_in = (uint64_t *)w, *k_out = (uint64_t *)(w + 2 * arr);
  int32_t *v_in = (int32_t *)(w + 4 * arr), *v_out = (int32_t *)(w + 5 * arr);
  void *tmp = w + 6 * arr;
  size_t tmp_bytes = workspace_bytes - 6 * arr;
  const unsigned blocks = (unsigned)((n_slots + 255) / 256);
  k_rbs_keys<<<blocks, 256, 0, st>>>(nbr_in, kvol, n_slots, n_out, n_out_dev, k_in, v_in);
  IMF_CHECK_LAUNCH("k_rbs_keys");
  // only the bits that can differ are sorted: 27 mask bits + the window index's (+ 1, so that the all-ones key of the
  // padding slots stays above every window index): 32 bits = four radix passes for the pair's 10 windows
  unsigned wbits = 1;
  while ((n_slots >> kWindowShift) >> wbits) ++wbits;
  IMF_CHECK_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)n_slots, 0u, 27u + wbits + 1u, st));
  k_rbs_gather<<<blocks, 256, 0, st>>>(nbr_in, kvol, n_slots, n_out, n_out_dev, v_out, tile_rows, nbr_out, tile_mask);
  IMF_CHECK_LAUNCH("k_rbs_gather");
  return IMF_OK;
}

}  // extern "C"
