// The neighbour-map builder's per-tile body (geometry.hip: k_rulebook; spconv.hip: the launch that builds the level-0
// 3x3x3 map beside conv1).
#pragma once
#include "common.h"

namespace imf {

// ---- rulebooks ---------------------------------------------------------------------------------
__device__ __forceinline__ void kernel_offset(int k, int ksize, int &dx, int &dy, int &dz) {
  const int r = ksize >> 1;              // ME kernel_region: axis 0 (x) fastest
  dx = k % ksize - r;
  dy = (k / ksize) % ksize - r;
  dz = k / (ksize * ksize) - r;
}

// One workgroup per 64-slot tile; wavefront g of its four takes the offsets k = g, g + 4, g + 8, ... (lane = slot), so
//   * a row's coordinates are loaded once per thread, not once per (slot, offset);
//   * the probes of a thread's (up to 7 of 27) offsets are independent and in flight together;
//   * every nbr store of a wavefront is one coalesced 256-byte line (64 consecutive slots of one offset);
//   * the tile's active-offset mask is OR-ed in registers / LDS and written by ONE plain store: no atomics, and no
//     memset launch ahead of the kernel (round 3; the (slot, k)-thread kernel before it: 47 us for the level-0 3x3x3 map
//     of the pair + a 4 us memset, on the critical path since conv1 got shorter than it).
// SIGN = +1: in = out + off*ts (conv); SIGN = -1: coarse = fine - off*ts (transposed conv).
template <int SIGN, bool INDIRECT>
__device__ __forceinline__ void rulebook_tile(const imf_slot *__restrict__ tab, uint32_t capmask,
                                              const int32_t *__restrict__ out_coords, int64_t n_out,
                                              const int32_t *__restrict__ n_out_dev, int ts, int ksize, int kvol,
                                              int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask, int64_t n_slots,
                                              const int64_t tile) {
  __shared__ uint32_t wmask[4][IMF_MASK_WORDS];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t slot = tile * IMF_TILE_ROWS + lane;
  if (n_out_dev) n_out = min((int64_t)*n_out_dev, n_out);   // capacity-sized table, actual rows on the device
  int row;
  if (INDIRECT) {
    row = tile_rows[slot];
  } else {
    row = slot < n_out ? (int)slot : -1;
    if (g == 0) tile_rows[slot] = row;
  }
  // a tile without rows (capacity padding) gets mask 0: the convolution never looks at its neighbour slice
  const bool empty_tile = __ballot(row >= 0) == 0ull;
  // the probed table's level: the input's (conv; its coordinates are multiples of ts) or the coarse one's (transposed: 2 ts)
  const int tshift = __builtin_ctz((unsigned)ts) + (SIGN < 0 ? 1 : 0);
  uint32_t m[IMF_MASK_WORDS] = {0u, 0u, 0u, 0u};
  if (!(empty_tile && n_out_dev)) {   // (exact-size tables: padding tiles are written as 'no input' too)
    int4 c = make_int4(0, 0, 0, 0);
    if (row >= 0) c = reinterpret_cast<const int4 *>(out_coords)[row];
    if (kvol <= 28) {
      // 3x3x3 (and 1x1x1): the thread's <= 7 offsets as two phases -- every first-slot load issued before any is
      // looked at (one 16-byte slot = key + row), then the rare collisions walk on
      constexpr int KPT = 7;
      uint64_t want[KPT];
      uint32_t hs[KPT];
      uint4 got[KPT];
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const int k = g + 4 * j;
        want[j] = kEmptyKey;                           // "no probe": resolves to -1 below
        hs[j] = 0;
        if (k < kvol && row >= 0) {
          int dx, dy, dz;
          kernel_offset(k, ksize, dx, dy, dz);
          const int x = c.y + SIGN * dx * ts, y = c.z + SIGN * dy * ts, z = c.w + SIGN * dz * ts;
          if (coord_in_range(x, y, z)) {
            want[j] = pack_key(c.x, x, y, z);
            hs[j] = hash_slot(want[j], tshift, capmask);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < KPT; ++j) got[j] = *reinterpret_cast<const uint4 *>(tab + hs[j]);
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const int k = g + 4 * j;
        if (k >= kvol) continue;                       // wavefront-uniform
        int found = -1;
        if (want[j] != kEmptyKey) {
          const uint64_t k0 = ((uint64_t)got[j].y << 32) | got[j].x;
          if (k0 == want[j]) found = (int)got[j].z;
          else if (k0 != kEmptyKey) found = hash_find(tab, capmask, want[j], tshift);   // collision: walk on
        }
        nbr[(int64_t)k * n_slots + slot] = found;
        if (__ballot(found >= 0) != 0ull) m[k >> 5] |= 1u << (k & 31);
      }
    } else {
      for (int k = g; k < kvol; k += 4) {
        int found = -1;
        if (row >= 0) {
          int dx, dy, dz;
          kernel_offset(k, ksize, dx, dy, dz);
          const int x = c.y + SIGN * dx * ts, y = c.z + SIGN * dy * ts, z = c.w + SIGN * dz * ts;
          if (coord_in_range(x, y, z)) found = hash_find(tab, capmask, pack_key(c.x, x, y, z), tshift);
        }
        nbr[(int64_t)k * n_slots + slot] = found;
        if (__ballot(found >= 0) != 0ull) m[k >> 5] |= 1u << (k & 31);
      }
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < IMF_MASK_WORDS; ++q) wmask[g][q] = m[q];
  }
  __syncthreads();
  if (threadIdx.x < IMF_MASK_WORDS)
    tile_mask[tile * IMF_MASK_WORDS + threadIdx.x] =
        wmask[0][threadIdx.x] | wmask[1][threadIdx.x] | wmask[2][threadIdx.x] | wmask[3][threadIdx.x];
}


}  // namespace imf
