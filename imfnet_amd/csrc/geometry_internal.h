// Pieces of the pyramid build that the fragment-level executor (executor.hip) schedules itself: level 0 on
// the main stream, the coarser levels interleaved with the rulebook builds on the side stream.
#pragma once
#include "common.h"

namespace imf {

// ---- conv1's occupancy bit grid (spconv.hip: k_conv_first_bits; geometry.hip: filled by the level-0 compaction) ----
struct GridDesc {
  int b0, x0, y0, z0;      // origin (bounding-box min minus the kernel radius)
  int nb, nx, ny, nz;      // extent in voxels (margins included)
  int row_words;           // 32-bit words per x-row (>= nx/32 + 2: an unaligned window never leaves the row)
};

__host__ __device__ inline bool grid_desc_from_bbox(const int32_t *bbox, int ksize, GridDesc &g, size_t &words) {
  const int r = ksize >> 1;
  g.b0 = bbox[0]; g.x0 = bbox[1] - r; g.y0 = bbox[2] - r; g.z0 = bbox[3] - r;
  g.nb = bbox[4] - bbox[0] + 1;
  g.nx = bbox[5] - bbox[1] + 1 + 2 * r; g.ny = bbox[6] - bbox[2] + 1 + 2 * r; g.nz = bbox[7] - bbox[3] + 1 + 2 * r;
  if (g.nb <= 0 || g.nx <= 0 || g.ny <= 0 || g.nz <= 0) return false;
  g.row_words = g.nx / 32 + 2;
  const double w = (double)g.nb * g.nz * g.ny * g.row_words;
  if (w > (double)(1ull << 28)) return false;            // > 1 GiB of grid: use the hash path
  words = (size_t)w;
  return true;
}

// capacity mode: the grid descriptor is derived on the device from the level's bounding box (meta block of
// imf_pyramid_build); a box that does not fit the provided grid raises bit 2 of *err and the launch does nothing
struct DynGrid {
  const int32_t *n_dev, *bbox_dev;
  int32_t *err;
  unsigned long long words_cap;
  const float *w_image = nullptr;   // conv1's hi / lo f16 weight image (imf_pack_first_kernel), or NULL: split in the kernel
};

__device__ __forceinline__ bool dyn_grid(const DynGrid &d, int ksize, GridDesc &g, long long &n) {
  if (!d.bbox_dev) return true;
  const long long nd = *d.n_dev;
  n = nd < n ? nd : n;
  int32_t bb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bb[i] = d.bbox_dev[i];
  size_t words = 0;
  if (n <= 0) return false;
  if (!grid_desc_from_bbox(bb, ksize, g, words) || words > d.words_cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(d.err, 4);
    return false;
  }
  return true;
}

__device__ __forceinline__ long long grid_row(const GridDesc &g, int b, int y, int z) {
  return ((((long long)(b - g.b0) * g.nz + (z - g.z0)) * g.ny + (y - g.y0)) * g.row_words);
}

struct PyramidBuild {
  const void *xyz;
  int xyz_is_f64;
  int64_t n;                 // points (capacity in capacity mode)
  double voxel;
  int batch_index, n_levels;
  int32_t *meta;
  int n_meta;
  imf_level *levels;         // [host] caller-owned array, filled by pyramid_prepare
  const int32_t *dyn;        // device: n_points, n_items, item starts (capacity mode) or NULL
  int64_t row_cap[8];        // 0 = exact-size mode
  int32_t *slot_of, *block_sums;
  bool batched;
  int64_t n_table_slots;     // slots of all levels' tables (one contiguous region)
  uint32_t *grid;            // optional (imf_fragment_forward): conv1's occupancy bit grid, zeroed by the caller; the level-0
  size_t grid_words;         // compaction sets a bit per voxel (capacity in words; overflow raises IMF_FLAG_BITGRID in meta[1])
  int grid_ksize;
  alignas(8) char batch_starts[8 * IMF_MAX_BATCH + 16];
};

int pyramid_prepare(PyramidBuild &b, const void *xyz, int xyz_is_f64, int64_t n, double voxel_size, int batch_index,
                    const int64_t *item_starts, int n_items, int n_levels, void *arena, size_t arena_bytes,
                    int32_t *meta, imf_level *levels_out, const int32_t *dyn, const int64_t *row_caps);
// hash tables empty, meta block (counts, flag words, bounding box, item starts) reset -- the first launch of level 0
int pyramid_init(const PyramidBuild &b, hipStream_t st);
int pyramid_level0(const PyramidBuild &b, hipStream_t st, bool init = true);
int pyramid_coarse_level(const PyramidBuild &b, int l, hipStream_t st);
int pyramid_item_starts(const PyramidBuild &b, hipStream_t st, int l_begin, int l_end);

}  // namespace imf
