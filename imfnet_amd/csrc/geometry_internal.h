// Pieces of the pyramid build that the fragment-level executor (executor.hip) schedules itself: level 0 on
// the main stream, the coarser levels interleaved with the rulebook builds on the side stream.
#pragma once
#include "common.h"

namespace imf {

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
