/* TEST INFRASTRUCTURE ONLY -- the `imf_cpu_*` twins SURVEY.md 8(b) B3 asks for: the geometry and convolution entry
 * points of include/imfnet_hip.h with the SAME signatures, on HOST pointers, computed the way MinkowskiEngine 0.5.4's CPU
 * path does (sequential insert into a coordinate hash map => first-occurrence row order; one probe per (output row, kernel
 * offset); per-output-row gather-FMA).  `stream` and `workspace` arguments are accepted and ignored; the hash table the
 * GPU functions fill and hand on (imf_slot[capacity]) is filled and handed on here too (linear probing: which slot a key
 * lands in is nobody's business on either side).  Outputs have the GPU library's layouts bit for bit -- coords int32[n, 4],
 * first indices, offset-major neighbour tables nbr[k][slot], tile_rows, per-tile active-offset masks, parity-class grouped
 * transposed maps -- so tests/test_gpu_cpu_twins.py compares the HIP results with these DIRECTLY (integer outputs equal,
 * convolution within fp32 roundoff).  Never linked into or called by imfnet_amd/.
 *
 * Reference call sites restated (file:line under /root/reference), as in include/imfnet_hip.h:
 *   imf_cpu_voxelize            util/misc.py:82-87 (np.floor(xyz / voxel), ME.utils.sparse_quantize) + :95
 *   imf_cpu_downsample          implicit coordinate_manager.stride() of the stride-2 convolutions, model/resunet.py:54-85
 *   imf_cpu_rulebook_conv       kernel maps of ME.MinkowskiConvolution, model/resunet.py:42-99, model/residual_block.py:23-33
 *   imf_cpu_rulebook_transpose  kernel maps of ME.MinkowskiConvolutionTranspose, model/resunet.py:101-134
 *   imf_cpu_spconv_fwd_abi      conv + MinkowskiBatchNorm + relu + residual + cat + bias + L2 norm, model/resunet.py:168-233
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/imfnet_hip.h"

#define TW_EMPTY 0xFFFFFFFFFFFFFFFFull
#define TW_BITS 18
#define TW_LIM (1 << (TW_BITS - 1))

static inline uint64_t tw_pack(int b, int x, int y, int z) {   /* the library's key: b:9 | x:18 | y:18 | z:18 */
  return ((uint64_t)(uint32_t)b << (3 * TW_BITS)) | ((uint64_t)(x & 0x3FFFF) << (2 * TW_BITS)) |
         ((uint64_t)(y & 0x3FFFF) << TW_BITS) | (uint64_t)(z & 0x3FFFF);
}
static inline uint64_t tw_mix(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}
static void tw_clear(imf_slot *t, int64_t cap) {
  for (int64_t i = 0; i < cap; ++i) { t[i].key = TW_EMPTY; t[i].val = 0x7FFFFFFF; t[i].pad = 0; }
}
/* slot of `key` after the call; *is_new says whether it was inserted (with value `next`) */
static inline int64_t tw_get_or_insert(imf_slot *t, int64_t cap, uint64_t key, int32_t next, int *is_new) {
  uint64_t s = tw_mix(key) & (uint64_t)(cap - 1);
  for (;;) {
    if (t[s].key == key) { *is_new = 0; return (int64_t)s; }
    if (t[s].key == TW_EMPTY) { t[s].key = key; t[s].val = next; *is_new = 1; return (int64_t)s; }
    s = (s + 1) & (uint64_t)(cap - 1);
  }
}
static inline int32_t tw_find(const imf_slot *t, int64_t cap, uint64_t key) {
  uint64_t s = tw_mix(key) & (uint64_t)(cap - 1);
  for (;;) {
    if (t[s].key == key) return t[s].val;
    if (t[s].key == TW_EMPTY) return -1;
    s = (s + 1) & (uint64_t)(cap - 1);
  }
}
static inline int tw_floor_div(int a, int s) { return a >= 0 ? a / s : -((-a + s - 1) / s); }
static inline int tw_pow2(int64_t c) { return c > 0 && (c & (c - 1)) == 0; }

/* imf_voxelize: m_out[0] = M; *err_out |= 1 when a point is out of range / NaN (such points get no voxel) */
int imf_cpu_voxelize(const void *xyz, int xyz_is_f64, int64_t n, double voxel_size, int batch_index, int32_t *coords,
                     int32_t *first_idx, int32_t *m_out, imf_slot *table, int64_t capacity, void *workspace,
                     int32_t *err_out, void *stream) {
  (void)workspace; (void)stream;
  if (!xyz || !coords || !first_idx || !m_out || !table || !tw_pow2(capacity) || capacity < 2 * n) return IMF_EINVAL;
  tw_clear(table, capacity);
  int32_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    double p[3];
    for (int a = 0; a < 3; ++a)
      p[a] = xyz_is_f64 ? ((const double *)xyz)[3 * i + a] : (double)((const float *)xyz)[3 * i + a];
    const double fx = floor(p[0] / voxel_size), fy = floor(p[1] / voxel_size), fz = floor(p[2] / voxel_size);
    if (!(fx >= -TW_LIM && fx < TW_LIM && fy >= -TW_LIM && fy < TW_LIM && fz >= -TW_LIM && fz < TW_LIM)) {
      if (err_out) *err_out |= 1;
      continue;
    }
    int is_new;
    tw_get_or_insert(table, capacity, tw_pack(batch_index, (int)fx, (int)fy, (int)fz), m, &is_new);
    if (is_new) {
      coords[4 * m] = batch_index; coords[4 * m + 1] = (int)fx; coords[4 * m + 2] = (int)fy; coords[4 * m + 3] = (int)fz;
      first_idx[m] = (int32_t)i;
      ++m;
    }
  }
  m_out[0] = m;
  return IMF_OK;
}

/* imf_downsample: coarse = floor(c / out_stride) * out_stride, first-occurrence order; n_in_dev: host pointer to the count or NULL */
int imf_cpu_downsample(const int32_t *coords_in, const int32_t *n_in_dev, int64_t n_in_max, int out_stride, int32_t *coords_out,
                       int32_t *m_out, imf_slot *table, int64_t capacity, void *workspace, void *stream) {
  (void)workspace; (void)stream;
  if (!coords_in || !coords_out || !m_out || !table || !tw_pow2(capacity) || out_stride < 1) return IMF_EINVAL;
  const int64_t n = n_in_dev ? (*n_in_dev < n_in_max ? *n_in_dev : n_in_max) : n_in_max;
  if (capacity < 2 * n) return IMF_EINVAL;
  tw_clear(table, capacity);
  int32_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int b = coords_in[4 * i], x = tw_floor_div(coords_in[4 * i + 1], out_stride) * out_stride,
              y = tw_floor_div(coords_in[4 * i + 2], out_stride) * out_stride,
              z = tw_floor_div(coords_in[4 * i + 3], out_stride) * out_stride;
    int is_new;
    tw_get_or_insert(table, capacity, tw_pack(b, x, y, z), m, &is_new);
    if (is_new) {
      coords_out[4 * m] = b; coords_out[4 * m + 1] = x; coords_out[4 * m + 2] = y; coords_out[4 * m + 3] = z;
      ++m;
    }
  }
  m_out[0] = m;
  return IMF_OK;
}

static void tw_kernel_offset(int k, int ksize, int *dx, int *dy, int *dz) {   /* ME kernel_region: axis 0 (x) fastest */
  const int r = ksize / 2;
  *dx = k % ksize - r; *dy = (k / ksize) % ksize - r; *dz = k / (ksize * ksize) - r;
}

/* the probe loop shared by both maps: slot s holds output row tile_rows[s] (or -1); in = out + sign * off * ts */
static void tw_fill_map(const imf_slot *tab, int64_t cap, const int32_t *out_coords, int ts, int ksize, int sign,
                        const int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask, int64_t n_slots) {
  const int kvol = ksize * ksize * ksize;
  memset(tile_mask, 0, (size_t)(n_slots / IMF_TILE_ROWS) * IMF_MASK_WORDS * sizeof(uint32_t));
#pragma omp parallel for schedule(static)
  for (int64_t tile = 0; tile < n_slots / IMF_TILE_ROWS; ++tile) {
    for (int k = 0; k < kvol; ++k) {
      int dx, dy, dz, any = 0;
      tw_kernel_offset(k, ksize, &dx, &dy, &dz);
      for (int l = 0; l < IMF_TILE_ROWS; ++l) {
        const int64_t s = tile * IMF_TILE_ROWS + l;
        const int32_t row = tile_rows[s];
        int32_t f = -1;
        if (row >= 0) {
          const int32_t *c = out_coords + 4 * (int64_t)row;
          const int x = c[1] + sign * dx * ts, y = c[2] + sign * dy * ts, z = c[3] + sign * dz * ts;
          if (x >= -TW_LIM && x < TW_LIM && y >= -TW_LIM && y < TW_LIM && z >= -TW_LIM && z < TW_LIM)
            f = tw_find(tab, cap, tw_pack(c[0], x, y, z));
        }
        nbr[(int64_t)k * n_slots + s] = f;
        any |= f >= 0;
      }
      if (any) tile_mask[tile * IMF_MASK_WORDS + (k >> 5)] |= 1u << (k & 31);
    }
  }
}

/* imf_rulebook_conv: n_slots = imf_rulebook_slots(n_out) = roundup64(n_out); slot s <-> output row s */
int imf_cpu_rulebook_conv(const imf_slot *in_table, int64_t in_capacity, const int32_t *out_coords, int64_t n_out, int ts_in,
                          int ksize, int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask, void *stream) {
  (void)stream;
  if (!in_table || !out_coords || !tile_rows || !nbr || !tile_mask || !tw_pow2(in_capacity) || n_out <= 0) return IMF_EINVAL;
  const int64_t n_slots = (n_out + IMF_TILE_ROWS - 1) / IMF_TILE_ROWS * IMF_TILE_ROWS;
  for (int64_t s = 0; s < n_slots; ++s) tile_rows[s] = s < n_out ? (int32_t)s : -1;
  tw_fill_map(in_table, in_capacity, out_coords, ts_in, ksize, +1, tile_rows, nbr, tile_mask, n_slots);
  return IMF_OK;
}

/* imf_rulebook_transpose: fine rows grouped by the parity of coord / ts per axis (8 classes, each padded to whole
 * tiles, rows of a class in ascending row order); coarse = fine - off * ts, probed in the coarse level's table */
int imf_cpu_rulebook_transpose(const imf_slot *coarse_table, int64_t coarse_capacity, const int32_t *fine_coords,
                               int64_t n_fine, int ts_fine, int ksize, int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask,
                               int64_t n_slots, int32_t *counters, void *stream) {
  (void)stream;
  if (!coarse_table || !fine_coords || !tile_rows || !nbr || !tile_mask || !counters || ksize != 3 || n_fine <= 0 ||
      n_slots != ((n_fine + IMF_TILE_ROWS - 1) / IMF_TILE_ROWS + 8) * IMF_TILE_ROWS || !tw_pow2(coarse_capacity))
    return IMF_EINVAL;
  int32_t total[8] = {0}, base[8], fill[8] = {0};
  for (int64_t i = 0; i < n_fine; ++i) {
    const int32_t *c = fine_coords + 4 * i;
    total[((c[1] / ts_fine) & 1) | (((c[2] / ts_fine) & 1) << 1) | (((c[3] / ts_fine) & 1) << 2)]++;
  }
  int32_t run = 0;
  for (int q = 0; q < 8; ++q) {
    counters[q] = total[q];
    counters[8 + q] = base[q] = run;
    run += (total[q] + IMF_TILE_ROWS - 1) / IMF_TILE_ROWS * IMF_TILE_ROWS;
  }
  for (int64_t s = 0; s < n_slots; ++s) tile_rows[s] = -1;
  for (int64_t i = 0; i < n_fine; ++i) {
    const int32_t *c = fine_coords + 4 * i;
    const int p = ((c[1] / ts_fine) & 1) | (((c[2] / ts_fine) & 1) << 1) | (((c[3] / ts_fine) & 1) << 2);
    tile_rows[base[p] + fill[p]++] = (int32_t)i;
  }
  tw_fill_map(coarse_table, coarse_capacity, fine_coords, ts_fine, ksize, -1, tile_rows, nbr, tile_mask, n_slots);
  return IMF_OK;
}

/* imf_rulebook_sort_by_occupancy (csrc/rulebook_sort.hip): the slots of a map in identity order, stably sorted inside windows of
 * 16 384 slots by key = gray^-1(r), r = the occupancy bits (nbr_in[k][slot] >= 0) of the 12 edge offsets in bits 17 .. 6 and of
 * the 6 face offsets in bits 5 .. 0 (kvol != 27: the first 18 offsets in bits 0 .. 17); slots >= the row count last in their
 * window.  A plain stable merge sort here -- the HIP side's LDS radix sort must produce the same permutation. */
static uint32_t tw_occupancy_key(const int32_t *nbr_in, int kvol, int64_t n_slots, int64_t s) {
  static const int edges[12] = {1, 3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25}, faces[6] = {4, 10, 12, 14, 16, 22};
  uint32_t r = 0;
  if (kvol == 27) {
    for (int i = 0; i < 12; ++i) r |= (uint32_t)(nbr_in[(int64_t)edges[i] * n_slots + s] >= 0) << (17 - i);
    for (int i = 0; i < 6; ++i) r |= (uint32_t)(nbr_in[(int64_t)faces[i] * n_slots + s] >= 0) << (5 - i);
  } else {
    for (int q = 0; q < kvol && q < 18; ++q) r |= (uint32_t)(nbr_in[(int64_t)q * n_slots + s] >= 0) << q;
  }
  uint32_t b = 0;
  for (; r; r >>= 1) b ^= r;                       /* gray^-1: b = r ^ r >> 1 ^ r >> 2 ^ ... */
  return b & 0x3FFFFu;
}
static void tw_merge_sort(uint64_t *key, int32_t *val, uint64_t *tk, int32_t *tv, int64_t lo, int64_t hi) {
  if (hi - lo < 2) return;
  const int64_t mid = lo + (hi - lo) / 2;
  tw_merge_sort(key, val, tk, tv, lo, mid);
  tw_merge_sort(key, val, tk, tv, mid, hi);
  int64_t i = lo, j = mid, o = lo;
  while (i < mid && j < hi) {
    if (key[j] < key[i]) { tk[o] = key[j]; tv[o++] = val[j++]; }
    else { tk[o] = key[i]; tv[o++] = val[i++]; }
  }
  while (i < mid) { tk[o] = key[i]; tv[o++] = val[i++]; }
  while (j < hi) { tk[o] = key[j]; tv[o++] = val[j++]; }
  for (int64_t q = lo; q < hi; ++q) { key[q] = tk[q]; val[q] = tv[q]; }
}

size_t imf_cpu_rulebook_sorted_workspace_bytes(int64_t n_slots) { return n_slots > 0 ? (size_t)n_slots * 24 : 0; }

int imf_cpu_rulebook_sort_by_occupancy(const int32_t *nbr_in, int kvol, int64_t n_slots, int64_t n_out, const int32_t *n_out_dev,
                                       int32_t *tile_rows, int32_t *nbr_out, uint32_t *tile_mask, void *workspace,
                                       size_t workspace_bytes, void *stream) {
  (void)stream;
  if (!nbr_in || !tile_rows || !nbr_out || !tile_mask || !workspace || kvol < 1 || kvol > 27 || n_slots <= 0 ||
      n_slots % IMF_TILE_ROWS || n_out <= 0 || n_out > n_slots || workspace_bytes < (size_t)n_slots * 24 || nbr_in == nbr_out)
    return IMF_EINVAL;
  int64_t n = n_out;
  if (n_out_dev && *n_out_dev < n) n = *n_out_dev;
  uint64_t *key = (uint64_t *)workspace, *tk = key + n_slots;
  int32_t *val = (int32_t *)(tk + n_slots), *tv = val + n_slots;
  for (int64_t s = 0; s < n_slots; ++s) {
    uint64_t k = ((uint64_t)(s >> 14) << 19) | (1u << 18);          /* padding: last inside its window */
    if (s < n) k = ((uint64_t)(s >> 14) << 19) | tw_occupancy_key(nbr_in, kvol, n_slots, s);
    key[s] = k;
    val[s] = (int32_t)s;
  }
  tw_merge_sort(key, val, tk, tv, 0, n_slots);
  for (int64_t t = 0; t < n_slots / IMF_TILE_ROWS; ++t)
    for (int w = 0; w < IMF_MASK_WORDS; ++w) tile_mask[t * IMF_MASK_WORDS + w] = 0u;
  for (int64_t s = 0; s < n_slots; ++s) {
    const int32_t r = val[s];
    const int valid = r < n;                                         /* (padding slots stay inside their window: they sit at its end) */
    tile_rows[s] = valid ? r : -1;
    for (int q = 0; q < kvol; ++q) {
      const int32_t v = valid ? nbr_in[(int64_t)q * n_slots + r] : -1;
      nbr_out[(int64_t)q * n_slots + s] = v;
      if (v >= 0) tile_mask[(s / IMF_TILE_ROWS) * IMF_MASK_WORDS] |= 1u << q;
    }
  }
  return IMF_OK;
}

/* imf_spconv_fwd for args->variant == 0 (w_packed = the imf_pack_weights fp32 image): out[o] = epilogue(sum_k in[nbr[k][o]] W[k]),
 * two sources (cat), scale / shift, residual, ReLU, L2 norm; offsets ascending, channels ascending, fp32 FMA chain */
int imf_cpu_spconv_fwd_abi(const imf_conv_args *a, void *stream) {
  (void)stream;
  if (!a || !a->in_a || !a->w_packed || !a->out || a->variant != 0 || a->geglu || a->operand_format) return IMF_EINVAL;
  const int cin = a->c_a + a->c_b, cout = a->cout, kvol = a->kvol;
  const int CI = (cin % 64 == 0) ? 64 : 32, J = CI / 16, CB = (cout % 64 == 0) ? 4 : 2, CW = 16 * CB, ncc = cin / CI;
  /* unpack the fragment-major image once: W[k][ci][co] */
  float *W = (float *)malloc((size_t)kvol * cin * cout * sizeof(float));
  if (!W) return IMF_EINVAL;
  for (int64_t idx = 0; idx < (int64_t)kvol * cin * cout; ++idx) {
    int64_t r = idx;
    const int t = r & 3; r >>= 2;
    const int lane = r & 63; r >>= 6;
    const int cb = r % CB; r /= CB;
    const int j = r % J; r /= J;
    const int cc = r % ncc; r /= ncc;
    const int k = r % kvol; r /= kvol;
    const int y = (int)r;
    const int ci = cc * CI + 16 * j + 4 * (lane >> 4) + t, co = y * CW + 16 * cb + (lane & 15);
    W[((int64_t)k * cin + ci) * cout + co] = a->w_packed[idx];
  }
#pragma omp parallel
  {
    float *acc = (float *)malloc((size_t)cout * sizeof(float));
#pragma omp for schedule(static)
    for (int64_t s = 0; s < a->n_slots; ++s) {
      const int32_t orow = a->tile_rows ? a->tile_rows[s] : (s < a->n_out ? (int32_t)s : -1);
      if (orow < 0) continue;
      for (int c = 0; c < cout; ++c) acc[c] = 0.f;
      for (int k = 0; k < kvol; ++k) {
        const int32_t irow = a->nbr ? a->nbr[(int64_t)k * a->n_slots + s] : orow;
        if (irow < 0) continue;
        for (int ci = 0; ci < cin; ++ci) {
          const float x = ci < a->c_a ? a->in_a[(int64_t)irow * a->c_a + ci] : a->in_b[(int64_t)irow * a->c_b + (ci - a->c_a)];
          const float *w = W + ((int64_t)k * cin + ci) * cout;
          for (int c = 0; c < cout; ++c) acc[c] = fmaf(x, w[c], acc[c]);
        }
      }
      float ss = 0.f;
      for (int c = 0; c < cout; ++c) {
        float v = acc[c] * (a->scale ? a->scale[c] : 1.f) + (a->shift ? a->shift[c] : 0.f);
        if (a->residual) v += a->residual[(int64_t)orow * cout + c];
        if (a->relu) v = v > 0.f ? v : 0.f;
        acc[c] = v;
        ss += v * v;
      }
      if (a->l2norm) {
        const float nrm = sqrtf(ss);
        for (int c = 0; c < cout; ++c) acc[c] /= nrm;          /* no eps: model/resunet.py:230 */
      }
      memcpy(a->out + (int64_t)orow * cout, acc, (size_t)cout * sizeof(float));
    }
    free(acc);
  }
  free(W);
  return IMF_OK;
}
