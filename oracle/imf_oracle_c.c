/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the geometry half of IMFNet's descriptor
 * path, following MinkowskiEngine 0.5.4's CPU algorithm (sequential insert into a coordinate hash
 * map => first-occurrence row order; kernel map = one hash probe per (output row, kernel offset),
 * OpenMP over rows).  Used by tests as a second checker of oracle/imf_oracle.py and by bench.py's
 * `cpu_baseline` leg.  Never linked into or called by imfnet_amd/.
 *
 * Reference call sites restated (file:line under /root/reference):
 *   orc_voxelize    util/misc.py:82-87 (np.floor(xyz/voxel), ME.utils.sparse_quantize)
 *   orc_downsample  implicit coordinate_manager.stride() of the stride-2 convs, model/resunet.py:54-85
 *   orc_rulebook    kernel maps of ME.MinkowskiConvolution(Transpose), model/resunet.py:42-158
 * (imf_cpu_twins.c holds the twins with the C ABI's own signatures and layouts.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EMPTY 0xFFFFFFFFFFFFFFFFull
#define BITS 18
#define LIM (1 << (BITS - 1))

static inline uint64_t pack(int b, int x, int y, int z) {
  return ((uint64_t)(uint32_t)b << (3 * BITS)) | ((uint64_t)(x & 0x3FFFF) << (2 * BITS)) |
         ((uint64_t)(y & 0x3FFFF) << BITS) | (uint64_t)(z & 0x3FFFF);
}
static inline uint64_t mix(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}
typedef struct { uint64_t *keys; int32_t *vals; uint64_t mask; } table_t;

static int table_init(table_t *t, int64_t n) {
  uint64_t cap = 1024;
  while (cap < (uint64_t)(2 * n)) cap <<= 1;
  t->keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
  t->vals = (int32_t *)malloc(cap * sizeof(int32_t));
  if (!t->keys || !t->vals) return -1;
  memset(t->keys, 0xFF, cap * sizeof(uint64_t));
  t->mask = cap - 1;
  return 0;
}
static void table_free(table_t *t) { free(t->keys); free(t->vals); }
/* returns row of key; inserts with value `next` if absent (then *is_new = 1) */
static inline int32_t table_get_or_insert(table_t *t, uint64_t key, int32_t next, int *is_new) {
  uint64_t s = mix(key) & t->mask;
  for (;;) {
    if (t->keys[s] == key) { *is_new = 0; return t->vals[s]; }
    if (t->keys[s] == EMPTY) { t->keys[s] = key; t->vals[s] = next; *is_new = 1; return next; }
    s = (s + 1) & t->mask;
  }
}
static inline int32_t table_find(const table_t *t, uint64_t key) {
  uint64_t s = mix(key) & t->mask;
  for (;;) {
    if (t->keys[s] == key) return t->vals[s];
    if (t->keys[s] == EMPTY) return -1;
    s = (s + 1) & t->mask;
  }
}
static inline int floor_div(int a, int s) { return a >= 0 ? a / s : -((-a + s - 1) / s); }

/* coords4 [n,4] and inds [n] are caller buffers; returns M, or -1 (alloc) / -2 (range). */
int64_t orc_voxelize(const double *xyz, int64_t n, double voxel, int batch, int32_t *coords4,
                     int64_t *inds) {
  table_t t;
  if (table_init(&t, n)) return -1;
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    double fx = floor(xyz[3 * i] / voxel), fy = floor(xyz[3 * i + 1] / voxel),
           fz = floor(xyz[3 * i + 2] / voxel);
    if (!(fx >= -LIM && fx < LIM && fy >= -LIM && fy < LIM && fz >= -LIM && fz < LIM)) {
      table_free(&t);
      return -2;
    }
    int is_new;
    table_get_or_insert(&t, pack(batch, (int)fx, (int)fy, (int)fz), (int32_t)m, &is_new);
    if (is_new) {
      coords4[4 * m] = batch; coords4[4 * m + 1] = (int)fx; coords4[4 * m + 2] = (int)fy;
      coords4[4 * m + 3] = (int)fz;
      inds[m] = i;
      ++m;
    }
  }
  table_free(&t);
  return m;
}

int64_t orc_downsample(const int32_t *c4, int64_t n, int stride, int32_t *out4, int32_t *parent) {
  table_t t;
  if (table_init(&t, n)) return -1;
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    int b = c4[4 * i], x = floor_div(c4[4 * i + 1], stride) * stride,
        y = floor_div(c4[4 * i + 2], stride) * stride, z = floor_div(c4[4 * i + 3], stride) * stride;
    int is_new;
    int32_t r = table_get_or_insert(&t, pack(b, x, y, z), (int32_t)m, &is_new);
    if (is_new) {
      out4[4 * m] = b; out4[4 * m + 1] = x; out4[4 * m + 2] = y; out4[4 * m + 3] = z;
      ++m;
    }
    if (parent) parent[i] = r;
  }
  table_free(&t);
  return m;
}

/* nbr[o*K + k] = row of (out[o] + sign*off_k*ts) in `in`, or -1; k = (dx+r) + ks*(dy+r) + ks^2*(dz+r) */
int orc_rulebook(const int32_t *in4, int64_t n_in, const int32_t *out4, int64_t n_out, int ts,
                 int ksize, int sign, int32_t *nbr) {
  table_t t;
  if (table_init(&t, n_in)) return -1;
  for (int64_t i = 0; i < n_in; ++i) {
    int is_new;
    table_get_or_insert(&t, pack(in4[4 * i], in4[4 * i + 1], in4[4 * i + 2], in4[4 * i + 3]),
                        (int32_t)i, &is_new);
  }
  const int r = ksize / 2, K = ksize * ksize * ksize;
#pragma omp parallel for schedule(static)
  for (int64_t o = 0; o < n_out; ++o) {
    const int b = out4[4 * o], cx = out4[4 * o + 1], cy = out4[4 * o + 2], cz = out4[4 * o + 3];
    for (int k = 0; k < K; ++k) {
      int dx = k % ksize - r, dy = (k / ksize) % ksize - r, dz = k / (ksize * ksize) - r;
      int x = cx + sign * dx * ts, y = cy + sign * dy * ts, z = cz + sign * dz * ts;
      int32_t f = -1;
      if (x >= -LIM && x < LIM && y >= -LIM && y < LIM && z >= -LIM && z < LIM)
        f = table_find(&t, pack(b, x, y, z));
      nbr[o * K + k] = f;
    }
  }
  table_free(&t);
  return 0;
}

/* ---- CPU twin of imf_spconv_fwd (SURVEY 8b B3 "imf_cpu_*"; TEST INFRASTRUCTURE / reported CPU baseline only) ----------
 * out[o] = sum_k in[nbr[o][k]] @ W[k]  (model/resunet.py:168-226 convolutions; nbr row-major [n_out][kvol], -1 = no input
 * at that offset, the oracle's layout).  Output-stationary, OpenMP over blocks of output rows, the inner loop over the
 * output channels vectorises; k ascending then input channel = the GPU kernels' summation order. */
void imf_cpu_spconv_fwd(const float *in, int cin, const float *w, int kvol, int cout, const int32_t *nbr, int64_t n_out,
                        float *out) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t o = 0; o < n_out; ++o) {
    float acc[512];
    for (int c = 0; c < cout; ++c) acc[c] = 0.f;
    for (int k = 0; k < kvol; ++k) {
      const int32_t i = nbr ? nbr[o * kvol + k] : (int32_t)o;
      if (i < 0) continue;
      const float *x = in + (int64_t)i * cin;
      const float *wk = w + (int64_t)k * cin * cout;
      for (int ci = 0; ci < cin; ++ci) {
        const float a = x[ci];
        const float *wr = wk + (int64_t)ci * cout;
        for (int c = 0; c < cout; ++c) acc[c] += a * wr[c];
      }
    }
    float *dst = out + o * cout;
    for (int c = 0; c < cout; ++c) dst[c] = acc[c];
  }
}

#include <omp.h>
void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int orc_max_threads(void) { return omp_get_max_threads(); }
