/* imfnet_hip.h -- C ABI of libimfnet_hip.so: the MI355X (gfx950) sparse-3D-convolution engine
 * behind IMFNet's descriptor-generation path.
 *
 * The reference (XiaoshuiHuang/IMFNet) has no FFI of its own: its hot path calls the
 * MinkowskiEngine 0.5.4 Python API (requirements.txt:5).  Each entry point below states which
 * reference call site(s) it replaces (file:line under /root/reference).  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless marked [host]; the library never
 *     allocates or frees caller-visible memory and keeps no global mutable state (re-entrant per
 *     stream);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - return value: 0 on success, negative IMF_E* on error; imf_last_error() gives the message of
 *     the last failing call on the calling thread;
 *   - coordinates are int32 rows (b, x, y, z); b in [0,512), x/y/z in [-2^17, 2^17) in voxel units;
 *   - feature matrices are row-major float32 [rows, C];
 *   - all kernels are launched asynchronously; counts the caller needs on the host are written to
 *     device int32 words the caller reads back when it chooses to synchronise.
 */
#ifndef IMFNET_HIP_H
#define IMFNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMF_OK            0
#define IMF_EINVAL       -1   /* bad argument (null pointer, unsupported channel count, ...) */
#define IMF_ELAUNCH      -2   /* HIP launch / runtime error */
#define IMF_EUNSUPPORTED -3

#define IMF_TILE_ROWS    64   /* output rows per rulebook tile (= 4 wavefronts x 16-row MFMA blocks) */
#define IMF_MAX_KVOL     125  /* largest kernel volume (5x5x5, config_3dmatch.py:68) */
#define IMF_MAX_BATCH    8    /* items of a batched pyramid / forward */
#define IMF_MASK_WORDS   4    /* 128-bit active-offset mask per tile */

int imf_version(void);
const char *imf_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Voxel hash.  Open-addressing table of `capacity` 16-byte slots (power of two):
 *   key   uint64   packed (b,x,y,z) = b:9 | x:18 | y:18 | z:18 (two's complement fields), 0xFFFF... = empty
 *   val   int32    row index of that voxel in its level
 * Key and value share a slot so that a probe that hits is ONE 16-byte load (the rulebook builds are bound by
 * the number of random loads they issue: with separate key / value arrays a hit cost a second random line).
 * Probe sequence (csrc/common.h hash_slot / hash_step; a client that probes a table itself must walk the same one):
 * the level's coordinates are multiples of 2^shift (shift = log2 of the level's tensor stride); the 4 x 4 x 4 block of
 * voxels a key belongs to -- bits shift, shift + 1 of each coordinate field cleared -- is hashed (murmur3 finaliser) to a
 * 64-slot window, the two cleared bits of x, y, z pick the slot inside it (z fastest); a collision moves on by the
 * key-dependent ODD stride (hash(key) >> 3) | 1 (double hashing), modulo the capacity.  The library derives `shift` from the
 * stride arguments of its entry points: imf_rulebook_conv probes `in_table` with shift = log2(ts_in);
 * imf_rulebook_transpose probes the COARSE table with shift = log2(ts_fine) + 1, i.e. it serves coarse stride = 2 x fine
 * stride only (the three transposed convolutions of model/resunet.py:101-134) -- tables of other stride ratios miss.
 * Capacity to use for n keys: imf_hash_capacity(n).
 * ---------------------------------------------------------------------------------------------- */
typedef struct imf_slot {
  uint64_t key;
  int32_t val;
  int32_t pad;
} imf_slot;
int64_t imf_hash_capacity(int64_t n);

/* Bytes of scratch imf_voxelize / imf_downsample need for n input rows. */
size_t imf_unique_workspace_bytes(int64_t n);

/* Replaces: util/misc.py:82-87  coords = np.floor(xyz / voxel_size);
 *           ME.utils.sparse_quantize(coords, return_index=True); ME.utils.batched_coordinates;
 *           and the coordinate-manager insert of ME.SparseTensor(...) at util/misc.py:95.
 * xyz: [n,3] float64 (xyz_is_f64=1) or float32 (=0; widened to double before the division, which is
 *      what Open3D does to the float32 PLY at scripts/generate_desc.py:83,102).
 * Division is IEEE float64, floor() exact => voxel indices are bit-identical to the reference.
 * Output rows are in FIRST-OCCURRENCE order (ascending first point index):
 *   coords     int32[n,4]  (only the first *m_out rows are written)
 *   first_idx  int32[n]    index of the first point falling in each voxel (`inds`)
 *   m_out      int32[1]    number of voxels M
 *   table      hash of the M voxels (capacity = imf_hash_capacity(n) slots), val = row
 *   err_out    int32[1]    set non-zero if a coordinate was out of range (caller zeroes it)
 */
int imf_voxelize(const void *xyz, int xyz_is_f64, int64_t n, double voxel_size, int batch_index,
                 int32_t *coords, int32_t *first_idx, int32_t *m_out,
                 imf_slot *table, int64_t capacity,
                 void *workspace, int32_t *err_out, void *stream);

/* Replaces: the implicit coordinate_manager.stride() inside every stride-2
 *           ME.MinkowskiConvolution (model/resunet.py:54-85): coarse = floor(c / out_stride) *
 *           out_stride, unique rows, first-occurrence order.
 * n_in_dev: device int32 holding the actual number of input rows (<= n_in_max, which sizes the
 *           grid, the workspace and the table). */
int imf_downsample(const int32_t *coords_in, const int32_t *n_in_dev, int64_t n_in_max,
                   int out_stride,
                   int32_t *coords_out, int32_t *m_out,
                   imf_slot *table, int64_t capacity,
                   void *workspace, void *stream);

/* Replaces: `return_coords = xyz[inds]` (util/misc.py:92, returned at :104): the representative point of every voxel,
 * gathered on the device so that only M rows travel back to the host.  xyz as given to imf_voxelize / imf_pyramid_build
 * (float32 is widened, as Open3D's float64 point array is); first_idx from the same call; m_dev: optional device row
 * count (capacity mode: rows beyond it are not written), m_cap: rows (upper bound).  out: float64 [m_cap, 3]. */
int imf_gather_points(const void *xyz, int xyz_is_f64, const int32_t *first_idx, const int32_t *m_dev, int64_t m_cap,
                      double *out, void *stream);

/* One-call geometry: imf_voxelize followed by (n_levels - 1) imf_downsample (tensor strides 2, 4,
 * ...), all tables and scratch carved out of ONE caller-provided arena, row counts written to
 * meta[level][0] (meta[level][1] = error flag), followed by the bounding box of level 0:
 * meta[2*n_levels + 0..3] = min (b,x,y,z), [4..7] = max -- meta is int32[2*n_levels + 8], initialised
 * by the call.  18 kernel launches, no host synchronisation.  levels_out [host] receives the device
 * addresses inside the arena.
 * Replaces: util/misc.py:82-95 and the implicit cm.stride() chain of model/resunet.py:54-85. */
typedef struct imf_level {
  int32_t *coords;       /* [cap_rows, 4] rows (b,x,y,z), first-occurrence order                 */
  imf_slot *table;       /* hash table [capacity]: voxel key -> row                               */
  int64_t capacity;
  int32_t *first_idx;    /* level 0: index of each voxel's first point; NULL on coarser levels    */
  int64_t cap_rows;      /* rows allocated (upper bound = n points)                               */
  int32_t tensor_stride;
} imf_level;

size_t imf_pyramid_arena_bytes(int64_t n_points, int n_levels);
int imf_pyramid_build(const void *xyz, int xyz_is_f64, int64_t n, double voxel_size, int batch_index,
                      int n_levels, void *arena, size_t arena_bytes, int32_t *meta,
                      imf_level *levels_out /* [host] */, void *stream);
/* The same for a batch of fragments (the reference's ME.utils.batched_coordinates + one SparseTensor,
 * model/resunet.py:241-250): xyz holds the items' points back to back, item b = points
 * [item_starts[b], item_starts[b+1]) ([host], item_starts[0] == 0), batch index b.  Rows come out grouped
 * by item in first-occurrence order, exactly as concatenating the single-fragment results.  meta:
 * int32[2*n_levels + 8 + IMF_MAX_BATCH*n_levels]; the last block holds the first row of item b at level
 * l at [IMF_MAX_BATCH*l + b]. */
int imf_pyramid_build_batched(const void *xyz, int xyz_is_f64, int64_t n, double voxel_size,
                              const int64_t *item_starts /* [host] */, int n_items, int n_levels, void *arena,
                              size_t arena_bytes, int32_t *meta, imf_level *levels_out, void *stream);

/* Capacity mode of the pyramid (for launch sequences captured once and replayed): the point count, the number
 * of items and the items' first points are read from the device (dyn int32[16]: [0] points <= n_points_cap, [1]
 * items, [2 + b] first point of item b); level l holds at most row_caps[l] rows ([host], non-increasing, <=
 * n_points_cap) and its table is sized for the rows that can be inserted into it, so an over-full level cannot
 * fill a table: its count is clamped, the surplus voxels map to "no row" and bit 1 of meta[2l + 1] is raised.
 * meta always has the batched layout (int32[2*n_levels + 8 + IMF_MAX_BATCH*n_levels]). */
size_t imf_pyramid_arena_bytes_caps(int64_t n_points_cap, int n_levels, const int64_t *row_caps /* [host] */);
int imf_pyramid_build_dyn(const void *xyz, int xyz_is_f64, const int32_t *dyn, int64_t n_points_cap,
                          const int64_t *row_caps /* [host] */, double voxel_size, int n_levels, void *arena,
                          size_t arena_bytes, int32_t *meta, imf_level *levels_out /* [host] */, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Rulebook (MinkowskiEngine "kernel map"), tiled for the MFMA kernel:
 *   tile_rows int32[n_slots]             output row of every slot, -1 = padding
 *                                        (n_slots = n_tiles * IMF_TILE_ROWS)
 *   nbr       int32[kvol * n_slots]      nbr[k * n_slots + slot] = input row feeding that output
 *                                        row through kernel offset k, -1 = none
 *   tile_mask uint32[n_tiles * 4]        bit k set <=> some row of the tile has an input at offset k
 * Kernel offset order: k = (dx+r) + K1*(dy+r) + K1^2*(dz+r), x fastest (ME kernel_region.hpp).
 * ---------------------------------------------------------------------------------------------- */

/* Identity slot order (slot == output row), padded to a whole tile. */
int64_t imf_rulebook_slots(int64_t n_out);

/* Replaces: the kernel-map generation implicit in ME.MinkowskiConvolution(kernel_size=ksize,
 *           stride in {1,2}) -- model/resunet.py:42-88,140-158, model/residual_block.py:23-33.
 *   in  = out + off_k * ts_in    (ts_in = tensor stride of the INPUT level)
 * in_table/in_capacity: hash of the input level.  out_coords: the n_out output rows. */
int imf_rulebook_conv(const imf_slot *in_table, int64_t in_capacity,
                      const int32_t *out_coords, int64_t n_out, int ts_in, int ksize,
                      int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask, void *stream);

/* Occupancy-sorted twin of a kernel map (csrc/rulebook_sort.hip): the same rows and inputs, the SLOTS re-ordered so that the
 * rows of a tile / of a 16-row block have similar neighbour-occupancy patterns -- stable sort inside windows of 16 384
 * consecutive slots by key = gray^-1(r), r = the occupancy bits (nbr_in[k][slot] >= 0) of the 12 edge offsets of the 3x3x3
 * kernel in bits 17 .. 6 (k = 1, 3, .. 25 without 13, in that order from bit 17 down) and of the 6 face offsets (4, 10, 12,
 * 14, 16, 22) in bits 5 .. 0; slots >= the row count stay last in their window (kvol != 27: r = the bits of the first 18
 * offsets).  nbr_in: a map in identity slot order (imf_rulebook_conv(_dyn), kvol
 * <= 27); outputs: tile_rows[slot] = the row now in that slot (-1: padding), nbr_out[k][slot] = nbr_in[k][row], tile_mask
 * recomputed.  A 64-row tile then walks ~78 % of the 27 offsets instead of ~100 % (stride-1 level of a 3DMatch fragment; 1.47
 * issued multiply-adds per useful one instead of 1.91), with no change to imf_spconv_fwd.  n_out_dev: optional
 * device-side row count (capacity mode; n_out is then the capacity).  Three launches: keys, the window sort, the gather.  Replaces:
 * nothing in the reference (MinkowskiEngine's kernel maps have no tile structure); used by the executors for the decoder's
 * stride-1 blocks (model/resunet.py:136-146, imf_resunet_sorted_maps).
 * workspace: imf_rulebook_sorted_workspace_bytes(n_slots) bytes of device memory (the permutation). */
size_t imf_rulebook_sorted_workspace_bytes(int64_t n_slots);
int imf_rulebook_sort_by_occupancy(const int32_t *nbr_in, int kvol, int64_t n_slots, int64_t n_out, const int32_t *n_out_dev,
                                   int32_t *tile_rows, int32_t *nbr_out, uint32_t *tile_mask, void *workspace,
                                   size_t workspace_bytes, void *stream);

/* Capacity mode: tables sized for n_out_cap rows, the actual count read from *n_out_dev; tiles without rows keep
 * mask 0 and their neighbour slices are left unwritten (imf_spconv_fwd never looks at them). */
int imf_rulebook_conv_dyn(const imf_slot *in_table, int64_t in_capacity,
                          const int32_t *out_coords, int64_t n_out_cap, const int32_t *n_out_dev, int ts_in,
                          int ksize, int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask, void *stream);

/* Upper bound of slots for the transposed rulebook (rows grouped into 8 parity classes, every
 * class padded to whole tiles). */
int64_t imf_rulebook_transpose_slots(int64_t n_fine);

/* Replaces: the kernel map of ME.MinkowskiConvolutionTranspose(kernel_size=3, stride=2)
 *           (model/resunet.py:101-134): the forward fine->coarse map with in/out swapped, same k:
 *           out[f] += in[c] @ W[k]  for  f = c + off_k * ts_fine.
 * Output rows (fine voxels) are grouped by the parity of (coord / ts_fine) so that a tile shares
 * its (at most 8) active offsets.  counters: int32[16] scratch. */
int imf_rulebook_transpose(const imf_slot *coarse_table, int64_t coarse_capacity,
                           const int32_t *fine_coords, int64_t n_fine, int ts_fine, int ksize,
                           int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask,
                           int64_t n_slots, int32_t *counters, void *stream);

int imf_rulebook_transpose_dyn(const imf_slot *coarse_table, int64_t coarse_capacity,
                               const int32_t *fine_coords, int64_t n_fine_cap, const int32_t *n_fine_dev, int ts_fine,
                               int ksize, int32_t *tile_rows, int32_t *nbr, uint32_t *tile_mask, int64_t n_slots,
                               int32_t *counters, void *stream);   /* n_slots = imf_rulebook_transpose_slots(n_fine_cap) */

/* ------------------------------------------------------------------------------------------------
 * Sparse convolution.
 * ---------------------------------------------------------------------------------------------- */

/* Number of floats of the packed ("fragment-major") weight image for a [kvol, cin, cout] kernel. */
int64_t imf_packed_weight_floats(int kvol, int cin, int cout);

/* Re-lays a MinkowskiEngine kernel tensor W[kvol][cin][cout] (state_dict key '*.kernel';
 * [cin][cout] when kvol == 1) into the order the MFMA kernel streams it.  Done once per model
 * load.  cin % 32 == 0, cout % 32 == 0. */
int imf_pack_weights(const float *w, int kvol, int cin, int cout, float *packed, void *stream);
/* The same weights as hi/lo f16 B fragments for imf_conv_args.variant == 6 (split-f16 MFMA: every
 * fp32 operand x = f16(x) + f16(x - f16(x)); products hi*hi + hi*lo + lo*hi accumulate in fp32).  The kernel
 * is stored scaled by the power of two that puts max|w| in [2^13, 2^14) (keeps the lo halves normal f16
 * numbers; exact) and the convolution multiplies 2^-s back into its fp32 accumulators.  The image is
 * imf_packed_weight_floats_split16 floats: the fp32 image's size + a 64-float trailer holding 2^-s. */
int64_t imf_packed_weight_floats_split16(int kvol, int cin, int cout);
int imf_pack_weights_split16(const float *w, int kvol, int cin, int cout, float *packed, void *stream);
/* The same weights as THREE bf16 parts per value for imf_conv_args.variant == 3 ("bf16x3", round 5): w = p0 + p1 + p2
 * EXACTLY (p0 = bf16(w), p1 = bf16(w - p0), p2 = bf16(w - p0 - p1), round-to-nearest-even: three 8-bit significands carry
 * fp32's 24 bits, and bf16 has fp32's exponent range -- no pre-scaling, no range restriction).  The convolution splits its
 * fp32 input rows the same way in registers and forms a0 w2 + a1 w1 + a2 w0 + a0 w1 + a1 w0 + a0 w0 on the bf16 matrix
 * pipe with fp32 accumulation: every product term down to 2^-16 relative; the three dropped terms are together
 * <= 2^-26 |a| |w|, a quarter of an fp32 ulp of the product.  1.5 x the fp32 image's size.  Same reference as
 * imf_pack_weights (the '*.kernel' tensors of model/resunet.py:42-158, model/residual_block.py:23-33). */
int64_t imf_packed_weight_floats_bf16x3(int kvol, int cin, int cout);
int imf_pack_weights_bf16x3(const float *w, int kvol, int cin, int cout, float *packed, void *stream);

typedef struct imf_conv_args {
  const float *in_a;      /* [n_in, c_a]                                                      */
  const float *in_b;      /* [n_in, c_b] or NULL: second source of a fused ME.cat(a, b)       */
  int32_t c_a, c_b;       /* cin = c_a + c_b; c_a % 32 == 0, c_b % 32 == 0                    */
  const float *w_packed;  /* imf_pack_weights image                                           */
  int32_t kvol, cout;
  const int32_t *tile_rows;         /* NULL = identity (slot == row)                                  */
  const int32_t *nbr;               /* NULL allowed when kvol == 1: input row == output row           */
  const uint32_t *tile_mask;        /* NULL allowed when kvol == 1: every tile active at offset 0     */
  int64_t n_slots, n_out;
  const float *scale;     /* [cout] or NULL   y = acc * scale + shift  (folded eval BatchNorm /   */
  const float *shift;     /* [cout] or NULL                             bias)                     */
  const float *residual;  /* [n_out, cout] or NULL   y += residual                                */
  int32_t relu;           /* y = max(y, 0)                                                        */
  int32_t l2norm;         /* y /= ||y||_2 over the row (requires cout <= 64)                      */
  float *out;             /* [n_out, cout]                                                        */
  int32_t split_k;        /* 0 = choose automatically; >= 1 = number of kernel-offset partitions  */
  int32_t variant;        /* 0 = fp32 MFMA (v_mfma_f32_16x16x4_f32: the reference's arithmetic), since round 5 on the LDS-DMA
                                 kernels k_spconv_g / k_spconv_w (AR = kArF32) wherever their tables cover the shape
                                 (kvol <= 27, kvol * cin / 32 < 224, <= 1024 channels per source, inputs < 2 GiB), else --
                                 or with kernel_tag bit 1 / `tickets` -- on round 1's register-staged k_spconv_mfma;
                             1 = the same arithmetic without any pipeline (simple reference kernel; any kvol);
                             2, 4, 5 = retired round-1 experiments (their measurements: DESIGN.md 4), IMF_EINVAL;
                             3 = "bf16x3": fp32 operands carried exactly by three bf16 parts each, six
                                 v_mfma_f32_16x16x32_bf16 per 32 channels, fp32 accumulation (w_packed from
                                 imf_pack_weights_bf16x3; fp32 rows in, fp32 rows out, no range restriction; kvol <= 27;
                                 in_a / in_b smaller than 2 GiB each).  k_spconv_g / k_spconv_w with AR = kArBf16x3;
                             6 = fp32-class arithmetic on the f16 matrix pipe with split operands, both operands staged
                                 global -> LDS by DMA (k_spconv_g; see kernel_tag for the register-staged twin)
                                 (w_packed from imf_pack_weights_split16; kvol <= 27; |input| < 65504;
                                 in_a / in_b smaller than 2 GiB each: raw-buffer addressing) */
  void *workspace;        /* split-K partial sums (NULL allowed iff split_k resolves to 1) */
  size_t workspace_bytes; /* >= imf_spconv_workspace_bytes(n_slots, cout, split)                  */
  int32_t *tickets;       /* optional: int32[n_tiles * n_slabs] arrival counters, ZERO on entry (left zero on
                             exit): with split_k > 1 the last partition to finish a tile reduces it inside
                             the same launch (agent-scope release/acquire) -- no second kernel.  Variant 0; with
                             variant 6 in diagnostic builds only (measured slower than the second launch) */
  void *ev_begin, *ev_end; /* optional hipEvent_t pair recorded on `stream` immediately around the
                              main MFMA kernel (not the split-K reduce): live roofline timing     */
  /* Capacity mode (variant 6; variant 0 with split_k == 1), for launch sequences captured once and replayed on fragments of different
   * size: n_out / n_slots (and the rulebook) are sized for a CAPACITY and the actual row count is read from
   * device memory; tiles beyond it exit at once.  dyn_split_kvol != 0: the number of kernel-offset partitions
   * is imf_spconv_auto_split(actual slots, cout, dyn_split_kvol) evaluated on the device -- split_k then only
   * sizes the grid and the workspace: it must cover the largest value expected (imf_spconv_max_split covers any
   * row count; a smaller cover raises bit 4 of *dyn_err when exceeded) -- so the result is bit-identical to an
   * exact-size launch.  slots_extra: slots the rulebook lays out beyond roundup64(rows) (512 for
   * imf_rulebook_transpose's parity classes, else 0). */
  const int32_t *n_out_dev;
  int32_t dyn_split_kvol, slots_extra;
  int32_t kernel_tag;     /* variants 6, 3 and 0.  bit 0: profiling label -- the identical kernel under a second symbol
                             (k_spconv_g<.., 1>: the image branch's dense convolutions).  bit 1: the register-staged twin
                             k_spconv_h3 (csrc/spconv_h3.hip; same sums bit for bit) -- DIAGNOSTIC builds only (make h3 /
                             stamps): the product library answers IMF_EUNSUPPORTED, as it does for variant 6 + `tickets`;
                             with variant 0: round 1's register-staged fp32 kernel k_spconv_mfma (always built).
                             bit 2 (4) / bit 3 (8): the wave-split kernel k_spconv_w (csrc/spconv_w.hip) with 8 / 4
                             wavefronts per workgroup -- for levels of a few hundred 64-row tiles or fewer: one workgroup
                             owns a (tile, 64-column slab) for all kernel offsets, its wavefronts split the tile's
                             (offset, 32-channel chunk) list into contiguous ranges and combine their partial tiles through
                             LDS in wavefront order, epilogue in the same launch.  Needs kvol > 1, cout % 64 == 0,
                             split_k <= 1.  The wavefront count is part of the summation order (deterministic; a tile's
                             sums do not depend on the row count, so exact and capacity mode agree bit for bit).
                             bit 6 (64), with bit 3 and variant 3 only: HALF-TILE workgroups -- each 4-wavefront workgroup
                             owns 32 of a tile's 64 rows (same offset list, same per-row sums as bit 3 alone): twice the
                             workgroups for levels that leave CUs idle (one fragment per forward).
                             bit 7 (128), with bit 2 (any variant) or bit 3 (variant 3): 48-ROW UNITS -- each workgroup owns
                             slots 48 u .. 48 u + 47 whatever the tile boundaries and walks the union of the offset lists
                             of the tiles it touches (8 wavefronts: a pair's stride-8 level, 184 workgroups instead of 136;
                             4 wavefronts: the 64 -> 64 layers of level 0, three workgroups per CU).
                             bit 8 (256), with bit 3 and variant 3: the build for one more wavefront per SIMD -- whole tiles
                             for three (168 VGPRs), with bit 6 half tiles for four (127 VGPRs, 25 KiB of LDS; also with bit 2: half tiles of
                             8 wavefronts, two workgroups per CU); the same
                             sums as without the bit, bit for bit */
  int32_t *dyn_err;       /* optional device flag word (any mode): IMF_FLAG_SPLIT_COVER when the rule asks for more
                             partitions than split_k covers; IMF_FLAG_RANGE when an OUTPUT value is NaN or |y| >= 65504,
                             i.e. cannot be an operand of a following variant-6 convolution */
  int32_t geglu;          /* variant 6, kvol == 1, cout % 64 == 0, unsplit, no scale / residual / relu / l2norm: GEGLU
                             epilogue of the fusion block's feed-forward (model/attention_fusion.py:20-23,56-63).  The
                             weight image's columns are arranged per 64-column slab y as [32 values | 32 gates] of hidden
                             units 32 y .. 32 y + 31 (shift likewise); out is [n_out, cout / 2]:
                             out[r][32 y + c] = (acc_v + shift_v) * gelu(acc_g + shift_g), exact-erf GELU */
  int32_t operand_format; /* variant 6 (k_spconv_g / k_spconv_w, unsplit launches): which of the row-major [rows, channels]
                             buffers are SPLIT-F16 OPERAND IMAGES instead of fp32 -- IMF_FMT_A_SPLIT: in_a and in_b;
                             IMF_FMT_RES_SPLIT: residual; IMF_FMT_OUT_SPLIT: out is written as one (not with l2norm).
                             An operand image has the size and row stride of the fp32 buffer: per row and 32-channel chunk
                             128 bytes = [4 hi pieces | 4 lo pieces], piece j = the 8 halves of channels {4j..4j+3,
                             16+4j..16+4j+3}, hi = f16(x), lo = f16(x - hi) -- what the kernels' main loops derive from an
                             fp32 row themselves, 27 times per row for a 3x3x3 map; a producer that writes the image spares
                             its consumers the conversions (same products bit for bit).  A residual read takes hi + lo,
                             i.e. the value to 22 significant bits (relative 2^-22). */
} imf_conv_args;
#define IMF_FMT_A_SPLIT   1
#define IMF_FMT_RES_SPLIT 2
#define IMF_FMT_OUT_SPLIT 4

/* Flag bits the kernels OR into a caller-provided device word (imf_conv_args.dyn_err, imf_resunet_io.flags, meta[1]
 * of the capacity mode).  A flagged result must not be used: redo the fragment (larger capacities / exact mode for
 * 2..16; the fp32-MFMA variant 0 for IMF_FLAG_RANGE). */
#define IMF_FLAG_COORD_RANGE  1   /* a point fell outside [-2^17, 2^17) voxels or was NaN */
#define IMF_FLAG_CAPACITY     2   /* a pyramid level exceeded its row capacity */
#define IMF_FLAG_BITGRID      4   /* level-0 bounding box larger than the conv1 bit grid */
#define IMF_FLAG_EMPTY_ITEM   8   /* a batch item without voxels */
#define IMF_FLAG_SPLIT_COVER 16   /* fewer rows than a quarter of the capacity */
#define IMF_FLAG_RANGE       32   /* an activation left the f16 range of the split-f16 convolution operands */

/* Kernel-offset partitions imf_spconv_fwd will use for this shape when args.split_k == 0: small
 * levels (few tiles) are latency-bound, so their offsets are spread over several workgroups whose
 * partial sums are combined -- in a fixed order, hence still bit-reproducible -- by a second
 * kernel that also applies the epilogue. */
int imf_spconv_auto_split(int64_t n_slots, int cout, int kvol);
int imf_spconv_max_split(int cout, int kvol);   /* largest value the rule returns for any row count */
/* Tuning aid: resident workgroups per CU reported by the runtime for kernel `variant`. */
int imf_spconv_occupancy(int variant, int co_blk, int j);
size_t imf_spconv_workspace_bytes(int64_t n_slots, int cout, int split);

/* Replaces: ME.MinkowskiConvolution / ME.MinkowskiConvolutionTranspose forward
 *           (model/resunet.py:168-226, model/residual_block.py:40-48) with the following
 *           ME.MinkowskiBatchNorm (eval), MEF.relu, residual `out += residual`
 *           (residual_block.py:50), ME.cat (resunet.py:197,208,219), `final` bias and the
 *           L2 normalisation (resunet.py:228-233) fused as prologue / epilogue.
 * out[o] = epilogue( sum_k in[nbr[k][o]] @ W[k] ).  args->variant picks the arithmetic: 6 (what the model layers
 * use) = fp32 operands split into f16 hi + lo halves, three v_mfma_f32_16x16x32_f16 per 32 channels with fp32
 * accumulation (fp32-class accuracy; inputs must stay below 65 504, see dyn_err / IMF_FLAG_RANGE); 0 = fp32 MFMA
 * (v_mfma_f32_16x16x4_f32, an exact f32 FMA chain).  Either way the sum order per output element is fixed (k
 * ascending, then input channel) => deterministic. */
int imf_spconv_fwd(const imf_conv_args *args /* [host] */, void *stream);

/* Pointwise head: two chained 1x1x1 convolutions with their epilogues as ONE launch, the [n, 64] intermediate never
 * leaving the registers / LDS of the workgroup that produced it.
 * Replaces: conv1_tr (ME.MinkowskiConvolutionTranspose kernel_size=1 over ME.cat(out_s1_tr, out_s1)) + norm1_tr +
 *           MEF.relu + final (ME.MinkowskiConvolution kernel_size=1, has_bias) + the L2 normalisation
 *           (model/resunet.py:136-158 ctor, 219-233 forward).
 *   hid = relu?((cat(in_a, in_b) @ W1) * scale1 + shift1)      [n, 64]
 *   out = l2norm?((hid @ W2) * scale2 + shift2)                  [n, 32]
 * Same arithmetic and summation order as two imf_spconv_fwd calls with variant 6 (bit-identical result).  Weight
 * images from imf_pack_weights_split16(kvol = 1).  c_a + c_b in {64, 96, 128}; c_mid == 64, c_out == 32. */
typedef struct imf_head_args {
  const float *in_a;       /* [n, c_a] */
  const float *in_b;       /* [n, c_b] or NULL */
  int32_t c_a, c_b;
  const float *w1_packed;  /* split16 image of W1 [1][c_a + c_b][c_mid] */
  const float *scale1, *shift1;   /* [c_mid] or NULL (folded BatchNorm) */
  int32_t relu1, c_mid;
  const float *w2_packed;  /* split16 image of W2 [1][c_mid][c_out] */
  const float *scale2, *shift2;   /* [c_out] or NULL (shift2 = bias) */
  int32_t l2norm, c_out;
  int64_t n;               /* rows (capacity when n_dev is given) */
  const int32_t *n_dev;    /* optional: actual row count on the device (capacity mode) */
  float *out;              /* [n, c_out] */
  int32_t *flags;          /* optional device word: IMF_FLAG_RANGE when a hidden value cannot be a split-f16 operand */
  void *ev_begin, *ev_end; /* optional hipEvent_t pair recorded around the kernel */
  int32_t a_split;         /* in_a / in_b are split-f16 operand images (imf_conv_args.operand_format) */
  int32_t variant;         /* 0 or 6: split16 weight images (above).  3: imf_pack_weights_bf16x3 images, fp32 rows, c_a + c_b
                              <= 96, no range flag -- bit-identical to two imf_spconv_fwd calls with variant 3 */
} imf_head_args;
int imf_pointwise_head(const imf_head_args *args /* [host] */, void *stream);

/* First-layer convolution for a small number of input channels (cin <= 4, e.g. the all-ones
 * occupancy feature of util/misc.py:76-79 through conv1 k=5, model/resunet.py:42-49).
 * w is the UNPACKED ME kernel [kvol][cin][cout], cout in {32, 64}.  Identity slot order. */
int imf_spconv_small_cin(const float *in, int cin, const float *w, int kvol, int cout,
                         const int32_t *nbr, int64_t n_slots, int64_t n_out,
                         const float *scale, const float *shift, int relu,
                         float *out, void *stream);

/* First-layer convolution fused with its own kernel map: probes the input level's voxel hash
 * directly (one wavefront per output voxel) instead of materialising the 125-column neighbour
 * table.  in == NULL means the all-ones occupancy feature of util/misc.py:76-79.
 * Replaces: conv1 + norm1 of model/resunet.py:42-49,168-169 (kernel map included). */
int imf_conv_first_fused(const imf_slot *table, int64_t capacity,
                         const int32_t *coords, int64_t n, int ts, int ksize, const float *in, int cin,
                         const float *w /* [kvol][cin][cout], unpacked */, int cout,
                         const float *scale, const float *shift, int relu, float *out, void *stream);

/* First-layer convolution for the all-ones occupancy feature (util/misc.py:76-79) on a dense
 * occupancy BIT GRID over the level's bounding box (bbox [host] = min b,x,y,z, max b,x,y,z as
 * imf_pyramid_build reports it): 25 five-bit windows per voxel instead of 125 hash probes, then
 * out = occupancy . W on fp32 MFMA.  grid: caller scratch of >= imf_bitgrid_words(bbox, ksize)
 * uint32 words (0 = box too large: use imf_conv_first_fused).  Tensor stride 1.
 * Replaces: conv1 + norm1 of model/resunet.py:42-49,168-169 (kernel map included). */
size_t imf_bitgrid_words(const int32_t *bbox /* [host] */, int ksize);
int imf_conv_first_bitgrid(const int32_t *coords, int64_t n, const int32_t *bbox /* [host] */, int ksize,
                           uint32_t *grid, size_t grid_words, const float *w /* [kvol][1][cout] */,
                           int cout, const float *scale, const float *shift, int relu, float *out,
                           void *stream);

/* Capacity mode: row count and bounding box (8 ints) read from the device; a box that needs more than grid_words
 * raises bit 2 (value 4) of *err and the launch does nothing. */
/* The same with a flag word for IMF_FLAG_RANGE on the outputs (flags may be NULL). */
int imf_conv_first_bitgrid_flags(const int32_t *coords, int64_t n, const int32_t *bbox /* [host] */, int ksize,
                                 uint32_t *grid, size_t grid_words, const float *w, int cout, const float *scale,
                                 const float *shift, int relu, float *out, int32_t *flags, void *stream);
int imf_conv_first_bitgrid_dyn(const int32_t *coords, int64_t n_cap, const int32_t *n_dev, const int32_t *bbox_dev,
                               int32_t *err, int ksize, uint32_t *grid, size_t grid_words, const float *w, int cout,
                               const float *scale, const float *shift, int relu, float *out, void *stream);

/* Bottleneck fusion block (one image, one attention head, depth 0) as ONE kernel.
 * Replaces: ResUNet2.transformer + AttentionFusion.forward (model/resunet.py:237-273,
 *           model/attention_fusion.py:65-95,132-154) for the N stride-8 point rows x [n,256]:
 *   x = to_out(softmax(to_q(LN(x)) K^T * scale) V) + x ;  x = W2(GEGLU(W1 LN(x))) + x
 * The attention matrices are given in the fragment-major layout of imf_pack_weights(kvol = 1) applied to the
 * TRANSPOSED torch Linear weight ([in, out]): wq_p [256->128], wo_p [128->256]; kt_packed = pack(K^T [128,
 * tokens_padded]), v_packed = pack(V [tokens_padded, 128]) with K, V = chunks of to_kv(LN(image tokens)), zero-padded
 * to tokens_padded (multiple of 64, <= 320).  The feed-forward runs on imf_spconv_fwd's split-f16 kernels (a Linear
 * layer is a 1x1x1 convolution over the rows), so its matrices are imf_pack_weights_split16 images:
 *   w1_p  [1][256][2048] = W1^T with the output columns re-ordered for imf_conv_args.geglu: packed column
 *         64 j + c      (c < 32) = value column 32 j + c        (torch row 32 j + c of ff[0].weight)
 *         64 j + 32 + c          = gate column 1024 + 32 j + c  (the second chunk of GEGLU's `chunk(2, dim=-1)`)
 *   b1    [2048] in the same packed order;  w2_p [1][1024][256] = W2^T;  b2 [256].
 * workspace: imf_fusion_workspace_bytes(n) bytes, 16-byte aligned (y, LN2(y) and the GEGLU hidden, 6 KB per row). */
typedef struct imf_fusion_weights {
  const float *ln1_g, *ln1_b, *wq_p, *wo_p, *bo, *ln2_g, *ln2_b, *w1_p, *b1, *w2_p, *b2;
  /* the same two feed-forward matrices as fp32 imf_pack_weights images (same column order), or NULL: what the block
   * multiplies when the network runs on variant 0 -- the fp32 recompute of a fragment whose activations left the f16
   * range must not pass through f16 operands anywhere (the attention half is fp32 MFMA in either case) */
  const float *w1_f32, *w2_f32;
} imf_fusion_weights;
/* Workspace of the block's three launches (attention half -> GEGLU GEMM -> output GEMM): 6 KB per row. */
size_t imf_fusion_workspace_bytes(int64_t n);
int imf_fusion_attention(const float *x, int64_t n, const float *kt_packed, const float *v_packed,
                         int n_tokens, int tokens_padded, const imf_fusion_weights *w /* [host] */,
                         float scale, float *out, void *workspace, size_t workspace_bytes, void *stream);
/* Batched form (one image per batch item, model/resunet.py:241-250): rows [item_row0[b], +item_rows[b]) of x
 * attend to the tokens of image b (kt_packed[b], v_packed[b]); arrays are [host], n_items <= IMF_MAX_BATCH.
 * Workspace as imf_fusion_workspace_bytes(total rows). */
int imf_fusion_attention_batched(const float *x, int n_items, const int64_t *item_row0, const int64_t *item_rows,
                                 const float *const *kt_packed, const float *const *v_packed, int n_tokens,
                                 int tokens_padded, const imf_fusion_weights *w /* [host] */, float scale, float *out,
                                 void *workspace, size_t workspace_bytes, void *stream);
/* The same with the arithmetic of the feed-forward chosen (variant 6: split-f16 images w1_p / w2_p; variant 3: w1_p / w2_p are
 * imf_pack_weights_bf16x3 images, fp32 rows between the two GEMMs; variant 0: fp32 MFMA on
 * w1_f32 / w2_f32) and a flag word (device int32, may be NULL; caller zeroes) that receives IMF_FLAG_RANGE when a value
 * that feeds an f16 operand -- LN2's output, the GEGLU hidden, the block's output -- is NaN or >= 65504 in magnitude. */
int imf_fusion_attention_batched_v(const float *x, int n_items, const int64_t *item_row0, const int64_t *item_rows,
                                   const float *const *kt_packed, const float *const *v_packed, int n_tokens,
                                   int tokens_padded, const imf_fusion_weights *w /* [host] */, float scale, float *out,
                                   void *workspace, size_t workspace_bytes, int32_t *flags, int variant, void *stream);

typedef struct imf_net_conv {          /* static half of one fused convolution */
  const float *w_packed;               /* imf_pack_weights / imf_pack_weights_split16 image (see variant) */
  int32_t kvol, cin, cout;
  const float *scale, *shift;          /* folded BatchNorm or bias; NULL = identity */
  int32_t relu, l2norm, variant;
} imf_net_conv;

/* ---- Image branch of the fusion block ------------------------------------------------------------
 * Replaces: ImageEncoder / ResNet.forward (model/Img_Encoder.py:15-18, model/resnet.py:195-216: conv7x7/2 -
 *           bn - relu - maxpool3/2 - layer1 (3 BasicBlocks, 64) - layer2 (4 BasicBlocks, 128, first stride 2 with
 *           the 1x1 projection), model/resnet.py:35-72) and the image-only half of the cross attention
 *           (norm_context + to_kv, model/attention_fusion.py:36-46,84; token layout of model/resunet.py:257-261).
 * Every convolution runs on imf_spconv_fwd's kernel over static pixel tables (a dense 3x3 conv is a
 * sparse conv with a full rulebook; rows = NHWC pixels), BatchNorm folded into scale / shift.
 * Weight images (device, built once per model): conv weights torch [co][ci][ky][kx] re-laid as
 * [k = ky*3+kx][ci][co] and packed with imf_pack_weights(_split16) according to `variant`; the stem as a
 * [1][160][64] kernel whose row (ky*7+kx)*3 + ch holds w[co][ch][ky][kx] (rows 147..159 zero);
 * kv_w = to_kv.weight^T [128][256].
 * conv order: layer1.{0,1,2}.{conv1,conv2} (0-5), layer2.0.conv1 (6), layer2.0.downsample (7), layer2.0.conv2 (8),
 * layer2.{1,2,3}.{conv1,conv2} (9-14); relu = 1 on every entry except the projection (7). */
typedef struct imf_image_desc {
  const float *stem_w, *stem_scale, *stem_shift;
  imf_net_conv conv[15];
  const float *ln_g, *ln_b;            /* norm_context (LayerNorm 128, eps 1e-5); NULL if K/V are not wanted */
  const float *kv_w;                   /* packed to_kv^T [1][128][256] */
  int32_t variant;                     /* imf_conv_args.variant of the stem and K/V projections */
} imf_image_desc;

/* Bytes of the per-(B,H,W) workspace: static pixel tables (filled once by imf_image_tables_build) followed
 * by the feature buffers of one pass.  256-byte aligned. */
size_t imf_image_workspace_bytes(int B, int H, int W);
int imf_image_tokens(int H, int W);    /* tokens per image = (H/8)*(W/8) for H, W divisible by 8 */
int imf_image_tables_build(int B, int H, int W, void *workspace, size_t workspace_bytes, void *stream);
/* image: [B,3,H,W] fp32.  feat_out (optional): [B*tokens, 128] = the stride-8 map in NHWC order, i.e.
 * image_features.view(B,128,-1).permute(0,2,1) of model/resunet.py:257-261.  kt_packed / v_packed (optional,
 * both or neither): per image b at offset b*128*tokens_padded floats, K^T [128, tokens_padded] and
 * V [tokens_padded, 128] in imf_pack_weights(kvol = 1) order, zero-padded -- what imf_fusion_attention reads.
 * About 22 launches, no host synchronisation. */
int imf_image_branch(const imf_image_desc *net /* [host] */, const float *image, int B, int H, int W,
                     void *workspace, size_t workspace_bytes, float *feat_out, float *kt_packed,
                     float *v_packed, int tokens_padded, int32_t *flags /* device, optional: IMF_FLAG_RANGE */,
                     void *stream);

/* imf_fusion_attention_batched + IMF_FLAG_RANGE on the output rows (they feed conv4_tr); flags may be NULL. */
int imf_fusion_attention_batched_flags(const float *x, int n_items, const int64_t *item_row0, const int64_t *item_rows,
                                       const float *const *kt_packed, const float *const *v_packed, int n_tokens,
                                       int tokens_padded, const imf_fusion_weights *w /* [host] */, float scale,
                                       float *out, void *workspace, size_t workspace_bytes, int32_t *flags, void *stream);
/* Capacity mode: x has room for n_cap rows, the count is *n_dev and item b covers rows [item_starts_dev[b],
 * item_starts_dev[b+1]) (the last one up to the count); an item without rows raises bit 3 (value 8) of *err.  The same
 * three launches as the exact-size call (grids sized for the capacity, surplus tiles exit at once): bit-identical.
 * workspace: imf_fusion_workspace_bytes_cap(n_cap). */
size_t imf_fusion_workspace_bytes_cap(int64_t n_cap);
int imf_fusion_attention_dyn(const float *x, int64_t n_cap, const int32_t *n_dev, const int32_t *item_starts_dev,
                             int n_items, int32_t *err, const float *const *kt_packed, const float *const *v_packed,
                             int n_tokens, int tokens_padded, const imf_fusion_weights *w /* [host] */, float scale,
                             float *out, void *workspace, size_t workspace_bytes, void *stream);

/* ---- Native executor for the ResUNet layer schedule -----------------------------------------------
 * One call per fragment replaces the ~100 per-layer calls of model/resunet.py:163-235 (rulebook builds
 * on a side stream, first convolution, encoder, bottleneck fusion, decoder, head), in the launch order
 * and with the arithmetic of the per-layer entry points above.  Configuration: batch 1, one attention
 * head, fusion depth 0 (IMFNet's).  All pointers inside the structs are device pointers except where
 * noted; the structs themselves live on the host. */

typedef struct imf_resunet_desc {
  int32_t channels[5], tr_channels[5]; /* CHANNELS / TR_CHANNELS of model/resunet.py:22-23 (index 0 unused) */
  int32_t in_channels, out_channels, first_ksize;
  int32_t small_first;                 /* in_channels <= 4: conv1 runs as imf_conv_first_* (first_* below) */
  /* 0 conv1 | 1,2 block1 | 3 conv2 | 4,5 block2 | 6 conv3 | 7,8 block3 | 9 conv4 | 10,11 block4 |
   * 12 conv4_tr | 13,14 block4_tr | 15 conv3_tr | 16,17 block3_tr | 18 conv2_tr | 19,20 block2_tr |
   * 21 conv1_tr | 22 final */
  imf_net_conv conv[23];
  const float *first_kernel, *first_scale, *first_shift;   /* conv1 [kvol][cin][cout] unpacked + norm1 */
  imf_fusion_weights fusion;
  float fusion_scale;
  const float *first_kernel_image;     /* optional: imf_pack_first_kernel(first_kernel) -- conv1's hi / lo f16 weight image, so
                                          that the fragment forward's conv1 workgroups copy 16 KiB instead of re-splitting the
                                          kernel each (same bits either way); NULL: split in the kernel */
} imf_resunet_desc;
/* conv1 (k = 3 or 5, Cin = 1, Cout 32 / 64, model/resunet.py:42-49) as the f16 matrix pipe reads it: image of
 * imf_first_kernel_image_floats(kvol, cout) floats, 16-byte aligned.  One-time re-layout at model load; replaces nothing. */
int64_t imf_first_kernel_image_floats(int kvol, int cout);
int imf_pack_first_kernel(const float *w /* [kvol][1][cout] */, int kvol, int cout, float *image, void *stream);

typedef struct imf_net_trace {         /* optional per-convolution measurement record (index = conv id) */
  void *ev_begin, *ev_end;             /* in: hipEvent_t pair recorded around the main kernel */
  const int32_t *nbr;                  /* out: the launch's neighbour table (NULL for 1x1x1) */
  int32_t kvol, cin, cout, split;
  int64_t n_slots, n_out;
  int32_t launched;
  int32_t level, slots_extra;          /* out: pyramid level of the output rows; rulebook slots beyond roundup64(rows) */
  int32_t kernel_tag;                  /* out: imf_conv_args.kernel_tag of the launch (imf_resunet_conv_kernel_tag) */
} imf_net_trace;

/* Which LDS-DMA kernel (variants 3, 6 and 0 alike: the arithmetic is a template argument of the same two kernels) the
 * ResUNet executors (imf_resunet_forward, imf_fragment_forward and the Python plan that mirrors them) use for a convolution whose OUTPUT rows live on pyramid level `level` (0 = tensor stride 1):
 * the value for imf_conv_args.kernel_tag.  Level 0 (thousands of 64-row tiles): k_spconv_g, unsplit (variant 3: the
 * wave-split kernel with 4 wavefronts and 48-row units, kernel_tag 8 | 128, for the 64 -> 64 layers; with half tiles,
 * 8 | 64 | 256, for the 128 -> 64 up-convolution).  Level 1: the
 * wave-split kernel with 4 wavefronts per workgroup (kernel_tag 8; variant 3: 8 | 256); levels 2 and 3: with 8 (kernel_tag 4); variant 3 with
 * ONE fragment in the forward (n_items == 1): half-tile workgroups of 4 wavefronts (8 | 64 | 256) on level 1 and of 8 (4 | 64 | 256) on levels 2-3; every variant
 * with two or more fragments: 48-row units (4 | 128) on level 3, and from three fragments on 4 wavefronts on level 2.  The choice is
 * a function of the LEVEL, the layer's channel counts and the batch size only -- never of the row count -- so exact mode,
 * capacity mode and a graph replay form every sum in the
 * same order (bit-identical descriptors) without a device-side split rule.  What DOES depend on n_items is the partition of a
 * row's sum (wavefront count, unit shape): the descriptors of a fragment computed alone and computed inside a batch of two
 * or more agree to fp32 round-off (~1e-7), not bit for bit, and the same holds between batch sizes 2 and >= 3; set
 * n_items consistently where reproducibility across batch sizes matters.  No executor launch uses split-K partitions
 * or the k_spconv_reduce pass any more.  0 for shapes the wave-split kernel does not serve (kvol == 1, cout % 64 != 0,
 * a variant other than 6 / 3 / 0).  Replaces: the implicit per-layer algorithm choice inside
 * ME.MinkowskiConvolution (model/resunet.py:168-226). */
int imf_resunet_conv_kernel_tag(int level, int kvol, int cin, int cout, int variant, int n_items);
/* For which pyramid levels (bit i = level i, 0 .. 2) the executors (imf_resunet_forward, imf_fragment_forward, the Python plan)
 * build an occupancy-sorted TWIN of the stride-1 map (imf_rulebook_sort_by_occupancy) that the DECODER's block of the level
 * walks -- block2_tr / block3_tr / block4_tr of model/resunet.py:136-146; the encoder's blocks walk the map as built (a sort in
 * front of them would sit on the critical path).  A function of the arithmetic (imf_conv_args.variant of the layers: 7 for
 * bf16x3, 0 for fp32 MFMA and split-f16, where the sorts cost more than the twins return) and of the process environment
 * (IMF_SORTED_MAP overrides) only, so every execution mode walks the same maps and forms the same sums.  Replaces: nothing in
 * the reference. */
int imf_resunet_sorted_maps(int variant);
/* ... and of the batch size of the forward (static per call, like imf_resunet_conv_kernel_tag's n_items): with ONE fragment
 * only level 0 has a twin (its decoder reaches the coarse levels before their sorts would pay).  imf_resunet_sorted_maps(v) is
 * imf_resunet_sorted_maps_n(v, 2): the superset, which sizes the arenas. */
int imf_resunet_sorted_maps_n(int variant, int n_items);

typedef struct imf_resunet_io {        /* per fragment */
  imf_level level[4];                  /* tensor strides 1, 2, 4, 8 (imf_pyramid_build) */
  int64_t n[4];                        /* rows per level */
  const int32_t *bbox;                 /* [host] level-0 bounding box (8 ints) or NULL */
  const float *x;                      /* [n0, in_channels] input features (may be NULL when x_all_ones) */
  int32_t x_all_ones;                  /* util/misc.py:76-79 occupancy feature */
  int32_t n_items;                     /* fragments in the batch (1..IMF_MAX_BATCH), rows grouped by item */
  int64_t item_row0[IMF_MAX_BATCH];    /* first stride-8 row of item b ... */
  int64_t item_rows[IMF_MAX_BATCH];    /* ... and how many (imf_pyramid_build_batched's meta) */
  const float *kt_packed[IMF_MAX_BATCH], *v_packed[IMF_MAX_BATCH];   /* image b's tokens: K^T / V, packed */
  int32_t n_tokens, tokens_padded;
  void *image_ready;                   /* hipEvent_t the main stream waits on before the fusion, or NULL */
  void *fusion_done;                   /* hipEvent_t recorded on the main stream right after the fusion
                                          (the image tokens' K/V may be overwritten after it), or NULL */
  void *int_arena;  size_t int_arena_bytes;     /* >= imf_resunet_int_arena_bytes   */
  void *float_arena; size_t float_arena_bytes;  /* >= imf_resunet_float_arena_bytes */
  float *out;                          /* [n0, out_channels] descriptors */
  void *events[16];                    /* caller-owned hipEvent_t (imf_event_create): side-stream joins (10 used: seven
                                          maps + up to three sorted twins; 9 with `pyramid`) */
  void *side_stream, *main_stream;
  imf_net_trace *trace;                /* [host] 23 records or NULL */
  /* Capacity mode (dyn != 0): n[] are CAPACITIES (arenas: the *_cap size queries), the row counts, the level-0
   * bounding box and the items' first rows are read on the device from `meta`, the block imf_pyramid_build_dyn
   * writes (item_row0 / item_rows / bbox above are ignored); flags raised by the kernels are OR-ed into meta[1]:
   * 1 coordinate out of range, 2 a level exceeded its capacity, 4 bounding box larger than the bit grid, 8 an item
   * without rows, 16 fewer rows than a quarter of the capacity (split cover exceeded).  A flagged result must be
   * discarded and the fragment redone with larger capacities or in exact mode.  Needs ONE convolution variant for all
   * layers (6, or 0 = the strict-fp32 arithmetic), the all-ones occupancy input and in_channels 1.  Bit-identical to the
   * exact mode of the same variant. */
  int32_t dyn;
  const int32_t *meta;
  size_t bitgrid_words;                /* capacity (uint32 words) of the conv1 occupancy grid inside the int arena */
  const void *pyramid;                 /* internal (imf_fragment_forward): coarse pyramid levels still to be built */
  int32_t *flags;                      /* exact mode, optional: device word the kernels OR IMF_FLAG_RANGE into (caller zeroes) */
  int32_t fp32_buffers;                /* 0 (default): with every convolution on variant 6 the layers hand their outputs on as
                                          split-f16 operand images (imf_conv_args.operand_format) -- residual reads see 22 of
                                          the 24 significant bits; 1: every feature buffer stays fp32 (the arithmetic of the
                                          op-by-op path: each imf_spconv_fwd with operand_format 0) */
} imf_resunet_io;

size_t imf_resunet_int_arena_bytes(const imf_resunet_desc *net, const int64_t *n /* [4] */,
                                   const int32_t *bbox /* [host] or NULL */);
size_t imf_resunet_float_arena_bytes(const imf_resunet_desc *net, const int64_t *n /* [4] */);
int imf_resunet_forward(const imf_resunet_desc *net /* [host] */, const imf_resunet_io *io /* [host] */);
size_t imf_resunet_int_arena_bytes_cap(const imf_resunet_desc *net, const int64_t *row_caps /* [4] */, size_t bitgrid_words);
size_t imf_resunet_float_arena_bytes_cap(const imf_resunet_desc *net, const int64_t *row_caps /* [4] */);

/* ---- Whole fragment in one call, capturable as a hipGraph -----------------------------------------------
 * Replaces: util/misc.py:67-104 extract_features (voxelise -> SparseTensor -> model(stensor, image)) for a fragment
 * or a batch of fragments: pyramid (level 0 on the main stream, coarser levels interleaved with the rulebook
 * builds on the side stream), image branch on its own stream, conv1, encoder, fusion, decoder, head -- about 150
 * launches and no host synchronisation.  Every size below is a CAPACITY; the per-fragment scalars live in `dyn`
 * (device int32[IMF_DYN_WORDS]: [0] points, [1] items, [2 + b] first point of item b), which the caller writes
 * before the launch (or before each replay).  After completion meta (device int32[IMF_META_WORDS]) holds, for
 * level l, the row count at [2l], flags at [1] (imf_resunet_io), the level-0 bounding box at [8..15] and the first
 * row of item b of level l at [16 + IMF_MAX_BATCH*l + b]; out[0 .. meta[0]) are the descriptors, levels[0].first_idx
 * the voxels' first points.  Because addresses and grids depend on the capacities only, the call can be recorded
 * between imf_graph_begin_capture / imf_graph_end_capture on main_stream and replayed for every fragment that
 * fits the capacities (side_stream / image_stream join the capture through the events). */
#define IMF_DYN_WORDS  16
#define IMF_META_WORDS 64
typedef struct imf_fragment_caps {     /* [host] one capacity bucket */
  int64_t n_points;                    /* points (all items together) */
  int64_t rows[4];                     /* voxels at tensor strides 1, 2, 4, 8 */
  int32_t n_items, img_h, img_w;
  size_t bitgrid_words;                /* conv1 occupancy grid: >= imf_bitgrid_words of the largest bounding box, a multiple of 4 */
} imf_fragment_caps;

typedef struct imf_fragment_io {       /* [host]; all buffers device memory owned by the caller */
  const void *xyz; int32_t xyz_is_f64; /* [caps.n_points, 3] */
  double voxel_size;
  const int32_t *dyn;                  /* IMF_DYN_WORDS */
  const float *image;                  /* [n_items, 3, img_h, img_w] */
  int32_t *meta;                       /* IMF_META_WORDS, written */
  void *pyramid_arena; size_t pyramid_arena_bytes;   /* >= imf_fragment_pyramid_bytes(caps), 256-byte aligned */
  void *image_ws; size_t image_ws_bytes;             /* imf_image_workspace_bytes, tables built */
  float *kt_packed, *v_packed; int32_t tokens_padded; /* n_items * 128 * tokens_padded floats each (scratch) */
  void *int_arena; size_t int_arena_bytes;           /* >= imf_resunet_int_arena_bytes_cap   */
  void *float_arena; size_t float_arena_bytes;       /* >= imf_resunet_float_arena_bytes_cap */
  float *out;                          /* [caps.rows[0], out_channels] */
  void *events[16];                    /* 11 caller-owned hipEvent_t */
  void *main_stream, *side_stream, *image_stream;
  imf_net_trace *trace;                /* [host] 23 records or NULL */
  imf_level levels[4];                 /* out [host]: where the levels live inside pyramid_arena */
  int32_t serialize;                   /* measurement aid: issue everything on main_stream (no overlap between branches) */
  int32_t fp32_buffers;                /* as imf_resunet_io.fp32_buffers */
  /* Head on the side stream (a STREAM of forwards over several buckets).  0: the table reset, the level-0 pyramid and the
   * image fork are issued on main_stream, i.e. behind everything the main stream still holds -- the previous forward's
   * decoder.  1: they are issued on side_stream, which is idle from the middle of the previous forward on, and the main
   * stream joins in front of conv1: forward k+1's first ~60 us run under forward k's last convolutions (same kernels,
   * same results).  The head then no longer inherits the main stream's order, so the caller states its two dependencies:
   * inputs_event (NULL: none) = after which xyz / dyn / image are in place; reuse_event (NULL: none) = after which earlier
   * work on ANY stream no longer touches this bucket's buffers (the end of its previous forward and of whatever read its
   * outputs).  Not inside a hipGraph capture. */
  int32_t head_on_side;
  /* Hint, 0 or 1: the GPU is idle when this forward is issued (a synchronous call: nothing queued on main_stream).  The executor
   * then issues the side stream's pieces (coarse levels, rulebooks, transposed maps) right before the first main-stream launch
   * that waits for each instead of all of them ahead of conv1 -- same streams, events and results; the first convolution starts
   * ~0.1 ms of host time earlier.  With work queued ahead (a stream of forwards) leave it 0.  imf_pipeline_* sets it per job. */
  int32_t gpu_idle_hint;
  void *inputs_event, *reuse_event;
} imf_fragment_io;

size_t imf_fragment_pyramid_bytes(const imf_fragment_caps *caps);
int imf_fragment_forward(const imf_resunet_desc *net /* [host] */, const imf_image_desc *img /* [host] */,
                         const imf_fragment_caps *caps, imf_fragment_io *io);

/* ---- Streaming pipeline: host arrays in -> descriptors on the host, transfers under the neighbouring forwards ----
 * Replaces: the per-fragment body of scripts/generate_desc.py:99-123 (extract_features(...) on host arrays, then
 * feature.detach().cpu().numpy()) = util/misc.py:82-104 end to end, for a stream of fragments.  A job = one
 * imf_fragment_forward (one fragment or a batch) whose inputs sit in a PINNED host block laid out like the bucket's
 * device input block [dyn | images | points] and whose results land in a pinned block laid out like the device output
 * block [meta | xyz_down | descriptors]:
 *     image stream: host_in -> dev_in            (in_bytes; issued behind the PREVIOUS job's image branch)
 *     main stream : imf_fragment_forward(job->io), then xyz_down = xyz[first_idx] into `sel` (imf_gather_points)
 *     side stream : dev_out -> host_out          (meta, and the meta[0] rows of xyz_down and of the descriptors; with
 *                   defer_download issued behind the NEXT job's launches, i.e. behind its coarse levels and rulebooks)
 * ordered by events, so job k+1's upload and job k-1's download run under job k's kernels, in the idle halves of the two
 * streams a forward has besides its main stream (HIP has four hardware queues: further streams would share one of these
 * queues in an order nobody chose).  The transfers are copy KERNELS that address the pinned blocks directly (no copy
 * command ever sits in front of a launch: a hipMemcpyAsync issued behind queued kernels blocks the calling thread on
 * this stack); IMF_PIPELINE_SDMA_COPIES selects hipMemcpyAsync of the whole blocks instead (A/B).  Every HIP call of a
 * job is made by the pipeline's own worker thread: imf_pipeline_submit only queues a copy of the descriptor and returns
 * a ticket (>= 0) at once, imf_pipeline_wait blocks until that job's results are in host_out (it ends a deferral) and
 * frees the ticket.  The caller owns all memory and must not reuse a bucket (job->io and its blocks) or its pinned blocks
 * before the job's wait has returned; jobs run in submit order.  main_stream: a non-blocking stream distinct from every
 * job's io->side_stream / io->image_stream (all jobs of a pipeline should share those two). */
typedef struct imf_job {               /* [host] */
  const imf_resunet_desc *net;
  const imf_image_desc *img;
  const imf_fragment_caps *caps;
  imf_fragment_io *io;                 /* the bucket; main_stream is set by the pipeline */
  const void *host_in; void *dev_in; size_t in_bytes;       /* pinned -> device, 16-byte aligned */
  const void *dev_out; void *host_out; size_t out_bytes;    /* device -> pinned (whole block; SDMA mode copies all of it) */
  double *sel;                         /* device [caps->rows[0], 3] inside dev_out at sel_offset: xyz[first_idx]; NULL: none */
  size_t sel_offset;                   /* byte offset of `sel` in the output block */
  size_t out_offset; int32_t out_row_bytes;                 /* the descriptors' offset in the output block and bytes per row */
  int32_t defer_download;              /* another job follows: issue this one's download behind that job's launches */
} imf_job;
#define IMF_PIPELINE_SDMA_COPIES 1     /* flags bit 0; bits 8..19: workgroups per copy kernel (0 = 64) */
void *imf_pipeline_create(void *main_stream, int depth /* tickets */, int flags);
void imf_pipeline_destroy(void *pipeline);
int imf_pipeline_submit(void *pipeline, const imf_job *job);          /* ticket >= 0, or IMF_E* */
/* ms [host], 24 floats or NULL.  Device (HIP events): [0] upload, [1] upload done -> forward done (includes queueing behind
 * the previous forward), [2] forward done -> download done (includes a deferral).  Host clock: [3] submit -> taken by the
 * worker, [4] -> forward issued, [5] -> download issued, [6] -> seen complete, [7] time spent inside this call.  Time stamps
 * in ms since imf_pipeline_create -- device clock: [8] upload begins, [9] upload ends, [10] forward begins, [11] forward
 * ends, [12] download ends; host clock: [13] submit, [14] forward issued, [15] this call returns, [16] taken by the worker,
 * [17] / [18] upload issue begins / ends, [19] / [20] download issue begins / ends. */
int imf_pipeline_wait(void *pipeline, int ticket, float *ms);
/* [host pointers] float64 -> float32 when every value is a float32 (PLY points widened by the reader,
 * scripts/generate_desc.py:83-84): returns 1 and dst holds the narrowed values, 0 when some value does not survive
 * (the caller stages the float64 rows instead).  Voxels from the narrowed points are bit-identical (the voxeliser
 * widens before its fp64 divide, util/misc.py:82). */
int imf_host_narrow_points(const double *src, int64_t n_values, float *dst);

/* hipGraph plumbing for the above (thread-local capture mode: other host threads may keep using HIP).
 * imf_graph_end_capture instantiates; the handle is replayed with imf_graph_launch on any stream. */
int imf_graph_begin_capture(void *stream);
int imf_graph_end_capture(void *stream, void **graph_exec_out /* [host] */, int *n_nodes_out /* [host] or NULL */);
int imf_graph_abort_capture(void *stream);
int imf_graph_launch(void *graph_exec, void *stream);
void imf_graph_destroy(void *graph_exec);

/* ---- Descriptor matching for feature-match recall ---------------------------------
 * imf_nn_search replaces util/uio.py:245-258 `knn_search(points_src, points_dst, k=1)` (one Open3D
 * KD-tree query per row, fp64) as called twice at scripts/evaluation_3dmatch.py:207-210: for every
 * row of `query` [n_query, dim] the index of the row of `db` [n_db, dim] with the smallest squared L2
 * distance.  Exact: scores are formed in fp64 (f64 matrix pipe); ties go to the lowest index.
 * dim in {16, 32, 64}.  nn_dist2 (optional, may be null) receives the squared distance in fp64.
 * workspace: imf_nn_workspace_bytes(n_query, n_db) bytes of device memory. */
size_t imf_nn_workspace_bytes(int64_t n_query, int64_t n_db);
int imf_nn_search(const float *query, int64_t n_query, const float *db, int64_t n_db, int dim,
                  int32_t *nn_index, double *nn_dist2, void *workspace, size_t workspace_bytes,
                  void *stream);
/* imf_mutual_inliers replaces scripts/evaluation_3dmatch.py:212-234: frag2 index j survives iff
 * nn12[nn21[j]] == j; survivors are written ascending to match_idx2 (capacity n2).  When kpts1
 * [n1,3], kpts2 [n2,3] (device fp64) and pose_host (HOST pointer, 16 doubles, row-major 4x4) are all
 * non-null, each surviving frag2 keypoint is transformed by the pose (homogeneous multiply and
 * divide by w, as Open3D's PointCloud::transform) and counted as an inlier when its distance to
 * kpts1[nn21[j]] is < inlier_thresh.  meta (device int32[2]) = {n_matches, n_inliers}. */
int imf_mutual_inliers(const int32_t *nn21, int64_t n2, const int32_t *nn12, int64_t n1,
                       const double *kpts1, const double *kpts2, const double *pose_host,
                       double inlier_thresh, int32_t *match_idx2, int32_t *meta, void *stream);

/* ---- Keypoint -> voxel selection of the evaluator ----------------------------------
 * Replaces scripts/evaluation_3dmatch.py:162-171: the ascending indices i of the rows of `coords`
 * [n_voxels,3] (the `xyz` array of a descriptor file, device fp64) whose key
 * ME.utils.fnv_hash_vec(floor(coords[i] / voxel_size)) occurs among the keys of `samples`
 * [n_samples,3] (the sampled raw points, device fp64) -- np.where(np.isin(key_coords, key_points))[0].
 * inds: device int32, capacity n_voxels; count: device int32[1].  Same FNV-1a-64 key as the
 * reference, so the selected set is identical even under a key collision. */
size_t imf_keypoint_workspace_bytes(int64_t n_samples, int64_t n_voxels);
int imf_select_keypoints(const double *samples, int64_t n_samples, const double *coords, int64_t n_voxels,
                         double voxel_size, int32_t *inds, int32_t *count, void *workspace,
                         size_t workspace_bytes, void *stream);

/* ---- RANSAC registration on feature correspondences -------------------------------
 * Replaces scripts/benchmark_util.py:16-34 run_ransac = Open3D 0.12
 * registration_ransac_based_on_feature_matching(..., TransformationEstimationPointToPoint(False),
 * ransac_n, [EdgeLength(edge_similarity), Distance(max_corr_dist)], RANSACConvergenceCriteria(max_iter,
 * 1000), mutual_filter=False) given the correspondences corres[i] = nearest target feature of source
 * point i (imf_nn_search(src_feat, dst_feat)).  src [n_src,3], dst [n_dst,3] device fp64.  All
 * max_iter hypotheses are drawn (the reference's second criterion never fires); the draw sequence is a
 * counter-based generator of `seed` (Open3D's clock-seeded one cannot be reproduced), identical in
 * the oracle.  out_T: device 16 doubles, row-major 4x4 source->target (identity if nothing survives);
 * out_meta: device int32[3] = {winning iteration or -1, its inlier count, hypotheses that passed the
 * checkers}; out_stats: device double[2] = {fitness, inlier RMSE}. */
size_t imf_ransac_workspace_bytes(int max_iter);
int imf_ransac_registration(const double *src, int64_t n_src, const double *dst, int64_t n_dst,
                            const int32_t *corres, int ransac_n, double max_corr_dist, double edge_similarity,
                            int max_iter, uint64_t seed, double *out_T, int32_t *out_meta, double *out_stats,
                            void *workspace, size_t workspace_bytes, void *stream);

/* ---- Training backward of the sparse convolution (last item) ---------------------------------------
 * Replaces: the backward of ME.MinkowskiConvolution / ConvolutionTranspose under loss.backward(), lib/trainer.py:495-569.
 * The INPUT gradient is imf_spconv_fwd itself over the opposite kernel map with transposed weights
 * (imfnet_amd/autograd.py: same map with W[K-1-k]^T at stride 1, the transposed map for a strided convolution, the
 * strided map for a transposed one).  The WEIGHT gradient is this entry point:
 *     dw[k][ci][co] = sum over the pairs (i, o) of offset k of  in[i][ci] * grad_out[o][co]
 * over the same rulebook the forward used (tile_rows / nbr as in imf_conv_args; nbr NULL = identity when kvol == 1).
 * Any cin, cout >= 1.  Deterministic (chunk partials summed in order).  workspace: imf_spconv_wgrad_workspace_bytes. */
size_t imf_spconv_wgrad_workspace_bytes(int64_t n_slots, int kvol, int cin, int cout);
int imf_spconv_wgrad(const float *in, int cin, const float *grad_out, int cout, const int32_t *tile_rows,
                     const int32_t *nbr, int64_t n_slots, int64_t n_out, int kvol, float *dw /* [kvol][cin][cout] */,
                     void *workspace, size_t workspace_bytes, void *stream);

/* ---- Host-side codecs of the batch path: HOST pointers, no GPU involved ----------------------
 * Replace what the reference does around every fragment with Open3D / matplotlib / OpenCV / numpy
 * (scripts/generate_desc.py:83-97,118-123, util/uio.py:33-40).  All return 0 / a count on success, a negative
 * IMF_E* code otherwise (imf_last_error). */
int64_t imf_ply_vertex_count(const char *path);
/* x,y,z of every vertex as float64 [n,3] = np.array(o3d.io.read_point_cloud(path).points): ascii and both binary
 * byte orders, other scalar vertex properties skipped.  Returns n. */
int64_t imf_ply_read_points(const char *path, double *out /* [capacity,3] */, int64_t capacity);
int imf_png_info(const char *path, int *h, int *w, int *channels);
/* matplotlib.image.imread of a .png: float32 [H,W,C] in [0,1] (8-bit / 255, 16-bit / 65535, palette -> RGB).
 * IMF_EUNSUPPORTED for interlaced or sub-byte files (callers fall back to a generic decoder). */
int imf_png_read_f32(const char *path, float *out, int64_t capacity_floats, int *h, int *w, int *channels);

/* matplotlib.image.imread of a .jpg (scripts/generate_desc.py:88-92: PIL = libjpeg's defaults, integer "islow" inverse DCT
 * and fancy chroma upsampling): uint8 [H, W, 3].  Baseline / extended sequential Huffman files with three YCbCr components in
 * one scan, 4:4:4 / 4:2:2 / 4:2:0, restart intervals: bit-identical to PIL (tests/test_cabi_and_host.py).  IMF_EUNSUPPORTED
 * for everything else a JPEG file may hold (progressive, arithmetic, 12-bit, grey, CMYK, other sampling factors): callers
 * fall back to a generic decoder.  [host] */
int imf_jpeg_info(const char *path, int *h, int *w, int *channels);
int imf_jpeg_read_u8(const char *path, uint8_t *out, int64_t capacity_bytes, int *h, int *w, int *channels);
/* cv2.resize(INTER_LINEAR) for float images [H,W,C]; chw != 0 writes [C,H_out,W_out] (generate_desc.py:96-97). */
int imf_resize_bilinear_f32(const float *in, int H, int W, int C, float *out, int H_out, int W_out, int chw);
/* np.savez (level 0) / np.savez_compressed (level 1..9, raw deflate) of n_arrays C-ordered arrays: names[i] (member
 * name), dtype[i] (numpy descr, e.g. "<f8"), ndim[i], their dims concatenated in shape, data[i].  np.load reads the
 * file; the arrays are identical to numpy's own writers'.  Levels 2..9 are zlib's; level 1 uses the library's own raw-deflate
 * producers where the member allows (8-byte items: matches at value granularity -- the float32-valued float64 coordinates of
 * `points` / `xyz` repeat earlier values exactly 98-99 % of the time --; members without LZ77 matches: byte-wise Huffman),
 * zlib level 1 otherwise: scripts/generate_desc.py:118-123 at ~10x numpy's rate per core. */
int imf_npz_write(const char *path, int n_arrays, const char *const *names, const char *const *dtype,
                  const int32_t *ndim, const int64_t *shape, const void *const *data, int level);
/* The same with block-parallel deflate on `threads` host threads (256 KiB blocks as independent raw-deflate segments that
 * end on a byte boundary, concatenated; CRC-32s combined): the file's bytes are a function of the arrays and the level only,
 * never of `threads`.  threads <= 1 is imf_npz_write.  scripts/generate_desc.py:118-123 at the rate of eight GPUs. */
int imf_npz_write_mt(const char *path, int n_arrays, const char *const *names, const char *const *dtype,
                     const int32_t *ndim, const int64_t *shape, const void *const *data, int level, int threads);

/* Measurement helpers (bench.py): HIP events on the caller's stream. */
void *imf_stream_create(void);     /* non-blocking hipStream_t, distinct from any framework pool stream */
void imf_stream_destroy(void *stream);
void *imf_event_create(void);
void imf_event_destroy(void *ev);
int imf_event_record(void *ev, void *stream);                 /* hipEventRecord */
float imf_event_elapsed_ms(void *ev_begin, void *ev_end);   /* < 0 on error (e.g. not completed) */

#ifdef __cplusplus
}
#endif
#endif /* IMFNET_HIP_H */
