// Native executor for the ResUNet layer schedule and for a whole fragment (host code only: no kernels here).
//
// The reference runs one fragment as ~100 Python-level MinkowskiEngine calls
// (model/resunet.py:163-235).  imf_resunet_forward walks the same schedule natively: rulebook
// builds on a side stream joined by events, the first convolution, the encoder, the fused bottleneck
// attention (after the image branch's event), the decoder and the head -- one call per fragment.
//
// Two modes:
//   exact     the row count of every level is known on the host (one D2H readback after the pyramid build);
//             arenas, grids and split-K partitions are sized for exactly those rows.
//   capacity  (io->dyn) nothing is read back: arenas, rulebooks and grids are sized for CAPACITIES, the kernels
//             take the actual counts from the pyramid's device meta block, and the per-launch choices that depend
//             on the row count (split-K partitions, fusion hidden-split) are made on the device with the same rule
//             as the host -- so a capacity-mode forward is bit-identical to the exact one.  Because every address,
//             grid and argument is then a function of the capacities only, the whole fragment
//             (imf_fragment_forward: pyramid + rulebooks + image branch + 23 convolutions + fusion) can be
//             captured ONCE per capacity bucket as a hipGraph and replayed with no host work but one launch.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "geometry_internal.h"
#include "spconv_shared.h"

namespace imf {
namespace {

struct Rb {   // rulebook inside the int arena
  int32_t *tile_rows = nullptr, *nbr = nullptr;
  uint32_t *tile_mask = nullptr;
  int64_t n_slots = 0, n_out = 0;
  int kvol = 1, max_active = 1;
  int level = 0;          // pyramid level of the OUTPUT rows (row count: meta[2 * level] in capacity mode)
  int slots_extra = 0;    // parity-class padding of transposed maps (slots beyond roundup64(rows))
  int ready_event = -1;   // index into io->events that the main stream must wait on before first use
  size_t words() const { return (size_t)n_slots + (size_t)kvol * n_slots + (size_t)(n_slots / IMF_TILE_ROWS) * IMF_MASK_WORDS; }
  int32_t *place(int32_t *p) {
    tile_rows = p;
    nbr = p + n_slots;
    tile_mask = (uint32_t *)(nbr + (size_t)kvol * n_slots);
    return p + words();
  }
};

struct Sizes {
  int64_t n[4];
  int64_t slots[4], up_slots[3];
  int ch[5], tr[5], dec[3];
  int first_kvol;
  bool small_first;
};

Sizes sizes_of(const imf_resunet_desc *net, const int64_t *n) {
  Sizes s;
  for (int i = 0; i < 4; ++i) {
    s.n[i] = n[i];
    s.slots[i] = imf_rulebook_slots(n[i]);
  }
  for (int i = 0; i < 3; ++i) s.up_slots[i] = imf_rulebook_transpose_slots(n[i]);
  for (int i = 0; i < 5; ++i) {
    s.ch[i] = net->channels[i];
    s.tr[i] = net->tr_channels[i];
  }
  s.dec[2] = s.tr[4];
  s.dec[1] = s.tr[3];
  s.dec[0] = s.tr[2];
  s.first_kvol = net->first_ksize * net->first_ksize * net->first_ksize;
  s.small_first = net->small_first != 0;
  return s;
}

size_t rb_words(int64_t n_slots, int kvol) {
  return (size_t)n_slots + (size_t)kvol * n_slots + (size_t)(n_slots / IMF_TILE_ROWS) * IMF_MASK_WORDS;
}

// the occupancy-sorted twins of the stride-1 maps of levels 0-2 for the decoder's blocks (csrc/rulebook_sort.hip) + the sorts'
// workspace (one: the side stream runs them one after the other), behind the other maps
size_t sorted_map_words(const Sizes &s) {
  size_t w = 64 /* alignment slack */ + imf_rulebook_sorted_workspace_bytes(s.slots[0]) / 4;
  for (int i = 0; i < 3; ++i) w += rb_words(s.slots[i], 27);
  return w;
}

size_t int_words(const Sizes &s) {
  size_t w = 0;
  if (!s.small_first) w += rb_words(s.slots[0], s.first_kvol);
  for (int i = 0; i < 4; ++i) w += rb_words(s.slots[i], 27);
  for (int i = 0; i < 3; ++i) w += rb_words(s.slots[i + 1], 27);
  for (int i = 0; i < 3; ++i) w += rb_words(s.up_slots[i], 27);
  return w + sorted_map_words(s) + 16 * 3;
}

// feature buffers: e{i}{a,b,c} (encoder level i: conv out, block mid, block out), d{i}{a,b,c}, head, fused
enum { E0A = 0, D0A = 12, HEAD = 21, FUSED = 22, NBUF = 23 };
inline int ebuf(int i, int s) { return E0A + 3 * i + s; }
inline int dbuf(int i, int s) { return D0A + 3 * i + s; }

void buffer_floats(const Sizes &s, size_t (&cnt)[NBUF]) {
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 3; ++k) cnt[ebuf(i, k)] = (size_t)s.n[i] * s.ch[i + 1];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) cnt[dbuf(i, k)] = (size_t)s.n[i] * s.dec[i];
  cnt[HEAD] = (size_t)s.n[0] * s.tr[1];
  cnt[FUSED] = (size_t)s.n[3] * s.ch[4];
}

size_t float_arena_bytes(const Sizes &s, bool dyn) {
  size_t cnt[NBUF];
  buffer_floats(s, cnt);
  size_t total = 0;
  for (int i = 0; i < NBUF; ++i) total += (cnt[i] + 63) / 64 * 64;
  // largest split-K workspace of any launch (same rule as imf_spconv_fwd's automatic split)
  size_t ws = 0;
  auto consider = [&](int64_t n_slots, int cout, int max_active) {   // unsplit launches: the optional balanced tail only
    (void)max_active;
    const size_t need = imf_spconv_workspace_bytes(n_slots, cout, 1) / 4;
    ws = ws > need ? ws : need;
  };
  for (int i = 0; i < 4; ++i) {
    consider(s.slots[i], s.ch[i + 1], 27);
    if (i < 3) consider(s.slots[i], s.dec[i], 27);
    if (i > 0) consider(s.slots[i], s.ch[i + 1], 27);   // strided conv into level i
  }
  for (int i = 0; i < 3; ++i) consider(s.up_slots[i], s.dec[i], 8);
  if (!s.small_first) consider(s.slots[0], s.ch[1], s.first_kvol);
  total += ws;
  total += (dyn ? imf_fusion_workspace_bytes_cap(s.n[3]) : imf_fusion_workspace_bytes(s.n[3])) / 4;
  return total * 4 + 2048;   // alignment slack of the three carved regions
}

struct Step {   // one fused convolution of the schedule
  int conv;     // index into imf_resunet_desc::conv
  Rb *rb;
  int in_a, c_a, out, in_b, c_b, residual;   // buffer ids (-1 none; -2 = io->x; -3 = io->out)
};

// What imf_fragment_forward hands to imf_resunet_forward through imf_resunet_io::pyramid (internal): the coarse
// pyramid levels still to be built, and the image branch, to be forked off the main stream after encoder step
// `fork_after` of the schedule (-1: the caller has forked it already).
struct FragmentCtx {
  const PyramidBuild *pb;
  int fork_after;
  const imf_image_desc *img;
  const imf_fragment_caps *caps;
  imf_fragment_io *fio;
  hipStream_t imgs;
  bool head_on_side;   // level 0 was issued on the SIDE stream (imf_fragment_io.head_on_side): the main stream joins it
};

int fork_image_branch(const FragmentCtx &c, hipStream_t main) {
  imf_fragment_io *fio = c.fio;
  IMF_CHECK_HIP(hipEventRecord((hipEvent_t)fio->events[9], main));
  IMF_CHECK_HIP(hipStreamWaitEvent(c.imgs, (hipEvent_t)fio->events[9], 0));
  const int rc = imf_image_branch(c.img, fio->image, c.caps->n_items, c.caps->img_h, c.caps->img_w, fio->image_ws,
                                  fio->image_ws_bytes, nullptr, fio->kt_packed, fio->v_packed, fio->tokens_padded,
                                  fio->meta + 1, c.imgs);
  if (rc) return rc;
  IMF_CHECK_HIP(hipEventRecord((hipEvent_t)fio->events[10], c.imgs));
  return IMF_OK;
}

#ifndef IMF_SORTED_MAPS_DEFAULT
#define IMF_SORTED_MAPS_DEFAULT 0x07               // sorted twins of the stride-1 maps of levels 0, 1, 2 for the decoder (measured: LAB_NOTES round 6)
#endif
constexpr int kMetaBBox = 8;                        // meta[2 * n_levels + 0..7] with n_levels = 4
constexpr int kMetaStarts = 16;                     // meta[16 + IMF_MAX_BATCH * level + item]

}  // namespace
}  // namespace imf

using namespace imf;

extern "C" {

static int conv_kernel_tag_rule(int level, int kvol, int cin, int cout, int variant, int n_items);

int imf_resunet_conv_kernel_tag(int level, int kvol, int cin, int cout, int variant, int n_items) {
  // half tiles (8 | 64): the build for four wavefronts per SIMD (bit 8: 127 VGPRs, 25 KiB of LDS, the partial tiles combined one
  // row block at a time; the same sums).  Isolated, the shapes of a single fragment 299.8 / 298.2 -> 293.8 / 287.1 us in sum, of a
  // pair 479 / 488 -> 478 / 464; in the step, A/B three times over on one box: one fragment 0.8367 / 0.8424 / 0.8383 -> 0.8282 /
  // 0.8296 / 0.8313 ms, a pair (its two up-convolutions) 1.1987 / 1.1955 / 1.1917 -> 1.1930 / 1.1783 / 1.1897.
  // IMF_HALF_OCC4=0 (diagnostic): the three-wavefront build of round 5.
  static const bool occ4 = !getenv("IMF_HALF_OCC4") || atoi(getenv("IMF_HALF_OCC4")) != 0;
  const int tag = conv_kernel_tag_rule(level, kvol, cin, cout, variant, n_items);
  return (occ4 && variant == 3 && (tag & (8 | 64 | 128)) == (8 | 64)) ? (tag | 256) : tag;
}

static int conv_kernel_tag_rule(int level, int kvol, int cin, int cout, int variant, int n_items) {
  if ((variant != 6 && variant != 0 && variant != 3) || kvol <= 1 || cout % 64 != 0) return 0;   // (variants 0 / 3: the same kernels, other AR)
  // Stride-1 level: k_spconv_g, except bf16x3's two 64 -> 64 layers (block1_tr): with its weight halves loaded straight
  // into registers the wave-split kernel takes 126-136 us for such a layer in isolation (half-tile / whole-tile workgroups)
  // where k_spconv_g takes 136-140 (tools/conv_iso.py, VARIANT=3); in situ the pair step goes 1.259 -> 1.252 ms with whole
  // tiles and a further -0.7 % with half tiles (A/B/A/B on one box each) -- these two run at the end of the step, when the
  // side streams are idle.  The 128 -> 64 layer (conv2_tr) loses there (52-56 vs 49 us), and with EVERY stride-1 layer on
  // the wave-split kernel the step was 1.27 -> 1.33 ms (its workgroups leave no room for the side streams' kernels under
  // the encoder).
  // Round 6: 48-ROW UNITS of 4 wavefronts (kernel_tag 8 | 128) instead of half tiles: with the partial tiles combined two row
  // blocks at a time the workgroup needs 41 KiB of LDS and 145 VGPRs -- three per CU like the half tiles, with 18 KiB of
  // operands per 72 MFMAs instead of 16 per 48.  In isolation 136-142 -> 124-130 us (whole tiles at two wavefronts per SIMD:
  // 139-149; at three, kernel_tag 8 | 256: 140-143); pair step, A/B/C/D three times over on one box: half tiles 1.1706 /
  // 1.1699 / 1.1694, units 1.1640 / 1.1637 / 1.1654, whole tiles x 3: 1.1717 / 1.1609 / 1.1690, x 2: 1.1749 / 1.1779 / 1.1699.
  // IMF_L0_TAG (diagnostic): another tag for these two layers (72 = half tiles, 264, 8).
  static const int l0_tag = getenv("IMF_L0_TAG") ? atoi(getenv("IMF_L0_TAG")) : (8 | 128);
  // ... and (round 6) the stride-1 up-convolution conv2_tr (128 -> 64 over the parity-grouped transposed map) on half tiles of 4
  // wavefronts built for four per SIMD: round 5's wave-split kernels lost to k_spconv_g there (52-56 vs 49 us), these do not --
  // headline leg of bench.py, A/B/C/D three times over on one box: k_spconv_g 1.2005 / 1.2076 / 1.2031 ms, half tiles x 4
  // 1.1693 / 1.1692 / 1.1686 (-2.9 %), 48-row units 1.1680 / 1.1701 / 1.1666, whole tiles 1.1785 / 1.1743 / 1.1762; one fragment
  // per forward 0.8117 / 0.8177 / 0.8153 -> 0.7689 / 0.7882 / 0.7681.  IMF_L0_UP_TAG (diagnostic): another tag, 0 = k_spconv_g.
  static const int l0_up_tag = getenv("IMF_L0_UP_TAG") ? atoi(getenv("IMF_L0_UP_TAG")) : (8 | 64);
  if (level <= 0 && variant == 3 && cin > cout && cout == 64) return l0_up_tag;
  if (level <= 0) return (variant == 3 && cin == 64 && cout == 64) ? l0_tag : 0;
  // ONE fragment per forward (the reference's call pattern, resunet.py:163 from generate_desc.py:99): its stride-2/4/8 levels
  // have 219 / 61 / 17 tiles -- as half-tile workgroups of 4 wavefronts (kernel_tag 8 | 64, spconv_w.hip RB 2) they reach twice
  // as many CUs: forward 0.93 -> 0.87 ms.  A pair's levels (438 / 120 / 34 tiles) are NOT faster that way (+0.5-1 %: the
  // wavefronts of a half tile have half the MFMAs per request to hide its latency under) and keep whole tiles.
  // n_items is static in every mode (the batch of a forward), so exact mode, capacity mode and a replay still agree bit
  // for bit.
  {   // (diagnostic, A/B in the step: IMF_L1_TAG replaces the rule for level 1's cin <= cout layers.  Settled the same way and removed
      // again: overrides for levels 2 / 3 and for the strided convolutions -- LAB_NOTES 4g-10 has the arms.)
    static const int l1 = getenv("IMF_L1_TAG") ? atoi(getenv("IMF_L1_TAG")) : 0;
    if (level == 1 && l1 && cin <= cout && variant == 3) return l1;
  }
  // Round 6: its stride-4 / 8 levels (61 / 17 tiles) on half tiles of EIGHT wavefronts built for four per SIMD (4 | 64 | 256: two
  // workgroups per CU, twice the wavefronts on a tile's offset list): 128 -> 128 28.6 -> 23.5 us, 256 -> 256 45.1 -> 36.0,
  // 128 -> 256 25.2 -> 21.2, 256 -> 128 16.8 -> 14.8; the stride-2 level (219 tiles) stays on 4 wavefronts (23.0 vs 24.5 us).
  if (variant == 3 && n_items == 1) return level >= 2 ? (4 | 64 | 256) : (8 | 64);
  // A pair's (or triple's) stride-8 level: 34 tiles x 4 slabs = 136 workgroups for 256 CUs, and every tile-aligned way of
  // cutting them finer gives 17 x 2^n units.  48-row UNITS that ignore the tile boundaries (kernel_tag 4 | 128, spconv_w.hip
  // RB 3; a unit walks the union of its two tiles' offset lists) are 46 x 4 = 184 workgroups of 3 / 4 of the work: the
  // 256 -> 256 layer 67.7 -> 52.2 us, 128 -> 256 38.1 -> 29.5, pair step 1.246 -> 1.215 ms (A/B/A/B on one box).  The stride-4
  // level (120 tiles x 2 slabs = 240 workgroups already) loses with them (41 -> 58 us).
  // (the same on fp32 MFMA: 115.7 -> 91.2 us, step 2.00 -> 1.92 ms; split-f16: 38.6 -> 33.7 us, 0.940 -> 0.933 ms)
  // Larger batches too (units are 3 / 4 of a tile: shorter tails): four fragments per forward 2.197 -> 2.128 ms, eight
  // 3.978 -> 3.858 ms; and from three fragments on the stride-4 level (>= 180 tiles x 2 slabs: more than one round of
  // 8-wavefront workgroups) runs on 4-wavefront workgroups, two per CU: 2.128 -> 2.089 ms and 3.858 -> 3.828 ms.
  if (level == 3 && n_items >= 2) return 4 | 128;
  if (level == 2 && n_items >= 3) return 8;
  // the decoder's up-convolutions (cin > cout: conv3_tr 256 -> 64, conv4_tr 256 -> 128): their tiles are grouped by parity
  // class and walk 1-8 offsets -- short loops, so twice the workgroups help: half tiles 36.1 -> 31.7 us and 24.4 -> 21.8 us
  // in isolation, pair step -0.8 % (A/B/A/B on one box)
  static const int up_tag = getenv("IMF_UP_TAG") ? atoi(getenv("IMF_UP_TAG")) : (8 | 64);   // (IMF_UP_TAG: diagnostic)
  if (variant == 3 && cin > cout) return up_tag;
  // measured on the S50k pair (profiles/r03_conv_isolated.txt, r05_conv_isolated_*.txt): level 1 (438 tiles) is fastest
  // with two 4-wavefront workgroups per CU, levels 2 and 3 (<= 128 tiles) with one 8-wavefront workgroup
  // (round 6, bf16x3: level 1 on the whole-tile build for three wavefronts per SIMD, kernel_tag 8 | 256 -- the same sums; headline
  // leg of bench.py, A/B four times over on one box: 1.1302 / 1.1331 / 1.1342 / 1.1346 -> 1.1236 / 1.1291 / 1.1316 / 1.1316 ms;
  // half tiles x 4 there: 1.152, 48-row units 1.157; levels 2 / 3 on half tiles of 8 or 4 wavefronts: +0.3 ... +4 %)
  if (level == 1) return variant == 3 ? (8 | 256) : 8;
  return 4;
}

int imf_resunet_sorted_maps(int variant) {
  // bit i (0 .. 2): the decoder's block on level i walks an occupancy-sorted twin of the level's stride-1 map.  Default: all
  // three on bf16x3, none on fp32 MFMA / split-f16 -- measured on the S50k pair, A/B/A/B on one box (round 6, LAB_NOTES):
  // bf16x3 1.1919 / 1.1886 -> 1.1735 / 1.1733 ms (level 0 alone: 1.1787), fp32 1.914 -> 1.973, split-f16 0.914 -> 0.958: their
  // coarse-level kernels hold a CU's whole LDS, as the sort's workgroups do, and lose more to the sorts running beside them
  // than the decoder gains.  IMF_SORTED_MAP overrides for every arithmetic (A/B; 0 = none).
  return imf_resunet_sorted_maps_n(variant, 2);
}

int imf_resunet_sorted_maps_n(int variant, int n_items) {
  static const int env = getenv("IMF_SORTED_MAP") ? (int)strtol(getenv("IMF_SORTED_MAP"), nullptr, 0) & 0x07 : -1;
  if (env >= 0) return env;
  if (variant != 3) return 0;
  // ONE fragment per forward: its decoder starts ~0.35 ms into the forward, when the sorts of the coarse levels have barely
  // ended, and their blocks are short -- level 0 alone.  A/B x 4 on one box (tools/step_pair.py SINGLE=1, ms): all three
  // 0.8166 / 0.8091 / 0.8182 / 0.8019, level 0 alone 0.8027 / 0.7951 / 0.7915 / 0.7948, none 0.8061 / 0.8046 / 0.8007 / 0.7941.
  return n_items == 1 ? (IMF_SORTED_MAPS_DEFAULT & 1) : IMF_SORTED_MAPS_DEFAULT;
}

size_t imf_resunet_int_arena_bytes(const imf_resunet_desc *net, const int64_t *n, const int32_t *bbox) {
  if (!net || !n) return 0;
  const Sizes s = sizes_of(net, n);
  size_t words = int_words(s);
  if (s.small_first && bbox) words += imf_bitgrid_words(bbox, net->first_ksize);
  return words * 4 + 256;
}

size_t imf_resunet_float_arena_bytes(const imf_resunet_desc *net, const int64_t *n) {
  if (!net || !n) return 0;
  return float_arena_bytes(sizes_of(net, n), false);
}

size_t imf_resunet_int_arena_bytes_cap(const imf_resunet_desc *net, const int64_t *row_caps, size_t bitgrid_words) {
  if (!net || !row_caps) return 0;
  return (int_words(sizes_of(net, row_caps)) + bitgrid_words) * 4 + 256;
}

size_t imf_resunet_float_arena_bytes_cap(const imf_resunet_desc *net, const int64_t *row_caps) {
  if (!net || !row_caps) return 0;
  return float_arena_bytes(sizes_of(net, row_caps), true);
}

int imf_resunet_forward(const imf_resunet_desc *net, const imf_resunet_io *io) {
  IMF_REQUIRE(net && io, "imf_resunet_forward: null pointer");
  IMF_REQUIRE(io->int_arena && io->float_arena && io->out, "imf_resunet_forward: null arena / out");
  const bool dyn = io->dyn != 0;
  const FragmentCtx *fctx = (const FragmentCtx *)io->pyramid;    // fragment forward: coarse levels still to build, image branch
  const PyramidBuild *pyr = fctx ? fctx->pb : nullptr;
  for (int i = 0; i < 4; ++i)
    IMF_REQUIRE(io->n[i] > 0 && io->level[i].coords && io->level[i].table,
                "imf_resunet_forward: level %d missing", i);
  IMF_REQUIRE(net->small_first || io->x, "imf_resunet_forward: input features required");
  IMF_REQUIRE(io->n_items >= 1 && io->n_items <= IMF_MAX_BATCH, "imf_resunet_forward: n_items=%d", io->n_items);
  const Sizes s = sizes_of(net, io->n);
  const int32_t *meta = io->meta;
  if (dyn) {
    IMF_REQUIRE(meta && io->bitgrid_words > 0, "imf_resunet_forward: capacity mode needs meta and a bit-grid capacity");
    IMF_REQUIRE(s.small_first && io->x_all_ones && net->in_channels == 1 && (net->first_ksize == 3 || net->first_ksize == 5),
                "imf_resunet_forward: capacity mode covers the occupancy-feature first convolution only");
    for (int i = 0; i < 23; ++i)   // variant 6, or variant 0 throughout (the strict-fp32 recompute of a range-flagged fragment)
      IMF_REQUIRE(!net->conv[i].w_packed || net->conv[i].variant == net->conv[12].variant,
                  "imf_resunet_forward: capacity mode needs ONE convolution variant (6 or 0) for all layers");
    IMF_REQUIRE(net->conv[12].variant == 6 || net->conv[12].variant == 0 || net->conv[12].variant == 3,
                "imf_resunet_forward: capacity mode: variant 6, 3 or 0");
    IMF_REQUIRE(io->int_arena_bytes >= imf_resunet_int_arena_bytes_cap(net, io->n, io->bitgrid_words),
                "imf_resunet_forward: int arena %zu < %zu bytes", io->int_arena_bytes,
                imf_resunet_int_arena_bytes_cap(net, io->n, io->bitgrid_words));
    IMF_REQUIRE(io->float_arena_bytes >= imf_resunet_float_arena_bytes_cap(net, io->n),
                "imf_resunet_forward: float arena %zu < %zu bytes", io->float_arena_bytes,
                imf_resunet_float_arena_bytes_cap(net, io->n));
  } else {
    IMF_REQUIRE(!pyr, "imf_resunet_forward: a pending pyramid needs capacity mode");
    IMF_REQUIRE(io->int_arena_bytes >= imf_resunet_int_arena_bytes(net, io->n, io->bbox),
                "imf_resunet_forward: int arena %zu < %zu bytes", io->int_arena_bytes,
                imf_resunet_int_arena_bytes(net, io->n, io->bbox));
    IMF_REQUIRE(io->float_arena_bytes >= imf_resunet_float_arena_bytes(net, io->n),
                "imf_resunet_forward: float arena %zu < %zu bytes", io->float_arena_bytes,
                imf_resunet_float_arena_bytes(net, io->n));
  }
  hipStream_t main = (hipStream_t)io->main_stream, side = (hipStream_t)io->side_stream;
  const int n_events = pyr ? 9 : 10;   // (the occupancy-sorted twins have their own joins: up to three)
  for (int i = 0; i < n_events; ++i) IMF_REQUIRE(io->events[i], "imf_resunet_forward: events[%d] missing", i);
  // flag word: capacity mode collects every flag in the level-0 error word; exact mode takes the caller's (optional)
  int32_t *err = dyn ? const_cast<int32_t *>(meta) + 1 : io->flags;

  // ---- rulebooks in the int arena --------------------------------------------------------------
  Rb rb_first, rb_k3[4], rb_dn[3], rb_up[3], rb_id, rb_k3s[3];
  int32_t *const ibase = (int32_t *)(((uintptr_t)io->int_arena + 255) & ~(uintptr_t)255);
  int32_t *p = ibase;
  if (!s.small_first) {
    rb_first.n_slots = s.slots[0]; rb_first.n_out = s.n[0]; rb_first.kvol = rb_first.max_active = s.first_kvol;
    p = rb_first.place(p);
  }
  for (int i = 0; i < 4; ++i) {
    rb_k3[i].n_slots = s.slots[i]; rb_k3[i].n_out = s.n[i]; rb_k3[i].kvol = rb_k3[i].max_active = 27;
    rb_k3[i].level = i;
    p = rb_k3[i].place(p);
  }
  for (int i = 0; i < 3; ++i) {
    rb_dn[i].n_slots = s.slots[i + 1]; rb_dn[i].n_out = s.n[i + 1]; rb_dn[i].kvol = rb_dn[i].max_active = 27;
    rb_dn[i].level = i + 1;
    p = rb_dn[i].place(p);
  }
  for (int i = 0; i < 3; ++i) {
    rb_up[i].n_slots = s.up_slots[i]; rb_up[i].n_out = s.n[i]; rb_up[i].kvol = 27; rb_up[i].max_active = 8;
    rb_up[i].level = i;
    rb_up[i].slots_extra = 8 * IMF_TILE_ROWS;
    p = rb_up[i].place(p);
  }
  // Occupancy-sorted TWINS of the stride-1 maps (csrc/rulebook_sort.hip; imf_resunet_sorted_maps() says which levels): the
  // slots re-ordered so that the rows of a tile share their missing offsets -- tiles walk ~78 % of the 27 offsets instead of
  // ~100 %.  The ENCODER's blocks walk the maps as built (a sort in front of them sits on the step's critical path: measured
  // +65 us per sorted level, round 6); the DECODER's blocks (block4_tr on level 2, block3_tr on level 1, block2_tr on level 0)
  // walk the twins, which are sorted at the END of the side stream's chain, under the encoder and the fusion.
  const int sorted_maps = imf_resunet_sorted_maps_n(net->conv[19].variant, io->n_items);
  bool twin[3];
  for (int i = 0; i < 3; ++i) {
    twin[i] = ((sorted_maps >> i) & 1) != 0 && (i > 0 || s.small_first);
    rb_k3s[i].n_slots = s.slots[i]; rb_k3s[i].n_out = s.n[i]; rb_k3s[i].kvol = rb_k3s[i].max_active = 27;
    rb_k3s[i].level = i;
    p = rb_k3s[i].place(p);
  }
  int32_t *const sort_ws = (int32_t *)(((uintptr_t)p + 255) & ~(uintptr_t)255);
  const size_t sort_ws_bytes = imf_rulebook_sorted_workspace_bytes(s.slots[0]);
  p += 64 + sort_ws_bytes / 4;
  int32_t *counters = p;
  p += 16 * 3;
  uint32_t *bitgrid = (uint32_t *)p;
  IMF_REQUIRE(p == ibase + int_words(s), "imf_resunet_forward: int arena layout");
  rb_id.n_slots = s.slots[0]; rb_id.n_out = s.n[0]; rb_id.kvol = rb_id.max_active = 1;   // no tables: identity
  rb_id.level = 0;

  auto build_conv = [&](Rb &rb, int lin, int lout, int ksize, hipStream_t st) -> int {
    const imf_level &in = io->level[lin], &out = io->level[lout];
    if (dyn)
      return imf_rulebook_conv_dyn(in.table, in.capacity, out.coords, s.n[lout], meta + 2 * lout,
                                   in.tensor_stride, ksize, rb.tile_rows, rb.nbr, rb.tile_mask, st);
    return imf_rulebook_conv(in.table, in.capacity, out.coords, s.n[lout], in.tensor_stride, ksize,
                             rb.tile_rows, rb.nbr, rb.tile_mask, st);
  };
  int ev = 0;
  auto mark = [&](Rb &rb) -> int {   // record on the side stream; the main stream waits before first use
    IMF_CHECK_HIP(hipEventRecord((hipEvent_t)io->events[ev], side));
    rb.ready_event = ev++;
    return IMF_OK;
  };
  int rc;
  // conv1 + the level-0 map in one launch (fragment forward; measured against two launches on two streams in round 3)
  const bool first_and_map = dyn && pyr && s.small_first && side != main;
  int items_event = -1;
  bool image_joined_side = false;
  if (pyr && fctx->head_on_side) {   // level 0 was built on the side stream, ahead of the main stream: main joins here
    IMF_CHECK_HIP(hipEventRecord((hipEvent_t)io->events[7], side));
    IMF_CHECK_HIP(hipStreamWaitEvent(main, (hipEvent_t)io->events[7], 0));
  } else if (pyr) {   // level 0 was built on the main stream: the side stream (coarse levels, rulebooks) starts after it
    IMF_CHECK_HIP(hipEventRecord((hipEvent_t)io->events[7], main));
    IMF_CHECK_HIP(hipStreamWaitEvent(side, (hipEvent_t)io->events[7], 0));
  }
  if (!s.small_first) {
    if ((rc = build_conv(rb_first, 0, 0, net->first_ksize, main))) return rc;
    if ((rc = build_conv(rb_k3[0], 0, 0, 3, main))) return rc;
  } else if (first_and_map) {
    // (fragment forward: conv1 and the level-0 3x3x3 map are ONE launch on the main stream, below)
  } else {   // conv1 needs no rulebook: k3@1 is built under it
    if ((rc = build_conv(rb_k3[0], 0, 0, 3, side))) return rc;
    if ((rc = mark(rb_k3[0]))) return rc;
  }
  // The side chain in pieces: level i + 1 (coordinates, strided map, stride-1 map: what the encoder needs next) and the tail
  // (item starts, the three transposed maps, the join).  Fragment forward with sorts off the side stream (round 6): each piece is
  // ISSUED right before the first main-stream launch that waits for it instead of all of them up front -- on the GPU nothing
  // changes while the host runs ahead (the streaming pipeline, the bench's steps), but a forward issued into an idle GPU (the
  // synchronous extract_features call) starts its first convolution ~35 launches = ~0.1 ms of host time earlier: that call
  // 1.522 -> 1.446 ms host to host, 1.129 -> 1.086 with the inputs on the device (tools/sync_phases.py; the pair step and the
  // single-fragment step back to back: unchanged).  IMF_EAGER_SIDE=1 (diagnostic): everything up front.  (Deferring the
  // ISSUE of the image branch's launches the same way, behind block1's: measured, no further gain -- 0.970 instead of 0.950 of
  // the eager call -- and dropped.)
  int side_levels_issued = 0;
  bool side_tail_issued = false;
  auto side_level = [&](int i) -> int {
    int rc2;
    if (pyr && (rc2 = pyramid_coarse_level(*pyr, i + 1, side))) return rc2;
    if ((rc2 = build_conv(rb_dn[i], i, i + 1, 3, side))) return rc2;
    if ((rc2 = build_conv(rb_k3[i + 1], i + 1, i + 1, 3, side))) return rc2;
    return mark(rb_dn[i]);
  };
  auto side_tail = [&]() -> int {
    int rc2 = IMF_OK;
    if (pyr) {   // first row of every item at every level (the fusion reads the stride-8 ones)
      if ((rc2 = pyramid_item_starts(*pyr, side, 0, 4))) return rc2;
    }
    for (int i = 2; i >= 0; --i) {
      const imf_level &co = io->level[i + 1], &fi = io->level[i];
      if (dyn)
        rc2 = imf_rulebook_transpose_dyn(co.table, co.capacity, fi.coords, s.n[i], meta + 2 * i, 1 << i, 3,
                                         rb_up[i].tile_rows, rb_up[i].nbr, rb_up[i].tile_mask, rb_up[i].n_slots,
                                         counters + 16 * i, side);
      else
        rc2 = imf_rulebook_transpose(co.table, co.capacity, fi.coords, s.n[i], 1 << i, 3, rb_up[i].tile_rows,
                                     rb_up[i].nbr, rb_up[i].tile_mask, rb_up[i].n_slots, counters + 16 * i, side);
      if (rc2) return rc2;
      if (!pyr && (rc2 = mark(rb_up[i]))) return rc2;
    }
    if (pyr) {
      // Fragment forward: ONE join with the side stream for everything the second half of the step needs (item starts for
      // the fusion, the three transposed rulebooks for the decoder), waited for right before the fusion.  A stream-wait
      // costs the main stream ~5 us even when its event completed long ago (tools/conv_gaps.py: 10.6 us instead of 5.3 in
      // front of every convolution that carried one); the side stream's chain ends ~150 us before the main stream gets there.
      // The image branch joins the SIDE stream here (it was forked before this call, its end event is recorded), so the
      // main stream waits once, not twice, in front of the fusion.
      if (image_joined_side) IMF_CHECK_HIP(hipStreamWaitEvent(side, (hipEvent_t)io->image_ready, 0));
      IMF_CHECK_HIP(hipEventRecord((hipEvent_t)io->events[8], side));
    }
    side_tail_issued = true;
    return IMF_OK;
  };
  // issue the side chain up to (and including) level `upto` (1 .. 3); with `tail` the tail as well
  auto ensure_side = [&](int upto, bool tail) -> int {
    for (; side_levels_issued < upto; ++side_levels_issued) {
      const int rc2 = side_level(side_levels_issued);
      if (rc2) return rc2;
    }
    if (tail && !side_tail_issued) {
      for (; side_levels_issued < 3; ++side_levels_issued) {
        const int rc2 = side_level(side_levels_issued);
        if (rc2) return rc2;
      }
      return side_tail();
    }
    return IMF_OK;
  };
  if (pyr) {
    image_joined_side = io->image_ready && fctx->fork_after < 0 && side != main;
    items_event = 8;
  }
  // ... so only THEN: imf_fragment_io.gpu_idle_hint (the pipeline sets it when no earlier forward is still running; with work
  // queued the host is ahead anyway and the chain goes up front as before -- the streaming pipeline's host span measured 1-2 %
  // worse with the pieces interleaved: 1.198 / 1.183 -> 1.216 / 1.204 ms per pair on one box).  IMF_EAGER_SIDE=1 / 0 (diagnostic)
  // forces either order.
  bool main_idle = fctx && fctx->fio->gpu_idle_hint != 0;
  if (const char *e = getenv("IMF_EAGER_SIDE")) main_idle = atoi(e) == 0;
  const bool lazy_side = pyr && fctx->imgs && fctx->imgs != side && fctx->imgs != main && side != main && main_idle;
  if (!lazy_side && (rc = ensure_side(3, true))) return rc;

  // ---- feature buffers in the float arena ------------------------------------------------------
  size_t cnt[NBUF];
  buffer_floats(s, cnt);
  float *buf[NBUF];
  float *fp = (float *)(((uintptr_t)io->float_arena + 255) & ~(uintptr_t)255);
  for (int i = 0; i < NBUF; ++i) {
    buf[i] = fp;
    fp += (cnt[i] + 63) / 64 * 64;
  }
  float *ws = fp;
  const size_t fusion_ws_floats = (dyn ? imf_fusion_workspace_bytes_cap(s.n[3]) : imf_fusion_workspace_bytes(s.n[3])) / 4;
  float *fusion_ws = (float *)io->float_arena + (io->float_arena_bytes / 4) - fusion_ws_floats - 64;
  fusion_ws = (float *)((uintptr_t)fusion_ws & ~(uintptr_t)255);
  const size_t ws_bytes = ((char *)fusion_ws - (char *)ws);

  for (int i = 0; i < NBUF; ++i)   // variant 6 reads its inputs through a 2 GiB buffer window
    IMF_REQUIRE(cnt[i] * sizeof(float) < (1ull << 31), "imf_resunet_forward: feature buffer %d exceeds 2 GiB", i);

  // ---- schedule (model/resunet.py:168-226) ---------------------------------------------------------
  Step sched[24];
  int n_steps = 0, n_enc = 0;
  for (int i = 0; i < 4; ++i) {
    const int c = s.ch[i + 1];
    if (i > 0) sched[n_steps++] = Step{3 * i, &rb_dn[i - 1], ebuf(i - 1, 2), s.ch[i], ebuf(i, 0), -1, 0, -1};
    else if (!s.small_first) sched[n_steps++] = Step{0, &rb_first, -2, net->in_channels, ebuf(0, 0), -1, 0, -1};
    sched[n_steps++] = Step{3 * i + 1, &rb_k3[i], ebuf(i, 0), c, ebuf(i, 1), -1, 0, -1};
    sched[n_steps++] = Step{3 * i + 2, &rb_k3[i], ebuf(i, 1), c, ebuf(i, 2), -1, 0, ebuf(i, 0)};
  }
  n_enc = n_steps;
  for (int i = 2; i >= 0; --i) {   // output level of conv{i+2}_tr
    const int t = s.dec[i];
    const int conv0 = 12 + 3 * (2 - i);
    const int src = i == 2 ? FUSED : dbuf(i + 1, 2), c_src = i == 2 ? s.ch[4] : s.dec[i + 1];
    const int skip = i == 2 ? -1 : ebuf(i + 1, 2), c_skip = i == 2 ? 0 : s.ch[i + 2];
    sched[n_steps++] = Step{conv0, &rb_up[i], src, c_src, dbuf(i, 0), skip, c_skip, -1};
    Rb *const rbk = twin[i] ? &rb_k3s[i] : &rb_k3[i];
    sched[n_steps++] = Step{conv0 + 1, rbk, dbuf(i, 0), t, dbuf(i, 1), -1, 0, -1};
    sched[n_steps++] = Step{conv0 + 2, rbk, dbuf(i, 1), t, dbuf(i, 2), -1, 0, dbuf(i, 0)};
  }
  sched[n_steps++] = Step{21, &rb_id, dbuf(0, 2), s.tr[2], HEAD, ebuf(0, 2), s.ch[1], -1};
  sched[n_steps++] = Step{22, &rb_id, HEAD, s.tr[1], -3, -1, 0, -1};

  auto addr = [&](int id) -> float * {
    if (id == -1) return nullptr;
    if (id == -2) return const_cast<float *>(io->x);
    if (id == -3) return io->out;
    return buf[id];
  };

  // ---- operand formats --------------------------------------------------------------------------------------------
  // With every convolution on the split-f16 pipe, a layer's output is written as the operand image its consumers' main
  // loops would otherwise derive from the fp32 rows again (imf_conv_args.operand_format; IMF_PRESPLIT=0: fp32 buffers as
  // before).  fp32 stays where something other than a variant-6 convolution reads the buffer: the fusion's input
  // (stride-8 block output, read by the fp32-MFMA attention kernel), the descriptors.
  bool presplit = !io->fp32_buffers;
  for (int i = 0; i < n_steps; ++i) presplit &= net->conv[sched[i].conv].variant == 6;
  bool is_split[NBUF];
  for (int i = 0; i < NBUF; ++i) is_split[i] = false;
  auto fmt_of = [&](int id) { return id >= 0 && is_split[id]; };
  auto wants_split = [&](int id) { return presplit && id >= 0 && id != ebuf(3, 2) && id != HEAD; };

  // ---- first convolution (Cin <= 4): occupancy bit grid for the all-ones feature, else hash probing --
  if (s.small_first) {
    const int first_split = wants_split(ebuf(0, 0)) ? 1 : 0;
    bool wrote_split = first_split != 0;
    if (first_and_map) {
      rc = conv_first_and_map_dyn(io->level[0].coords, s.n[0], meta, meta + kMetaBBox, err, net->first_ksize, bitgrid,
                                  io->bitgrid_words, net->first_kernel, s.ch[1], net->first_scale, net->first_shift, 0,
                                  buf[ebuf(0, 0)], first_split, io->level[0].table, io->level[0].capacity, rb_k3[0].tile_rows,
                                  rb_k3[0].nbr, rb_k3[0].tile_mask, main, net->first_kernel_image);
    } else if (dyn && pyr) {   // imf_fragment_forward zeroed the grid before the level-0 pyramid
      rc = conv_first_bitgrid_dyn_cleared(io->level[0].coords, s.n[0], meta, meta + kMetaBBox, err, net->first_ksize, bitgrid,
                                          io->bitgrid_words, net->first_kernel, s.ch[1], net->first_scale,
                                          net->first_shift, 0, buf[ebuf(0, 0)], main, first_split, net->first_kernel_image);
    } else if (dyn) {
      rc = conv_first_bitgrid_dyn_fmt(io->level[0].coords, s.n[0], meta, meta + kMetaBBox, err, net->first_ksize, bitgrid,
                                      io->bitgrid_words, net->first_kernel, s.ch[1], net->first_scale,
                                      net->first_shift, 0, buf[ebuf(0, 0)], main, first_split);
    } else {
      size_t words = 0;
      if (io->x_all_ones && io->bbox && net->in_channels == 1) words = imf_bitgrid_words(io->bbox, net->first_ksize);
      if (words) {
        rc = conv_first_bitgrid_flags_fmt(io->level[0].coords, s.n[0], io->bbox, net->first_ksize, bitgrid, words,
                                          net->first_kernel, s.ch[1], net->first_scale, net->first_shift, 0,
                                          buf[ebuf(0, 0)], err, main, first_split);
      } else {
        rc = imf_conv_first_fused(io->level[0].table, io->level[0].capacity, io->level[0].coords,
                                  s.n[0], 1, net->first_ksize, io->x_all_ones ? nullptr : io->x, net->in_channels,
                                  net->first_kernel, s.ch[1], net->first_scale, net->first_shift, 0,
                                  buf[ebuf(0, 0)], main);
        wrote_split = false;   // (the hash-probing first layer writes fp32)
      }
    }
    if (rc) return rc;
    is_split[ebuf(0, 0)] = wrote_split;
  }

  // ---- the occupancy-sorted twins (level 2 first: the decoder reaches it first).  Fragment forward: on the IMAGE stream,
  // behind the image trunk -- that stream is idle from ~0.45 ms on, whereas the side stream must be free for the next
  // forward's head (streaming pipeline, bench's pipelined mode: with the sorts at the end of the SIDE chain the head of step
  // k + 1 queued behind 240 us of sorts and the steps lost what the twins gain).  Otherwise: at the end of the side chain.
  hipStream_t sort_stream = side;
  if (fctx && fctx->imgs && fctx->imgs != side && fctx->imgs != main && items_event >= 0) sort_stream = fctx->imgs;
  auto issue_sorts = [&]() -> int {
    if (!(twin[0] || twin[1] || twin[2])) return IMF_OK;
    if (sort_stream != side) {
      // Image stream: ONE dependency, on the MAIN stream at the point of the call (behind conv3's launch: the main stream
      // has waited for the side chain's level-1 and level-2 maps by then, and the level-0 map is its own).  Not on the side
      // stream's events: the side stream may have waited for the image branch (image_joined_side), and two captured streams
      // that wait for each other send hipStreamEndCapture into an endless recursion (ROCm 7.2, found with rocgdb).
      IMF_CHECK_HIP(hipEventRecord((hipEvent_t)io->events[6], main));
      IMF_CHECK_HIP(hipStreamWaitEvent(sort_stream, (hipEvent_t)io->events[6], 0));
    } else if (twin[0] && first_and_map) {   // the level-0 map came out of the first convolution's launch on the MAIN stream
      IMF_CHECK_HIP(hipEventRecord((hipEvent_t)io->events[6], main));
      IMF_CHECK_HIP(hipStreamWaitEvent(sort_stream, (hipEvent_t)io->events[6], 0));
    }
    for (int i = 2; i >= 0; --i) {
      if (!twin[i]) continue;
      int rc2 = imf_rulebook_sort_by_occupancy(rb_k3[i].nbr, 27, rb_k3[i].n_slots, s.n[i], dyn ? meta + 2 * i : nullptr,
                                               rb_k3s[i].tile_rows, rb_k3s[i].nbr, rb_k3s[i].tile_mask, sort_ws, sort_ws_bytes,
                                               sort_stream);
      if (rc2) return rc2;
      if (sort_stream != main) {   // an event per twin: the decoder's first block must not wait for the LAST sort (measured, round 6:
        // one event behind all three sorts and one wait: one fragment per forward 0.82 -> 0.875 ms, a pair's one-bucket step
        // 1.187 -> 1.201 -- the level-0 sort ends after the stride-4 block starts; folding the twins into the join in front
        // of the fusion instead: headline leg +0.8 %)
        IMF_CHECK_HIP(hipEventRecord((hipEvent_t)io->events[ev], sort_stream));
        rb_k3s[i].ready_event = ev++;
      }
    }
    return IMF_OK;
  };
  if (sort_stream == side && (rc = issue_sorts())) return rc;

  auto launch = [&](const Step &st) -> int {
    const imf_net_conv &c = net->conv[st.conv];
    IMF_REQUIRE(c.w_packed, "imf_resunet_forward: conv %d has no weights", st.conv);
    IMF_REQUIRE(st.c_a + st.c_b == c.cin, "imf_resunet_forward: conv %d expects %d input channels, got %d",
                st.conv, c.cin, st.c_a + st.c_b);
    Rb &rb = *st.rb;
    if (lazy_side) {   // the side chain's piece this launch (or a later one on the same map) waits for
      int rc2 = IMF_OK;
      for (int i = 0; i < 3; ++i)
        if (&rb == &rb_dn[i] || &rb == &rb_k3[i + 1] || (i < 2 && &rb == &rb_k3s[i + 1])) rc2 = ensure_side(i + 1, false);
      for (int i = 0; i < 3; ++i)
        if (&rb == &rb_up[i]) rc2 = ensure_side(3, true);
      if (rc2) return rc2;
    }
    if (rb.ready_event >= 0) {
      IMF_CHECK_HIP(hipStreamWaitEvent(main, (hipEvent_t)io->events[rb.ready_event], 0));
      rb.ready_event = -1;
    }
    imf_conv_args a;
    memset(&a, 0, sizeof(a));
    a.in_a = addr(st.in_a); a.in_b = addr(st.in_b); a.c_a = st.c_a; a.c_b = st.c_b;
    a.w_packed = c.w_packed; a.kvol = c.kvol; a.cout = c.cout;
    a.tile_rows = rb.tile_rows; a.nbr = rb.nbr; a.tile_mask = rb.tile_mask;
    a.n_slots = rb.n_slots; a.n_out = rb.n_out;
    a.scale = c.scale; a.shift = c.shift; a.residual = addr(st.residual);
    a.relu = c.relu; a.l2norm = c.l2norm; a.out = addr(st.out);
    a.dyn_err = c.variant == 6 && !c.l2norm ? err : nullptr;   // outputs that feed another split-f16 convolution
    if (dyn) {
      a.n_out_dev = meta + 2 * rb.level;
      a.slots_extra = rb.slots_extra;
      a.dyn_err = err;
    }
    // one workgroup (or its wavefronts) owns a tile for all kernel offsets: no split-K partitions, no reduce launch, and
    // the kernel is a function of the level and the layer's channels only -- both modes form the same sums
    a.split_k = 1;
    a.kernel_tag = imf_resunet_conv_kernel_tag(rb.level, c.kvol, c.cin, c.cout, c.variant, io->n_items);
    a.variant = c.variant;
    a.workspace = ws; a.workspace_bytes = ws_bytes;   // split-K partials or the balanced tail's
    IMF_REQUIRE(st.in_b < 0 || fmt_of(st.in_a) == fmt_of(st.in_b), "imf_resunet_forward: conv %d concatenates an operand "
                "image with an fp32 buffer", st.conv);
    const bool out_split = wants_split(st.out) && !c.l2norm;
    a.operand_format = (fmt_of(st.in_a) ? IMF_FMT_A_SPLIT : 0) | (fmt_of(st.residual) ? IMF_FMT_RES_SPLIT : 0) |
                       (out_split ? IMF_FMT_OUT_SPLIT : 0);
    if (st.out >= 0) is_split[st.out] = out_split;
    if (io->trace) {
      imf_net_trace &t = io->trace[st.conv];
      a.ev_begin = t.ev_begin; a.ev_end = t.ev_end;
      t.nbr = rb.nbr; t.kvol = c.kvol; t.cin = c.cin; t.cout = c.cout; t.split = a.split_k;
      t.n_slots = rb.n_slots; t.n_out = rb.n_out; t.launched = 1;
      t.level = rb.level; t.slots_extra = rb.slots_extra; t.kernel_tag = a.kernel_tag;
    }
    return imf_spconv_fwd(&a, main);
  };

  bool sorts_issued = false, saw_conv3 = false;
  for (int i = 0; i < n_enc; ++i) {
    if ((rc = launch(sched[i]))) return rc;
    if (fctx && fctx->fork_after == i && (rc = fork_image_branch(*fctx, main))) return rc;
    // the sorts on the image stream: behind the image branch's fork and behind the launch that made the main stream wait for
    // the level-2 map (conv3, the consumer of rb_dn[1]) -- see issue_sorts
    if (sched[i].rb == &rb_dn[1]) saw_conv3 = true;
    if (sort_stream != side && !sorts_issued && saw_conv3 && i >= fctx->fork_after) {
      if (lazy_side && (rc = ensure_side(2, false))) return rc;
      if ((rc = issue_sorts())) return rc;
      sorts_issued = true;
    }
  }

  // ---- bottleneck fusion (model/resunet.py:237-273) ----------------------------------------------------
  // diagnostic marks (tools/branch_times.py hands events[11], [12] in): the main stream's arrival at the join, the fusion's end
  if (lazy_side && (rc = ensure_side(3, true))) return rc;
  const bool diag_marks = pyr && io->events[11] && io->events[12];
  if (diag_marks) IMF_CHECK_HIP(hipEventRecord((hipEvent_t)io->events[11], main));
  if (io->image_ready && !image_joined_side) IMF_CHECK_HIP(hipStreamWaitEvent(main, (hipEvent_t)io->image_ready, 0));
  if (items_event >= 0) IMF_CHECK_HIP(hipStreamWaitEvent(main, (hipEvent_t)io->events[items_event], 0));
  const int fused_split = wants_split(FUSED) ? 1 : 0;   // the block's output: conv4_tr's operand image
  const int fusion_variant = (net->conv[12].variant == 6 || net->conv[12].variant == 3) ? net->conv[12].variant : 0;
  is_split[FUSED] = fused_split != 0;
  if (dyn)
    rc = fusion_attention_dyn_fmt(buf[ebuf(3, 2)], s.n[3], meta + 6, meta + kMetaStarts + IMF_MAX_BATCH * 3, io->n_items,
                                  err, io->kt_packed, io->v_packed, io->n_tokens, io->tokens_padded, &net->fusion,
                                  net->fusion_scale, buf[FUSED], fusion_ws, fusion_ws_floats * 4, main, fused_split,
                                  fusion_variant);
  else
    rc = fusion_attention_batched_fmt(buf[ebuf(3, 2)], io->n_items, io->item_row0, io->item_rows, io->kt_packed,
                                      io->v_packed, io->n_tokens, io->tokens_padded, &net->fusion,
                                      net->fusion_scale, buf[FUSED], fusion_ws, fusion_ws_floats * 4,
                                      err, main, fused_split, fusion_variant);
  if (rc) return rc;
  if (io->fusion_done) IMF_CHECK_HIP(hipEventRecord((hipEvent_t)io->fusion_done, main));
  if (diag_marks) IMF_CHECK_HIP(hipEventRecord((hipEvent_t)io->events[12], main));

  // The head (conv1_tr + norm + ReLU + final + L2 norm, model/resunet.py:219-233) as one launch when its shapes are the
  // ones imf_pointwise_head serves; bit-identical to the two convolution launches.
  const imf_net_conv &h1 = net->conv[21], &h2 = net->conv[22];
  const int head_cin = s.tr[2] + s.ch[1];
  const bool head_b3 = h1.variant == 3 && h2.variant == 3;           // bf16x3 images: 64 or 96 input channels (head.hip)
  const bool fused_head = h1.w_packed && h2.w_packed && ((h1.variant == 6 && h2.variant == 6) || head_b3) && h1.kvol == 1 &&
                          h2.kvol == 1 && h1.cout == 64 && h2.cin == 64 && h2.cout == 32 && h1.cin == head_cin &&
                          s.tr[2] % 32 == 0 && s.ch[1] % 32 == 0 && head_cin >= 64 && head_cin <= (head_b3 ? 96 : 128) && !h1.l2norm &&
                          (size_t)s.n[0] * (size_t)(s.tr[2] > s.ch[1] ? s.tr[2] : s.ch[1]) * 4 < (1ull << 31);
  const int n_tail = fused_head ? n_steps - 2 : n_steps;
  for (int i = n_enc; i < n_tail; ++i)
    if ((rc = launch(sched[i]))) return rc;
  if (fused_head) {
    imf_head_args a;
    memset(&a, 0, sizeof(a));
    a.in_a = buf[dbuf(0, 2)]; a.c_a = s.tr[2];
    a.in_b = buf[ebuf(0, 2)]; a.c_b = s.ch[1];
    IMF_REQUIRE(is_split[dbuf(0, 2)] == is_split[ebuf(0, 2)], "imf_resunet_forward: the head's two sources differ in format");
    a.a_split = is_split[dbuf(0, 2)] ? 1 : 0;
    a.variant = head_b3 ? 3 : 6;
    a.w1_packed = h1.w_packed; a.scale1 = h1.scale; a.shift1 = h1.shift; a.relu1 = h1.relu; a.c_mid = 64;
    a.w2_packed = h2.w_packed; a.scale2 = h2.scale; a.shift2 = h2.shift; a.l2norm = h2.l2norm; a.c_out = 32;
    a.n = s.n[0];
    a.n_dev = dyn ? meta : nullptr;
    a.out = io->out;
    a.flags = err;
    if (io->trace) {   // one record (conv1_tr's) carries the launch; `final` is marked as not launched
      imf_net_trace &t = io->trace[21];
      a.ev_begin = t.ev_begin; a.ev_end = t.ev_end;
      t.nbr = nullptr; t.kvol = 1; t.cin = h1.cin; t.cout = h1.cout; t.split = 1;
      t.n_slots = rb_id.n_slots; t.n_out = rb_id.n_out; t.launched = 1;
      t.level = 0; t.slots_extra = 0; t.kernel_tag = 16;
      io->trace[22].launched = 0;
    }
    if ((rc = imf_pointwise_head(&a, main))) return rc;
  }
  return IMF_OK;
}

/* ---- one fragment (or batch of fragments), points to descriptors, with no host synchronisation -------- */
size_t imf_fragment_pyramid_bytes(const imf_fragment_caps *caps) {
  if (!caps) return 0;
  return imf_pyramid_arena_bytes_caps(caps->n_points, 4, caps->rows);
}

int imf_fragment_forward(const imf_resunet_desc *net, const imf_image_desc *img, const imf_fragment_caps *caps,
                         imf_fragment_io *fio) {
  IMF_REQUIRE(net && img && caps && fio, "imf_fragment_forward: null pointer");
  IMF_REQUIRE(fio->xyz && fio->dyn && fio->image && fio->meta && fio->pyramid_arena && fio->image_ws && fio->kt_packed &&
                  fio->v_packed && fio->out, "imf_fragment_forward: null buffer");
  IMF_REQUIRE(caps->n_items >= 1 && caps->n_items <= IMF_MAX_BATCH, "imf_fragment_forward: n_items=%d", caps->n_items);
  hipStream_t main = (hipStream_t)fio->main_stream, side = (hipStream_t)fio->side_stream,
              imgs = (hipStream_t)fio->image_stream;
  if (fio->serialize) side = imgs = main;
  else IMF_REQUIRE(side && imgs && side != main && imgs != main && side != imgs, "imf_fragment_forward: three distinct streams");
  for (int i = 0; i < 11; ++i) IMF_REQUIRE(fio->events[i], "imf_fragment_forward: events[%d] missing", i);
  const int ntok = imf_image_tokens(caps->img_h, caps->img_w);
  IMF_REQUIRE(fio->tokens_padded % 64 == 0 && fio->tokens_padded >= ntok && fio->tokens_padded <= 320,
              "imf_fragment_forward: tokens_padded=%d for %d tokens", fio->tokens_padded, ntok);

  // The meta block (row counts, flag words) is reset on the main stream BEFORE the image stream forks: the image branch
  // ORs IMF_FLAG_RANGE into meta[1], which must not race with that reset.
  PyramidBuild pb;
  int rc = pyramid_prepare(pb, fio->xyz, fio->xyz_is_f64, caps->n_points, fio->voxel_size, 0, nullptr, 1, 4, fio->pyramid_arena,
                           fio->pyramid_arena_bytes, fio->meta, fio->levels, fio->dyn, caps->rows);
  if (rc) return rc;
  // conv1's occupancy bit grid (the int arena's tail, as imf_resunet_forward lays it out) is zeroed ahead of the
  // pyramid, by the launch that resets the hash tables (pyramid_init), instead of between the pyramid and conv1
  IMF_REQUIRE(net->first_ksize == 3 || net->first_ksize == 5, "imf_fragment_forward: first_ksize=%d", net->first_ksize);
  {
    IMF_REQUIRE(net->small_first && caps->bitgrid_words > 0 && fio->int_arena, "imf_fragment_forward: needs the occupancy-feature first convolution and a bit-grid capacity");
    IMF_REQUIRE(fio->int_arena_bytes >= imf_resunet_int_arena_bytes_cap(net, caps->rows, caps->bitgrid_words),
                "imf_fragment_forward: int arena %zu < %zu bytes", fio->int_arena_bytes,
                imf_resunet_int_arena_bytes_cap(net, caps->rows, caps->bitgrid_words));
    int32_t *ibase = (int32_t *)(((uintptr_t)fio->int_arena + 255) & ~(uintptr_t)255);
    uint32_t *bitgrid = (uint32_t *)(ibase + int_words(sizes_of(net, caps->rows)));
    IMF_REQUIRE(((uintptr_t)bitgrid & 15) == 0 && caps->bitgrid_words % 4 == 0, "imf_fragment_forward: bit grid must be 16-byte aligned, a multiple of 4 words");
    // ... and FILLED by the level-0 compaction kernel itself (the bounding box comes out of k_insert_points): no
    // k_bitgrid_fill launch between the pyramid and conv1
    pb.grid = bitgrid; pb.grid_words = caps->bitgrid_words; pb.grid_ksize = net->first_ksize;
  }
  // the head (table reset, level 0, image fork): on the main stream, or -- head_on_side -- on the side stream, where it
  // does not queue behind the previous forward's decoder (include/imfnet_hip.h, imf_fragment_io.head_on_side)
  const bool head_on_side = fio->head_on_side && !fio->serialize;
  hipStream_t head = head_on_side ? side : main;
  if (head_on_side) {
    if (fio->reuse_event) IMF_CHECK_HIP(hipStreamWaitEvent(side, (hipEvent_t)fio->reuse_event, 0));
    if (fio->inputs_event) IMF_CHECK_HIP(hipStreamWaitEvent(side, (hipEvent_t)fio->inputs_event, 0));
  }
  if ((rc = pyramid_init(pb, head))) return rc;

  // image branch on its own stream, forked from and later joined to the main one (events 9 / 10), ahead of the pyramid:
  // its ~50 small launches (~0.2 ms as a chain) must be done by the fusion block.  Forking it later in the step -- after
  // block1 / conv2 / block2 / conv3 / block3, beside the levels that leave CUs idle -- was measured slower (1.07 ->
  // 1.15 ... 1.23 ms, round 3): the chain then ends after the encoder and the fusion waits for it.
  const int fork_env = -1;
  FragmentCtx fctx{&pb, fio->serialize ? -1 : fork_env, img, caps, fio, imgs, head_on_side};
  if (fctx.fork_after < 0 && (rc = fork_image_branch(fctx, head))) return rc;

  // level 0 of the pyramid on the head's stream (conv1 needs it first); the coarse levels go to the side stream
  if ((rc = pyramid_level0(pb, head, false))) return rc;

  imf_resunet_io io;
  memset(&io, 0, sizeof(io));
  const size_t per = (size_t)128 * fio->tokens_padded;
  for (int i = 0; i < 4; ++i) {
    io.level[i] = fio->levels[i];
    io.n[i] = caps->rows[i];
  }
  io.x_all_ones = 1;
  io.n_items = caps->n_items;
  for (int b = 0; b < caps->n_items; ++b) {
    io.kt_packed[b] = fio->kt_packed + b * per;
    io.v_packed[b] = fio->v_packed + b * per;
  }
  io.n_tokens = ntok; io.tokens_padded = fio->tokens_padded;
  io.image_ready = fio->events[10];
  io.int_arena = fio->int_arena; io.int_arena_bytes = fio->int_arena_bytes;
  io.float_arena = fio->float_arena; io.float_arena_bytes = fio->float_arena_bytes;
  io.out = fio->out;
  for (int i = 0; i < 9; ++i) io.events[i] = fio->events[i];
  io.events[11] = fio->events[11]; io.events[12] = fio->events[12];   // optional diagnostic marks
  io.side_stream = side; io.main_stream = main;
  io.trace = fio->trace;
  io.fp32_buffers = fio->fp32_buffers;
  io.dyn = 1; io.meta = fio->meta; io.bitgrid_words = caps->bitgrid_words; io.pyramid = &fctx;
  return imf_resunet_forward(net, &io);
}

/* ---- hipGraph helpers: capture a launch sequence once, replay it per fragment ---------------------------- */
int imf_graph_begin_capture(void *stream) {
  IMF_CHECK_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return IMF_OK;
}

int imf_graph_end_capture(void *stream, void **graph_exec_out, int *n_nodes_out) {
  IMF_REQUIRE(graph_exec_out, "imf_graph_end_capture: null pointer");
  hipGraph_t graph = nullptr;
  IMF_CHECK_HIP(hipStreamEndCapture((hipStream_t)stream, &graph));
  IMF_REQUIRE(graph, "imf_graph_end_capture: the capture was invalidated");
  size_t n_nodes = 0;
  (void)hipGraphGetNodes(graph, nullptr, &n_nodes);
  if (n_nodes_out) *n_nodes_out = (int)n_nodes;
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
    return IMF_ELAUNCH;
  }
  *graph_exec_out = (void *)exec;
  return IMF_OK;
}

int imf_graph_abort_capture(void *stream) {   // after a failed enqueue: leave capture mode, drop the partial graph
  hipGraph_t graph = nullptr;
  (void)hipStreamEndCapture((hipStream_t)stream, &graph);
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipGetLastError();
  return IMF_OK;
}

int imf_graph_launch(void *graph_exec, void *stream) {
  IMF_REQUIRE(graph_exec, "imf_graph_launch: null graph");
  IMF_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return IMF_OK;
}

void imf_graph_destroy(void *graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
}

}  // extern "C"
