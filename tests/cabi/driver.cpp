// A C-level client of libimfnet_hip.so: no Python, no torch -- only include/imfnet_hip.h and the HIP runtime for
// device memory.  It voxelises a synthetic cloud, builds the k=3 rulebook and runs fused sparse convolutions -- fp32 MFMA
// (variant 0), the DEFAULT arithmetic bf16x3 (variant 3: imf_pack_weights_bf16x3 + imf_spconv_fwd on k_spconv_g and on every
// workgroup shape of the wave-split k_spconv_w), the split-f16 fast mode (variant 6, incl. operand images and the fused
// pointwise head) -- with all-ones features and weights, whose exact result is known: out[row][c] = 32 * (number of occupied
// neighbours).  Then the boundary the product uses: ONE imf_fragment_forward call (points + image -> descriptors, capacity
// mode, three streams) on a network built from C with constant weights, whose answer is known as well -- every channel of
// every layer carries the same value, so each L2-normalised 32-D descriptor is 1 / sqrt(32) in every component.
// Exit code 0 = every check passed.  Built by __graft_entry__.build(), run by tests/test_gpu_cabi_driver.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <set>
#include <tuple>
#include <array>
#include <map>
#include <vector>

#include "imfnet_hip.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define IMF(x) do { int rc_ = (x); if (rc_ != 0) { printf("%s failed: %s\n", #x, imf_last_error()); return 3; } } while (0)

template <typename T> static T *dev(size_t n) { T *p = nullptr; if (hipMalloc((void **)&p, n * sizeof(T) + 256) != hipSuccess) return nullptr; return p; }

int main() {
  printf("imf_version %d\n", imf_version());
  // a 20 x 20 x 3 slab of voxels at 5 cm, 4 points per voxel (duplicates exercise the first-occurrence rule)
  const double voxel = 0.05;
  std::vector<double> xyz;
  for (int rep = 0; rep < 4; ++rep)
    for (int x = 0; x < 20; ++x) for (int y = 0; y < 20; ++y) for (int z = 0; z < 3; ++z) {
      xyz.push_back((x - 10 + 0.1 + 0.2 * rep) * voxel); xyz.push_back((y - 10 + 0.3) * voxel); xyz.push_back((z + 0.5) * voxel);
    }
  const int64_t n = (int64_t)xyz.size() / 3;
  const int64_t cap = imf_hash_capacity(n);
  double *d_xyz = dev<double>(3 * n);
  int32_t *coords = dev<int32_t>(4 * n), *first = dev<int32_t>(n), *meta = dev<int32_t>(2);
  imf_slot *table = dev<imf_slot>(cap);
  void *ws = dev<char>(imf_unique_workspace_bytes(n));
  HIP(hipMemcpy(d_xyz, xyz.data(), xyz.size() * 8, hipMemcpyHostToDevice));
  HIP(hipMemset(meta, 0, 8));
  IMF(imf_voxelize(d_xyz, 1, n, voxel, 0, coords, first, meta, table, cap, ws, meta + 1, nullptr));
  int32_t h_meta[2];
  HIP(hipMemcpy(h_meta, meta, 8, hipMemcpyDeviceToHost));
  const int64_t m = h_meta[0];
  if (m != 1200 || h_meta[1] != 0) { printf("voxel count %lld (expected 1200), err %d\n", (long long)m, h_meta[1]); return 4; }
  std::vector<int32_t> h_coords(4 * m), h_first(m);
  HIP(hipMemcpy(h_coords.data(), coords, 16 * m, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(h_first.data(), first, 4 * m, hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < m; ++i)
    if (h_first[i] != i) { printf("first-occurrence order broken at row %lld\n", (long long)i); return 5; }   // rep 0 comes first
  // rulebook of the 3x3x3 stride-1 convolution + one fused convolution (variant 0: fp32 MFMA, exact on small integers)
  const int64_t slots = imf_rulebook_slots(m);
  int32_t *tile_rows = dev<int32_t>(slots), *nbr = dev<int32_t>(27 * slots);
  uint32_t *mask = dev<uint32_t>(slots / IMF_TILE_ROWS * IMF_MASK_WORDS);
  IMF(imf_rulebook_conv(table, cap, coords, m, 1, 3, tile_rows, nbr, mask, nullptr));
  const int cin = 32, cout = 32;
  std::vector<float> ones((size_t)27 * cin * cout, 1.f), feat((size_t)m * cin, 1.f);
  float *d_w = dev<float>(ones.size()), *d_wp = dev<float>(imf_packed_weight_floats(27, cin, cout)), *d_f = dev<float>(feat.size()),
        *d_out = dev<float>((size_t)m * cout);
  HIP(hipMemcpy(d_w, ones.data(), ones.size() * 4, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_f, feat.data(), feat.size() * 4, hipMemcpyHostToDevice));
  IMF(imf_pack_weights(d_w, 27, cin, cout, d_wp, nullptr));
  imf_conv_args a;
  memset(&a, 0, sizeof(a));
  a.in_a = d_f; a.c_a = cin; a.w_packed = d_wp; a.kvol = 27; a.cout = cout;
  a.tile_rows = tile_rows; a.nbr = nbr; a.tile_mask = mask; a.n_slots = slots; a.n_out = m;
  a.relu = 1; a.out = d_out; a.split_k = 1; a.variant = 0;
  IMF(imf_spconv_fwd(&a, nullptr));
  HIP(hipDeviceSynchronize());
  std::vector<float> out((size_t)m * cout);
  HIP(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
  std::set<std::tuple<int, int, int>> occ;
  for (int64_t i = 0; i < m; ++i) occ.insert({h_coords[4 * i + 1], h_coords[4 * i + 2], h_coords[4 * i + 3]});
  for (int64_t i = 0; i < m; ++i) {
    int cnt = 0;
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx)
      cnt += (int)occ.count({h_coords[4 * i + 1] + dx, h_coords[4 * i + 2] + dy, h_coords[4 * i + 3] + dz});
    for (int c = 0; c < cout; ++c)
      if (out[i * cout + c] != (float)(cnt * cin)) { printf("row %lld col %d: %g != %d\n", (long long)i, c, out[i * cout + c], cnt * cin); return 6; }
  }
  // the FAST mode of the model layers: variant 6 (split-f16 operands on the f16 matrix pipe, csrc/spconv_g.hip:
  // both operands global -> LDS by DMA).  Ones are exact in f16 and the weight image's power-of-two pre-scale is undone
  // exactly, so the same integers must come out.  Then the wave-split kernel of the coarse levels (csrc/spconv_w.hip,
  // kernel_tag 4 / 8) on a 64-column layer, and the fused pointwise head.
  {
    float *d_wp6 = dev<float>(imf_packed_weight_floats_split16(27, cin, cout));
    int32_t *d_flag = dev<int32_t>(1);
    HIP(hipMemset(d_flag, 0, 4));
    IMF(imf_pack_weights_split16(d_w, 27, cin, cout, d_wp6, nullptr));
    imf_conv_args b = a;
    b.w_packed = d_wp6; b.variant = 6; b.dyn_err = d_flag;
    HIP(hipMemset(d_out, 0, out.size() * 4));
    IMF(imf_spconv_fwd(&b, nullptr));
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < m; ++i) {
      int cnt = 0;
      for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx)
        cnt += (int)occ.count({h_coords[4 * i + 1] + dx, h_coords[4 * i + 2] + dy, h_coords[4 * i + 3] + dz});
      for (int c = 0; c < cout; ++c)
        if (out[i * cout + c] != (float)(cnt * cin)) { printf("variant 6: row %lld col %d: %g != %d\n", (long long)i, c, out[i * cout + c], cnt * cin); return 8; }
    }
    if (cin == cout) {
      // operand images (imf_conv_args.operand_format): the layer above written as a split-f16 image, then a second layer fed
      // that image and fed the fp32 rows -- same output, and the exact integers: out2[i] = cin * sum over i's neighbours of out1
      float *d_img = dev<float>(out.size()), *d_o2a = dev<float>(out.size()), *d_o2b = dev<float>(out.size());
      imf_conv_args p1 = b, p2 = b, p3 = b;
      p1.out = d_img; p1.operand_format = IMF_FMT_OUT_SPLIT;
      p2.in_a = d_img; p2.out = d_o2a; p2.operand_format = IMF_FMT_A_SPLIT;
      p3.in_a = d_out; p3.out = d_o2b;
      p2.dyn_err = p3.dyn_err = nullptr;   // (their sums leave the f16 range: nothing reads them as operands)
      IMF(imf_spconv_fwd(&p1, nullptr));
      IMF(imf_spconv_fwd(&p2, nullptr));
      IMF(imf_spconv_fwd(&p3, nullptr));
      HIP(hipDeviceSynchronize());
      std::vector<float> o2a(out.size()), o2b(out.size());
      HIP(hipMemcpy(o2a.data(), d_o2a, out.size() * 4, hipMemcpyDeviceToHost));
      HIP(hipMemcpy(o2b.data(), d_o2b, out.size() * 4, hipMemcpyDeviceToHost));
      std::map<std::array<int, 3>, int64_t> row_of;
      for (int64_t i = 0; i < m; ++i) row_of[{h_coords[4 * i + 1], h_coords[4 * i + 2], h_coords[4 * i + 3]}] = i;
      for (int64_t i = 0; i < m; ++i) {
        double want = 0;
        for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
          auto it = row_of.find({h_coords[4 * i + 1] + dx, h_coords[4 * i + 2] + dy, h_coords[4 * i + 3] + dz});
          if (it != row_of.end()) want += (double)out[it->second * cout] * cin;
        }
        for (int c = 0; c < cout; ++c)
          if (o2a[i * cout + c] != o2b[i * cout + c] || o2a[i * cout + c] != (float)want) {
            printf("operand image: row %lld col %d: image-fed %g, fp32-fed %g, expected %g\n", (long long)i, c, o2a[i * cout + c], o2b[i * cout + c], want);
            return 12;
          }
      }
    }
    const int c64 = 64;
    std::vector<float> ones64((size_t)27 * cin * c64, 1.f);
    float *d_w64 = dev<float>(ones64.size()), *d_wp64 = dev<float>(imf_packed_weight_floats_split16(27, cin, c64)),
          *d_out64 = dev<float>((size_t)m * c64);
    HIP(hipMemcpy(d_w64, ones64.data(), ones64.size() * 4, hipMemcpyHostToDevice));
    IMF(imf_pack_weights_split16(d_w64, 27, cin, c64, d_wp64, nullptr));
    std::vector<float> o64((size_t)m * c64);
    for (int tag : {4, 8}) {
      imf_conv_args w = b;
      w.w_packed = d_wp64; w.cout = c64; w.out = d_out64; w.kernel_tag = tag;
      HIP(hipMemset(d_out64, 0, o64.size() * 4));
      IMF(imf_spconv_fwd(&w, nullptr));
      HIP(hipDeviceSynchronize());
      HIP(hipMemcpy(o64.data(), d_out64, o64.size() * 4, hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < m; ++i)
        for (int c = 0; c < c64; ++c)
          if (o64[i * c64 + c] != out[i * cout]) { printf("wave-split kernel (tag %d): row %lld col %d: %g != %g\n", tag, (long long)i, c, o64[i * c64 + c], out[i * cout]); return 9; }
    }
    // pointwise head: cat([m,64] ones, [m,32] ones) @ ones[96,64] -> relu -> @ ones[64,32] = 96 * 64, exactly
    std::vector<float> w1((size_t)96 * 64, 1.f), w2((size_t)64 * 32, 1.f), fa((size_t)m * 64, 1.f);
    float *d_w1 = dev<float>(w1.size()), *d_w2 = dev<float>(w2.size()), *d_fa = dev<float>(fa.size()),
          *d_w1p = dev<float>(imf_packed_weight_floats_split16(1, 96, 64)), *d_w2p = dev<float>(imf_packed_weight_floats_split16(1, 64, 32));
    HIP(hipMemcpy(d_w1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_w2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_fa, fa.data(), fa.size() * 4, hipMemcpyHostToDevice));
    IMF(imf_pack_weights_split16(d_w1, 1, 96, 64, d_w1p, nullptr));
    IMF(imf_pack_weights_split16(d_w2, 1, 64, 32, d_w2p, nullptr));
    imf_head_args h;
    memset(&h, 0, sizeof(h));
    h.in_a = d_fa; h.c_a = 64; h.in_b = d_f; h.c_b = 32; h.w1_packed = d_w1p; h.relu1 = 1; h.c_mid = 64;
    h.w2_packed = d_w2p; h.c_out = 32; h.l2norm = 0; h.n = m; h.out = d_out; h.flags = d_flag;
    IMF(imf_pointwise_head(&h, nullptr));
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < out.size(); ++i)
      if (out[i] != 6144.f) { printf("pointwise head: element %zu = %g != 6144\n", i, out[i]); return 10; }
    int32_t h_flag = -1;
    HIP(hipMemcpy(&h_flag, d_flag, 4, hipMemcpyDeviceToHost));
    if (h_flag != 0) { printf("range flag raised on in-range data: %d\n", h_flag); return 11; }
  }
  // ---- the DEFAULT arithmetic: variant 3 (bf16x3: every fp32 operand as three bf16 parts, exact; six bf16 MFMAs per 32
  // channels).  Small integers are exact in bf16, so the same integers must come out -- on k_spconv_g (32-column layer) and on
  // each workgroup shape of the wave-split kernel (64-column layer: kernel_tag 8 = 4 wavefronts, 8 | 64 = half tiles,
  // 4 = 8 wavefronts, 4 | 128 = 48-row units), over the map as built and over its occupancy-sorted twin.
  {
    float *d_wp3 = dev<float>(imf_packed_weight_floats_bf16x3(27, cin, cout));
    IMF(imf_pack_weights_bf16x3(d_w, 27, cin, cout, d_wp3, nullptr));
    imf_conv_args c3 = a;
    c3.w_packed = d_wp3; c3.variant = 3;
    HIP(hipMemset(d_out, 0, out.size() * 4));
    IMF(imf_spconv_fwd(&c3, nullptr));
    HIP(hipDeviceSynchronize());
    HIP(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    std::vector<float> want(m);
    for (int64_t i = 0; i < m; ++i) {
      int cnt = 0;
      for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx)
        cnt += (int)occ.count({h_coords[4 * i + 1] + dx, h_coords[4 * i + 2] + dy, h_coords[4 * i + 3] + dz});
      want[i] = (float)(cnt * cin);
      for (int c = 0; c < cout; ++c)
        if (out[i * cout + c] != want[i]) { printf("variant 3: row %lld col %d: %g != %g\n", (long long)i, c, out[i * cout + c], want[i]); return 13; }
    }
    // the occupancy-sorted twin of the map (imf_rulebook_sort_by_occupancy): same rows, other slots
    int32_t *s_rows = dev<int32_t>(slots), *s_nbr = dev<int32_t>(27 * slots);
    uint32_t *s_mask = dev<uint32_t>(slots / IMF_TILE_ROWS * IMF_MASK_WORDS);
    const size_t sws = imf_rulebook_sorted_workspace_bytes(slots);
    void *d_sws = dev<char>(sws);
    IMF(imf_rulebook_sort_by_occupancy(nbr, 27, slots, m, nullptr, s_rows, s_nbr, s_mask, d_sws, sws, nullptr));
    std::vector<int32_t> h_rows(slots);
    HIP(hipMemcpy(h_rows.data(), s_rows, slots * 4, hipMemcpyDeviceToHost));
    std::vector<char> seen(m, 0);
    for (int64_t q = 0; q < slots; ++q)
      if (h_rows[q] >= 0) { if (h_rows[q] >= m || seen[h_rows[q]]) { printf("sorted map: slot %lld holds row %d twice / out of range\n", (long long)q, h_rows[q]); return 14; } seen[h_rows[q]] = 1; }
    for (int64_t i = 0; i < m; ++i) if (!seen[i]) { printf("sorted map: row %lld missing\n", (long long)i); return 14; }
    const int c64 = 64;
    std::vector<float> ones64((size_t)27 * cin * c64, 1.f), o64((size_t)m * c64);
    float *d_w64 = dev<float>(ones64.size()), *d_wp64 = dev<float>(imf_packed_weight_floats_bf16x3(27, cin, c64)), *d_out64 = dev<float>(o64.size());
    HIP(hipMemcpy(d_w64, ones64.data(), ones64.size() * 4, hipMemcpyHostToDevice));
    IMF(imf_pack_weights_bf16x3(d_w64, 27, cin, c64, d_wp64, nullptr));
    for (int sorted = 0; sorted < 2; ++sorted)
      for (int tag : {8, 8 | 64, 4, 4 | 128}) {
        imf_conv_args w3 = c3;
        w3.w_packed = d_wp64; w3.cout = c64; w3.out = d_out64; w3.kernel_tag = tag;
        if (sorted) { w3.tile_rows = s_rows; w3.nbr = s_nbr; w3.tile_mask = s_mask; }
        HIP(hipMemset(d_out64, 0, o64.size() * 4));
        IMF(imf_spconv_fwd(&w3, nullptr));
        HIP(hipDeviceSynchronize());
        HIP(hipMemcpy(o64.data(), d_out64, o64.size() * 4, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < m; ++i)
          for (int c = 0; c < c64; ++c)
            if (o64[i * c64 + c] != want[i]) { printf("variant 3, wave-split tag %d, %s map: row %lld col %d: %g != %g\n", tag, sorted ? "sorted" : "plain", (long long)i, c, o64[i * c64 + c], want[i]); return 15; }
      }
  }

  // ---- imf_fragment_forward: points + image -> descriptors in ONE call (the boundary extract_features uses, util/misc.py:67-104
  // -> model/resunet.py:163-235), on a ResUNetBN2C-shaped network built here with CONSTANT weights per tensor (default arithmetic,
  // variant 3).  With every weight of a tensor equal, all channels of a layer carry the same value at every row (LayerNorm
  // of a constant row is its bias, attention over equal scores is a mean, ...), all of them positive; the head's L2 norm then
  // makes every component of every descriptor 1 / sqrt(32), whatever the geometry -- and the row counts of the four levels
  // are those of the slab (x, y in -10 .. 9, z in 0 .. 2): 1200 / 200 / 36 / 16.
  {
    auto fill = [&](size_t nf, float v) -> float * {
      std::vector<float> h(nf, v);
      float *d = dev<float>(nf);
      if (d) (void)hipMemcpy(d, h.data(), nf * 4, hipMemcpyHostToDevice);
      return d;
    };
    auto pack3 = [&](int kvol, int ci, int co, float v) -> float * {   // constant [kvol][ci][co] kernel as a bf16x3 image
      float *raw = fill((size_t)kvol * ci * co, v), *img = dev<float>(imf_packed_weight_floats_bf16x3(kvol, ci, co));
      if (!raw || !img || imf_pack_weights_bf16x3(raw, kvol, ci, co, img, nullptr) != 0) return nullptr;
      return img;
    };
    auto pack0 = [&](int kvol, int ci, int co, float v) -> float * {   // ... as the fp32 image
      float *raw = fill((size_t)kvol * ci * co, v), *img = dev<float>(imf_packed_weight_floats(kvol, ci, co));
      if (!raw || !img || imf_pack_weights(raw, kvol, ci, co, img, nullptr) != 0) return nullptr;
      return img;
    };
    imf_resunet_desc net;
    memset(&net, 0, sizeof(net));
    const int CH[5] = {0, 32, 64, 128, 256}, TR[5] = {0, 64, 64, 64, 128};
    for (int i = 1; i < 5; ++i) { net.channels[i] = CH[i]; net.tr_channels[i] = TR[i]; }
    net.in_channels = 1; net.out_channels = 32; net.first_ksize = 5; net.small_first = 1;
    struct L { int idx, kvol, ci, co, relu; };
    const L layers[] = {{1, 27, 32, 32, 1}, {2, 27, 32, 32, 1}, {3, 27, 32, 64, 0}, {4, 27, 64, 64, 1}, {5, 27, 64, 64, 1},
                        {6, 27, 64, 128, 0}, {7, 27, 128, 128, 1}, {8, 27, 128, 128, 1}, {9, 27, 128, 256, 0},
                        {10, 27, 256, 256, 1}, {11, 27, 256, 256, 1}, {12, 27, 256, 128, 0}, {13, 27, 128, 128, 1},
                        {14, 27, 128, 128, 1}, {15, 27, 256, 64, 0}, {16, 27, 64, 64, 1}, {17, 27, 64, 64, 1},
                        {18, 27, 128, 64, 0}, {19, 27, 64, 64, 1}, {20, 27, 64, 64, 1}, {21, 1, 96, 64, 1}, {22, 1, 64, 32, 0}};
    for (const L &l : layers) {
      imf_net_conv &c = net.conv[l.idx];
      c.w_packed = pack3(l.kvol, l.ci, l.co, 1.f / (float)(l.kvol * l.ci));
      c.kvol = l.kvol; c.cin = l.ci; c.cout = l.co; c.relu = l.relu; c.variant = 3;
      c.scale = l.idx >= 21 ? nullptr : fill(l.co, 1.f);
      c.shift = l.idx == 21 ? nullptr : fill(l.co, 0.01f);
      c.l2norm = l.idx == 22;
      if (!c.w_packed) { printf("fragment forward: packing layer %d failed: %s\n", l.idx, imf_last_error()); return 20; }
    }
    net.first_kernel = fill((size_t)125 * 32, 1.f / 125.f);
    net.first_scale = fill(32, 1.f); net.first_shift = fill(32, 0.01f);
    float *first_img = dev<float>(imf_first_kernel_image_floats(125, 32));
    IMF(imf_pack_first_kernel(net.first_kernel, 125, 32, first_img, nullptr));
    net.first_kernel_image = first_img;
    imf_fusion_weights &fw = net.fusion;
    fw.ln1_g = fill(256, 1.f); fw.ln1_b = fill(256, 0.5f);
    fw.wq_p = pack0(1, 256, 128, 1.f / 256.f); fw.wo_p = pack0(1, 128, 256, 1.f / 128.f); fw.bo = fill(256, 0.1f);
    fw.ln2_g = fill(256, 1.f); fw.ln2_b = fill(256, 0.5f);
    fw.w1_p = pack3(1, 256, 2048, 1.f / 256.f); fw.b1 = fill(2048, 0.1f);
    fw.w2_p = pack3(1, 1024, 256, 1.f / 1024.f); fw.b2 = fill(256, 0.1f);
    fw.w1_f32 = pack0(1, 256, 2048, 1.f / 256.f); fw.w2_f32 = pack0(1, 1024, 256, 1.f / 1024.f);
    net.fusion_scale = 0.08838834764831845f;           // 128^-0.5 (model/attention_fusion.py:70)
    imf_image_desc img;
    memset(&img, 0, sizeof(img));
    img.variant = 3;
    img.stem_w = pack3(1, 160, 64, 1.f / 147.f); img.stem_scale = fill(64, 1.f); img.stem_shift = fill(64, 0.01f);
    for (int i = 0; i < 15; ++i) {
      const int kv = i == 7 ? 1 : 9, ci = i <= 7 ? 64 : 128, co = i <= 5 ? 64 : 128;
      imf_net_conv &c = img.conv[i];
      c.w_packed = pack3(kv, ci, co, 1.f / (float)(kv * ci));
      c.kvol = kv; c.cin = ci; c.cout = co; c.scale = fill(co, 1.f); c.shift = fill(co, 0.01f);
      c.relu = i != 7; c.l2norm = 0; c.variant = 3;
    }
    img.ln_g = fill(128, 1.f); img.ln_b = fill(128, 0.5f); img.kv_w = pack3(1, 128, 256, 1.f / 128.f);

    const int H = 120, W = 160;
    imf_fragment_caps caps;
    memset(&caps, 0, sizeof(caps));
    caps.n_points = n + 512; caps.rows[0] = 1536; caps.rows[1] = 320; caps.rows[2] = 64; caps.rows[3] = 64;
    caps.n_items = 1; caps.img_h = H; caps.img_w = W;
    const int32_t box[8] = {0, -16, -16, -4, 0, 16, 16, 8};
    caps.bitgrid_words = (imf_bitgrid_words(box, 5) + 3) / 4 * 4;      // (a multiple of 4 words: the grid is cleared 16 bytes at a time)
    if (!caps.bitgrid_words) { printf("fragment forward: bit grid size query failed\n"); return 21; }
    imf_fragment_io io;
    memset(&io, 0, sizeof(io));
    double *f_xyz = dev<double>(3 * caps.n_points);
    HIP(hipMemcpy(f_xyz, xyz.data(), xyz.size() * 8, hipMemcpyHostToDevice));
    io.xyz = f_xyz; io.xyz_is_f64 = 1; io.voxel_size = voxel;
    int32_t h_dyn[IMF_DYN_WORDS] = {0};
    h_dyn[0] = (int32_t)n; h_dyn[1] = 1; h_dyn[2] = 0;
    int32_t *d_dyn = dev<int32_t>(IMF_DYN_WORDS), *d_meta = dev<int32_t>(IMF_META_WORDS);
    HIP(hipMemcpy(d_dyn, h_dyn, sizeof(h_dyn), hipMemcpyHostToDevice));
    io.dyn = d_dyn; io.meta = d_meta;
    std::vector<float> h_img((size_t)3 * H * W);
    for (size_t i = 0; i < h_img.size(); ++i) h_img[i] = 0.25f + 0.5f * (float)((i * 2654435761u) >> 24) / 255.f;   // any image in [0, 1]
    float *d_img = dev<float>(h_img.size());
    HIP(hipMemcpy(d_img, h_img.data(), h_img.size() * 4, hipMemcpyHostToDevice));
    io.image = d_img;
    io.pyramid_arena_bytes = imf_fragment_pyramid_bytes(&caps);
    io.pyramid_arena = dev<char>(io.pyramid_arena_bytes + 256);
    io.pyramid_arena = (void *)(((uintptr_t)io.pyramid_arena + 255) & ~(uintptr_t)255);
    io.image_ws_bytes = imf_image_workspace_bytes(1, H, W);
    io.image_ws = dev<char>(io.image_ws_bytes);
    IMF(imf_image_tables_build(1, H, W, io.image_ws, io.image_ws_bytes, nullptr));
    const int tokens = imf_image_tokens(H, W);
    io.tokens_padded = (tokens + 63) / 64 * 64;
    io.kt_packed = dev<float>((size_t)128 * io.tokens_padded); io.v_packed = dev<float>((size_t)128 * io.tokens_padded);
    io.int_arena_bytes = imf_resunet_int_arena_bytes_cap(&net, caps.rows, caps.bitgrid_words);
    io.int_arena = dev<char>(io.int_arena_bytes);
    io.float_arena_bytes = imf_resunet_float_arena_bytes_cap(&net, caps.rows);
    io.float_arena = dev<char>(io.float_arena_bytes);
    float *f_out = dev<float>((size_t)caps.rows[0] * 32);
    io.out = f_out;
    for (int i = 0; i < 11; ++i) io.events[i] = imf_event_create();
    io.main_stream = imf_stream_create(); io.side_stream = imf_stream_create(); io.image_stream = imf_stream_create();
    if (!io.pyramid_arena || !io.image_ws || !io.int_arena || !io.float_arena || !f_out || !io.main_stream || !io.side_stream || !io.image_stream) {
      printf("fragment forward: allocation failed\n"); return 22;
    }
    HIP(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {                 // twice: a bucket is reused for every fragment that fits it
      HIP(hipMemset(f_out, 0xFF, (size_t)caps.rows[0] * 32 * 4));
      IMF(imf_fragment_forward(&net, &img, &caps, &io));
      HIP(hipDeviceSynchronize());
      int32_t h_m[IMF_META_WORDS];
      HIP(hipMemcpy(h_m, d_meta, sizeof(h_m), hipMemcpyDeviceToHost));
      if (h_m[0] != 1200 || h_m[2] != 200 || h_m[4] != 36 || h_m[6] != 16 || h_m[1] != 0) {
        printf("fragment forward: rows %d / %d / %d / %d (expected 1200 / 200 / 36 / 16), flags %d\n", h_m[0], h_m[2], h_m[4], h_m[6], h_m[1]);
        return 23;
      }
      std::vector<float> F((size_t)1200 * 32);
      HIP(hipMemcpy(F.data(), f_out, F.size() * 4, hipMemcpyDeviceToHost));
      const float want = 0.17677669529663687f;          // 1 / sqrt(32)
      for (size_t i = 0; i < F.size(); ++i)
        if (!(F[i] > want - 1e-6f && F[i] < want + 1e-6f)) { printf("fragment forward (call %d): descriptor %zu component %zu = %.9g, expected %.9g\n", rep, i / 32, i % 32, F[i], want); return 24; }
      std::vector<int32_t> h_first(1200);
      HIP(hipMemcpy(h_first.data(), io.levels[0].first_idx, 1200 * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < 1200; ++i) if (h_first[i] != i) { printf("fragment forward: first-occurrence index %d = %d\n", i, h_first[i]); return 25; }
    }
    printf("imf_fragment_forward OK: 4800 points + 120x160 image -> 1200 descriptors, every component 1 / sqrt(32) (constant-weight network, variant 3)\n");
  }

  // a bad argument is reported, not executed
  a.cout = 33;
  if (imf_spconv_fwd(&a, nullptr) != IMF_EINVAL || !strstr(imf_last_error(), "cout")) { printf("argument check missing\n"); return 7; }
  printf("C ABI driver OK: %lld points -> %lld voxels, fp32-MFMA / bf16x3 / split-f16 (LDS-DMA, wave-split, sorted map, operand images) convolutions and the pointwise head exact on %lld rows; one fragment forward from C\n", (long long)n, (long long)m, (long long)m);
  return 0;
}
