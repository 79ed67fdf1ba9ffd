// A C-level client of libimfnet_hip.so: no Python, no torch -- only include/imfnet_hip.h and the HIP runtime for
// device memory.  It voxelises a synthetic cloud, builds the k=3 rulebook and runs fused sparse convolutions (fp32 MFMA
// variant 0, the default split-f16 variant 6 on k_spconv_g and on the wave-split k_spconv_w, the fused pointwise head)
// with all-ones features and weights, whose exact result is known: out[row][c] = 32 * (number of occupied neighbours).
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
  // the default arithmetic of the model layers: variant 6 (split-f16 operands on the f16 matrix pipe, csrc/spconv_g.hip:
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
  // a bad argument is reported, not executed
  a.cout = 33;
  if (imf_spconv_fwd(&a, nullptr) != IMF_EINVAL || !strstr(imf_last_error(), "cout")) { printf("argument check missing\n"); return 7; }
  printf("C ABI driver OK: %lld points -> %lld voxels, fp32-MFMA / split-f16 (LDS-DMA, wave-split, operand images) convolutions and the pointwise head exact on %lld rows\n", (long long)n, (long long)m, (long long)m);
  return 0;
}
