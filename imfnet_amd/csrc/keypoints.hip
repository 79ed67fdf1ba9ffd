// Keypoint -> voxel selection of the evaluator.
//
// Reference being replaced: scripts/evaluation_3dmatch.py:162-171 --
//   key_points = ME.utils.fnv_hash_vec(np.floor(sample / voxel_size))
//   key_coords = ME.utils.fnv_hash_vec(np.floor(coord  / voxel_size))
//   inds       = np.where(np.isin(key_coords, key_points))[0]
// i.e. the ascending row indices of the voxels (rows of `xyz` in the descriptor file) whose FNV-1a-64
// key occurs among the keys of the sampled raw points.  The device version keeps the reference's
// key (so even an FNV collision selects the same rows): sample keys go into an open-addressing set,
// voxel keys probe it, survivors are compacted in order (count / scan / emit).
#include "common.h"

namespace imf {
namespace {

constexpr int kKpBlock = 1024;

// ME.utils.fnv_hash_vec on one row of np.floor(p / voxel): the float -> uint64 cast of numpy on
// x86-64 wraps negatives two's-complement, so cast through int64.
__device__ __forceinline__ uint64_t fnv_key(const double *__restrict__ p, double voxel) {
  uint64_t h = 14695981039346656037ull;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    h *= 1099511628211ull;
    h ^= (uint64_t)(int64_t)floor(p[j] / voxel);
  }
  return h;
}

__global__ __launch_bounds__(256) void k_kp_init(uint64_t *keys, int64_t cap) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) keys[i] = kEmptyKey;
}

__global__ __launch_bounds__(256) void k_kp_insert(const double *__restrict__ samples, int64_t ns,
                                                   double voxel, uint64_t *keys, uint32_t capmask) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  uint64_t k = fnv_key(samples + 3 * i, voxel);
  if (k == kEmptyKey) k = kEmptyKey - 1;   // 2^-64: keep the sentinel free (probe applies the same map)
  hash_insert_key(keys, capmask, k);
}

__device__ __forceinline__ bool set_contains(const uint64_t *__restrict__ keys, uint32_t capmask, uint64_t key) {
  uint32_t s = hash64(key) & capmask;
  while (true) {
    const uint64_t k = keys[s];
    if (k == key) return true;
    if (k == kEmptyKey) return false;
    s = (s + 1) & capmask;
  }
}

__global__ __launch_bounds__(kKpBlock) void k_kp_probe(const double *__restrict__ coords, int64_t m,
                                                       double voxel, const uint64_t *__restrict__ keys,
                                                       uint32_t capmask, uint8_t *__restrict__ flags,
                                                       int32_t *__restrict__ block_sums) {
  __shared__ int wsum[kKpBlock / 64];
  const int64_t i = (int64_t)blockIdx.x * kKpBlock + threadIdx.x;
  int f = 0;
  if (i < m) {
    uint64_t k = fnv_key(coords + 3 * i, voxel);
    if (k == kEmptyKey) k = kEmptyKey - 1;
    f = set_contains(keys, capmask, k) ? 1 : 0;
    flags[i] = (uint8_t)f;
  }
  const unsigned long long bal = __ballot(f);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(bal);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < kKpBlock / 64; ++w) t += wsum[w];
    block_sums[blockIdx.x] = t;
  }
}

// exclusive scan of the per-block counts in place (one workgroup; nb is a few hundred at most)
__global__ __launch_bounds__(256) void k_kp_scan(int32_t *block_sums, int nb, int32_t *count_out) {
  __shared__ int part[256];
  const int t = threadIdx.x;
  const int per = (nb + 255) / 256;
  const int lo = min(nb, t * per), hi = min(nb, lo + per);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += block_sums[i];
  part[t] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int v = (t >= o) ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;
  for (int i = lo; i < hi; ++i) {
    const int c = block_sums[i];
    block_sums[i] = run;
    run += c;
  }
  if (t == 255) *count_out = part[255];
}

__global__ __launch_bounds__(kKpBlock) void k_kp_emit(const uint8_t *__restrict__ flags, int64_t m,
                                                      const int32_t *__restrict__ block_offs,
                                                      int32_t *__restrict__ inds) {
  __shared__ int wsum[kKpBlock / 64];
  const int64_t i = (int64_t)blockIdx.x * kKpBlock + threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int f = (i < m) ? flags[i] : 0;
  const unsigned long long bal = __ballot(f);
  if (lane == 0) wsum[w] = __popcll(bal);
  __syncthreads();
  if (f) {
    int off = block_offs[blockIdx.x];
    for (int q = 0; q < w; ++q) off += wsum[q];
    inds[off + __popcll(bal & ((1ull << lane) - 1ull))] = (int32_t)i;
  }
}

}  // namespace
}  // namespace imf

using namespace imf;

extern "C" {

size_t imf_keypoint_workspace_bytes(int64_t n_samples, int64_t n_voxels) {
  if (n_samples < 0 || n_voxels < 0) return 0;
  const int64_t cap = imf_hash_capacity(n_samples);
  const int64_t nb = div_up(n_voxels, kKpBlock);
  return (size_t)cap * 8 + (size_t)(nb + 1) * 4 + (size_t)n_voxels + 64;
}

int imf_select_keypoints(const double *samples, int64_t n_samples, const double *coords, int64_t n_voxels,
                         double voxel_size, int32_t *inds, int32_t *count, void *workspace,
                         size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IMF_REQUIRE(n_samples >= 0 && n_voxels >= 0 && n_samples < (1ll << 30) && n_voxels < (1ll << 30),
              "imf_select_keypoints: bad sizes");
  IMF_REQUIRE(voxel_size > 0.0, "imf_select_keypoints: voxel_size must be > 0");
  IMF_REQUIRE(count && workspace && (n_samples == 0 || samples) && (n_voxels == 0 || (coords && inds)),
              "imf_select_keypoints: null pointer");
  IMF_REQUIRE(workspace_bytes >= imf_keypoint_workspace_bytes(n_samples, n_voxels),
              "imf_select_keypoints: workspace %zu < %zu bytes", workspace_bytes,
              imf_keypoint_workspace_bytes(n_samples, n_voxels));
  const int64_t cap = imf_hash_capacity(n_samples);
  const int nb = (int)div_up(n_voxels, kKpBlock);
  uint64_t *keys = (uint64_t *)workspace;
  int32_t *block_sums = (int32_t *)(keys + cap);
  uint8_t *flags = (uint8_t *)(block_sums + nb + 1);
  if (n_voxels == 0) {
    IMF_CHECK_HIP(hipMemsetAsync(count, 0, 4, stream));
    return IMF_OK;
  }
  k_kp_init<<<(unsigned)div_up(cap, 256), 256, 0, stream>>>(keys, cap);
  if (n_samples)
    k_kp_insert<<<(unsigned)div_up(n_samples, 256), 256, 0, stream>>>(samples, n_samples, voxel_size, keys,
                                                                     (uint32_t)(cap - 1));
  k_kp_probe<<<nb, kKpBlock, 0, stream>>>(coords, n_voxels, voxel_size, keys, (uint32_t)(cap - 1), flags,
                                          block_sums);
  k_kp_scan<<<1, 256, 0, stream>>>(block_sums, nb, count);
  k_kp_emit<<<nb, kKpBlock, 0, stream>>>(flags, n_voxels, block_sums, inds);
  IMF_CHECK_LAUNCH("imf_select_keypoints");
  return IMF_OK;
}

}  // extern "C"
