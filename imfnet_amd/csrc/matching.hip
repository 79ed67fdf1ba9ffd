// Descriptor matching for the feature-match-recall evaluation.
//
// Reference being replaced: scripts/evaluation_3dmatch.py:207-234 -- two `uio.knn_search` calls
// (util/uio.py:245-258: one Open3D KD-tree query per row, fp64, k=1), the mutual check
// `arange(n2) == nn12[nn21]`, the ground-truth transform of the matched frag2 keypoints and the
// `distance < inlier_thresh` count.
//
// The KD-tree is exact, so the device search has to be exact too: scores are formed in fp64 on the
// f64 matrix pipe (v_mfma_f64_16x16x4_f64) as |d|^2 - 2 q.d, the argmin over the database is kept
// per lane in registers (ties -> lowest index, like a first-minimum scan), database splits are
// combined by a second small kernel.  5 000 x 5 000 x 32 is 1.6 GFLOP of fp64 per direction.
#include "common.h"

namespace imf {
namespace {

using f64x4 = __attribute__((ext_vector_type(4))) double;

constexpr int kNnQueriesPerBlock = 64;   // 4 waves x 16 query columns
constexpr int kNnTileRows = 64;          // database rows staged in LDS per iteration

__global__ __launch_bounds__(256) void k_row_norm2(const float *__restrict__ x, int64_t n, int dim,
                                                   double *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *r = x + i * dim;
  double s = 0.0;
  for (int c = 0; c < dim; ++c) s += (double)r[c] * (double)r[c];
  out[i] = s;
}

// One wave = 16 queries (MFMA column = lane & 15) against database rows [d_begin, d_end) of split
// blockIdx.y.  C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg.
template <int D>
__global__ __launch_bounds__(256) void k_nn_search(const float *__restrict__ Q, int nq,
                                                   const float *__restrict__ Db,
                                                   const double *__restrict__ dnorm, int nd,
                                                   int split_len, double *__restrict__ part_best,
                                                   int32_t *__restrict__ part_idx) {
  constexpr int KS = D / 4;
  constexpr int LDW = D + 4;   // row stride in floats: 16-B aligned rows, conflict-free A-operand reads
  __shared__ __attribute__((aligned(16))) float tile[kNnTileRows * LDW];
  __shared__ double tnorm[kNnTileRows];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 15, kq = lane >> 4;
  const int q0 = (blockIdx.x * 4 + wave) * 16;
  const int d_begin = blockIdx.y * split_len;
  const int d_end = min(nd, d_begin + split_len);

  double qf[KS];
  {
    const int qr = min(q0 + col, nq - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = -2.0 * (double)Q[(int64_t)qr * D + 4 * ks + kq];
  }

  double best = __builtin_huge_val();
  int bidx = 0x7fffffff;

  for (int t0 = d_begin; t0 < d_end; t0 += kNnTileRows) {
    __syncthreads();
    // stage 64 rows x D floats (float4 per thread, coalesced); rows past the end are zero
    for (int e = tid; e < kNnTileRows * (D / 4); e += 256) {
      const int r = e / (D / 4), c4 = e % (D / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t0 + r < d_end) v = reinterpret_cast<const float4 *>(Db + (int64_t)(t0 + r) * D)[c4];
      *reinterpret_cast<float4 *>(&tile[r * LDW + 4 * c4]) = v;
    }
    if (tid < kNnTileRows) tnorm[tid] = (t0 + tid < d_end) ? dnorm[t0 + tid] : 0.0;
    __syncthreads();

#pragma unroll
    for (int sub = 0; sub < kNnTileRows / 16; ++sub) {
      f64x4 acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = tnorm[sub * 16 + kq + 4 * r];
      const float *arow = &tile[(sub * 16 + col) * LDW + kq];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)arow[4 * ks], qf[ks], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = t0 + sub * 16 + kq + 4 * r;
        if (idx < d_end && acc[r] < best) {   // idx ascends per lane: strict < keeps the first minimum
          best = acc[r];
          bidx = idx;
        }
      }
    }
  }

  // the 4 lanes {col, col+16, col+32, col+48} hold disjoint rows of the same query
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    const double ob = __shfl_xor(best, off);
    const int oi = __shfl_xor(bidx, off);
    if (ob < best || (ob == best && oi < bidx)) {
      best = ob;
      bidx = oi;
    }
  }
  if (lane < 16 && q0 + col < nq) {
    part_best[(int64_t)blockIdx.y * nq + q0 + col] = best;
    part_idx[(int64_t)blockIdx.y * nq + q0 + col] = bidx;
  }
}

__global__ __launch_bounds__(256) void k_nn_combine(const double *__restrict__ part_best,
                                                    const int32_t *__restrict__ part_idx, int nq,
                                                    int splits, const float *__restrict__ Q, int dim,
                                                    int32_t *__restrict__ nn, double *__restrict__ dist2) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  double best = part_best[q];
  int bidx = part_idx[q];
  for (int s = 1; s < splits; ++s) {   // splits ascend in index: strict < keeps the lowest index
    const double b = part_best[(int64_t)s * nq + q];
    if (b < best) {
      best = b;
      bidx = part_idx[(int64_t)s * nq + q];
    }
  }
  nn[q] = bidx;
  if (dist2) {
    double qn = 0.0;
    for (int c = 0; c < dim; ++c) qn += (double)Q[(int64_t)q * dim + c] * (double)Q[(int64_t)q * dim + c];
    dist2[q] = fmax(best + qn, 0.0);
  }
}

struct Pose {
  double m[16];
};

// Single workgroup: mutual check, ordered compaction of the surviving frag2 indices, transform by
// the ground-truth pose (Open3D PointCloud::Transform: homogeneous multiply then divide by w) and
// the inlier count.  n2 <= a few thousand, so one 1024-thread block scans it in chunks.
__global__ __launch_bounds__(1024) void k_mutual_inliers(const int32_t *__restrict__ nn21, int n2,
                                                         const int32_t *__restrict__ nn12, int n1,
                                                         const double *__restrict__ kp1,
                                                         const double *__restrict__ kp2, Pose T,
                                                         int has_geometry, double thresh,
                                                         int32_t *__restrict__ match2,
                                                         int32_t *__restrict__ meta) {
  __shared__ int wave_sum[16];
  __shared__ int base_s, inl_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    base_s = 0;
    inl_s = 0;
  }
  __syncthreads();
  for (int c0 = 0; c0 < n2; c0 += 1024) {
    const int j = c0 + tid;
    int flag = 0, i = 0;
    if (j < n2) {
      i = nn21[j];
      flag = (i >= 0 && i < n1 && nn12[i] == j) ? 1 : 0;
    }
    const unsigned long long bal = __ballot(flag);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_sum[wave] = __popcll(bal);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wave_sum[w];
    if (flag) {
      match2[off + before] = j;
      if (has_geometry) {
        const double x = kp2[3 * j], y = kp2[3 * j + 1], z = kp2[3 * j + 2];
        const double w = T.m[12] * x + T.m[13] * y + T.m[14] * z + T.m[15];
        const double px = (T.m[0] * x + T.m[1] * y + T.m[2] * z + T.m[3]) / w;
        const double py = (T.m[4] * x + T.m[5] * y + T.m[6] * z + T.m[7]) / w;
        const double pz = (T.m[8] * x + T.m[9] * y + T.m[10] * z + T.m[11]) / w;
        const double dx = kp1[3 * i] - px, dy = kp1[3 * i + 1] - py, dz = kp1[3 * i + 2] - pz;
        if (sqrt(dx * dx + dy * dy + dz * dz) < thresh) atomicAdd(&inl_s, 1);
      }
    }
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += wave_sum[w];
      base_s += tot;
    }
    __syncthreads();
  }
  if (tid == 0) {
    meta[0] = base_s;
    meta[1] = inl_s;
  }
}

int nn_splits(int64_t nq, int64_t nd) {
  // enough workgroups for 256 CUs x 2, split length a multiple of the LDS tile
  const int64_t blocks = nq > 0 ? div_up(nq, kNnQueriesPerBlock) : 1;
  int64_t s = div_up(512, blocks);
  const int64_t max_s = div_up(nd, kNnTileRows);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (int)s;
}

}  // namespace
}  // namespace imf

using namespace imf;

extern "C" {

size_t imf_nn_workspace_bytes(int64_t n_query, int64_t n_db) {
  if (n_query < 0 || n_db < 0) return 0;
  const int s = nn_splits(n_query, n_db);
  return (size_t)n_db * 8 + (size_t)s * n_query * (8 + 4) + 64;
}

int imf_nn_search(const float *query, int64_t n_query, const float *db, int64_t n_db, int dim,
                  int32_t *nn_index, double *nn_dist2, void *workspace, size_t workspace_bytes,
                  void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IMF_REQUIRE(dim == 16 || dim == 32 || dim == 64, "imf_nn_search: dim %d not in {16,32,64}", dim);
  IMF_REQUIRE(n_query >= 0 && n_db >= 1 && n_query < (1ll << 30) && n_db < (1ll << 30),
              "imf_nn_search: n_query=%lld n_db=%lld (an empty database has no nearest neighbour)",
              (long long)n_query, (long long)n_db);
  if (n_query == 0) return IMF_OK;
  IMF_REQUIRE(query && db && nn_index && workspace, "imf_nn_search: null pointer");
  IMF_REQUIRE(workspace_bytes >= imf_nn_workspace_bytes(n_query, n_db),
              "imf_nn_search: workspace %zu < %zu bytes", workspace_bytes,
              imf_nn_workspace_bytes(n_query, n_db));
  const int nq = (int)n_query, nd = (int)n_db;
  const int splits = nn_splits(n_query, n_db);
  const int split_len = (int)(div_up(div_up(n_db, splits), kNnTileRows) * kNnTileRows);
  double *dnorm = (double *)workspace;
  double *part_best = dnorm + n_db;
  int32_t *part_idx = (int32_t *)(part_best + (size_t)splits * n_query);

  k_row_norm2<<<(unsigned)div_up(n_db, 256), 256, 0, stream>>>(db, n_db, dim, dnorm);
  dim3 grid((unsigned)div_up(n_query, kNnQueriesPerBlock), (unsigned)splits);
  if (dim == 16)
    k_nn_search<16><<<grid, 256, 0, stream>>>(query, nq, db, dnorm, nd, split_len, part_best, part_idx);
  else if (dim == 32)
    k_nn_search<32><<<grid, 256, 0, stream>>>(query, nq, db, dnorm, nd, split_len, part_best, part_idx);
  else
    k_nn_search<64><<<grid, 256, 0, stream>>>(query, nq, db, dnorm, nd, split_len, part_best, part_idx);
  k_nn_combine<<<(unsigned)div_up(n_query, 256), 256, 0, stream>>>(part_best, part_idx, nq, splits, query,
                                                                  dim, nn_index, nn_dist2);
  IMF_CHECK_LAUNCH("imf_nn_search");
  return IMF_OK;
}

int imf_mutual_inliers(const int32_t *nn21, int64_t n2, const int32_t *nn12, int64_t n1,
                       const double *kpts1, const double *kpts2, const double *pose_host,
                       double inlier_thresh, int32_t *match_idx2, int32_t *meta, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IMF_REQUIRE(n1 >= 0 && n2 >= 0 && n1 < (1ll << 30) && n2 < (1ll << 30), "imf_mutual_inliers: bad sizes");
  IMF_REQUIRE(meta && (n2 == 0 || (nn21 && nn12 && match_idx2)), "imf_mutual_inliers: null pointer");
  const int has_geometry = (kpts1 && kpts2 && pose_host) ? 1 : 0;
  Pose T;
  for (int i = 0; i < 16; ++i) T.m[i] = has_geometry ? pose_host[i] : (i % 5 == 0 ? 1.0 : 0.0);
  k_mutual_inliers<<<1, 1024, 0, stream>>>(nn21, (int)n2, nn12, (int)n1, kpts1, kpts2, T, has_geometry,
                                           inlier_thresh, match_idx2, meta);
  IMF_CHECK_LAUNCH("imf_mutual_inliers");
  return IMF_OK;
}

}  // extern "C"
