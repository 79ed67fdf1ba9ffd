// RANSAC registration on feature correspondences.
//
// Reference being replaced: scripts/benchmark_util.py:16-34 `run_ransac` = Open3D 0.12
// `registration_ransac_based_on_feature_matching(source, target, feat_s, feat_t,
// max_correspondence_distance = 1.5 voxel, TransformationEstimationPointToPoint(False), ransac_n,
// [CorrespondenceCheckerBasedOnEdgeLength(0.9), CorrespondenceCheckerBasedOnDistance(1.5 voxel)],
// RANSACConvergenceCriteria(50000, 1000), mutual_filter=False)`: correspondences = nearest target
// feature of every source point (imf_nn_search); then max_iteration hypotheses, each from ransac_n
// random correspondences: rigid fit (Umeyama without scale), edge-length and distance checkers, score
// = number of correspondences within max_correspondence_distance (ties: lower RMSE, then the earlier
// hypothesis -- the reference's sequential `IsBetterRANSACThan` keeps the first).  The criteria's second
// argument (1000) is outside Open3D 0.12's confidence range, so the loop never terminates early: all
// max_iteration hypotheses are drawn.  Open3D seeds its generator from std::random_device, so the
// reference's draw sequence cannot be reproduced; this implementation and its oracle restatement share
// a counter-based generator (splitmix64 of seed, iteration, pick) and agree hypothesis by hypothesis.
//
// On the GPU the 50 000 hypotheses are independent: one thread fits and checks each (fp64, 3x3
// one-sided Jacobi SVD), one wavefront scores each surviving hypothesis over all correspondences, one
// workgroup picks the winner in a fixed order.
#include "common.h"

namespace imf {
namespace {

constexpr int kMaxSample = 4;   // ransac_n: 3 (3DMatch) or 4 (KITTI)

__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

struct V3 {
  double x, y, z;
};
__device__ inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ inline V3 load3(const double *p, long long i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

// Rigid fit dst ~ R src + t of n <= 4 pairs (Kabsch / Umeyama without scale): H = sum (s - ms)(d - md)^T
// = U S V^T, R = V diag(1, 1, det(V U^T)) U^T.  One-sided Jacobi on the columns of H; the column of
// the smallest singular value is replaced by the cross product of the other two, which is exactly the
// determinant correction.  T = row-major 3x4 [R | t].
__device__ void rigid_fit(const V3 *s, const V3 *d, int n, double *T) {
  V3 ms{0, 0, 0}, md{0, 0, 0};
  for (int i = 0; i < n; ++i) {
    ms.x += s[i].x; ms.y += s[i].y; ms.z += s[i].z;
    md.x += d[i].x; md.y += d[i].y; md.z += d[i].z;
  }
  const double inv = 1.0 / n;
  ms = {ms.x * inv, ms.y * inv, ms.z * inv};
  md = {md.x * inv, md.y * inv, md.z * inv};
  double B[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};   // B = H V (starts as H), column k = B[.][k]
  for (int i = 0; i < n; ++i) {
    const V3 a = sub(s[i], ms), b = sub(d[i], md);
    const double av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) B[r][c] += av[r] * bv[c];
  }
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int r = 0; r < 3; ++r) {
          al += B[r][p] * B[r][p];
          be += B[r][q] * B[r][q];
          ga += B[r][p] * B[r][q];
        }
        off = fmax(off, fabs(ga) / (sqrt(al * be) + 1e-300));
        if (fabs(ga) <= 1e-300) continue;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int r = 0; r < 3; ++r) {
          const double bp = B[r][p], bq = B[r][q];
          B[r][p] = c * bp - sn * bq;
          B[r][q] = sn * bp + c * bq;
          const double vp = V[r][p], vq = V[r][q];
          V[r][p] = c * vp - sn * vq;
          V[r][q] = sn * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  double sg[3];
  for (int k = 0; k < 3; ++k) sg[k] = sqrt(B[0][k] * B[0][k] + B[1][k] * B[1][k] + B[2][k] * B[2][k]);
  int m = 0;
  if (sg[1] < sg[m]) m = 1;
  if (sg[2] < sg[m]) m = 2;
  const int a = (m + 1) % 3, b = (m + 2) % 3;           // (a, b, m) is a cyclic permutation
  V3 ua{B[0][a], B[1][a], B[2][a]}, ub{B[0][b], B[1][b], B[2][b]};
  const double na = sg[a] > 0 ? 1.0 / sg[a] : 0.0, nb = sg[b] > 0 ? 1.0 / sg[b] : 0.0;
  ua = {ua.x * na, ua.y * na, ua.z * na};
  ub = {ub.x * nb, ub.y * nb, ub.z * nb};
  const V3 um = cross(ua, ub);                          // det[ua ub um] = +1
  const V3 va{V[0][a], V[1][a], V[2][a]}, vb{V[0][b], V[1][b], V[2][b]};
  const V3 vm = cross(va, vb);                          // V is a rotation: equals its third column
  const double U3[3][3] = {{ua.x, ub.x, um.x}, {ua.y, ub.y, um.y}, {ua.z, ub.z, um.z}};
  const double V3m[3][3] = {{va.x, vb.x, vm.x}, {va.y, vb.y, vm.y}, {va.z, vb.z, vm.z}};
  // H = sum a b^T maps the roles: columns of B live in the "d" space? B = H V with H = A^T-like sum over
  // a (rows) x b (cols): B columns are combinations of the a-space (rows index a).  R takes s to d:
  // R = Vd Us^T with Us = left vectors (a-space = source), Vd = right vectors (b-space = destination).
  double R[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R[r][c] = V3m[r][0] * U3[c][0] + V3m[r][1] * U3[c][1] + V3m[r][2] * U3[c][2];
  const double msv[3] = {ms.x, ms.y, ms.z}, mdv[3] = {md.x, md.y, md.z};
  for (int r = 0; r < 3; ++r) {
    T[4 * r + 0] = R[r][0];
    T[4 * r + 1] = R[r][1];
    T[4 * r + 2] = R[r][2];
    T[4 * r + 3] = mdv[r] - (R[r][0] * msv[0] + R[r][1] * msv[1] + R[r][2] * msv[2]);
  }
}

__device__ inline V3 apply(const double *T, V3 p) {
  return {T[0] * p.x + T[1] * p.y + T[2] * p.z + T[3], T[4] * p.x + T[5] * p.y + T[6] * p.z + T[7],
          T[8] * p.x + T[9] * p.y + T[10] * p.z + T[11]};
}

// one thread per hypothesis: sample, fit, checkers; valid[it] = 1 and Ts[it] = [R|t] when it survives
__global__ __launch_bounds__(256) void k_ransac_hypotheses(const double *__restrict__ src, const double *__restrict__ dst,
                                                           const int32_t *__restrict__ corres, int n_corres,
                                                           int ransac_n, double max_dist, double edge_sim,
                                                           uint64_t seed, int max_iter, double *__restrict__ Ts,
                                                           uint8_t *__restrict__ valid) {
  const int it = blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= max_iter) return;
  V3 s[kMaxSample], d[kMaxSample];
  for (int j = 0; j < ransac_n; ++j) {
    const uint64_t r = splitmix64(seed ^ ((uint64_t)it * kMaxSample + j));
    const int ci = (int)(r % (uint64_t)n_corres);
    s[j] = load3(src, ci);
    d[j] = load3(dst, corres[ci]);
  }
  bool ok = true;
  for (int i = 0; i < ransac_n && ok; ++i)      // CorrespondenceCheckerBasedOnEdgeLength
    for (int j = i + 1; j < ransac_n; ++j) {
      const V3 es = sub(s[i], s[j]), ed = sub(d[i], d[j]);
      const double ds = sqrt(dot(es, es)), dd = sqrt(dot(ed, ed));
      if (ds < dd * edge_sim || dd < ds * edge_sim) {
        ok = false;
        break;
      }
    }
  double T[12];
  if (ok) {
    rigid_fit(s, d, ransac_n, T);
    for (int j = 0; j < ransac_n; ++j) {          // CorrespondenceCheckerBasedOnDistance
      const V3 e = sub(apply(T, s[j]), d[j]);
      if (sqrt(dot(e, e)) > max_dist) ok = false;
    }
  }
  valid[it] = ok ? 1 : 0;
  if (ok)
    for (int k = 0; k < 12; ++k) Ts[(long long)it * 12 + k] = T[k];
}

// one wavefront per hypothesis: inlier count and squared error over all correspondences
__global__ __launch_bounds__(256) void k_ransac_score(const double *__restrict__ src, const double *__restrict__ dst,
                                                      const int32_t *__restrict__ corres, int n_corres, double max_dist,
                                                      int max_iter, const double *__restrict__ Ts,
                                                      const uint8_t *__restrict__ valid, int32_t *__restrict__ inliers,
                                                      double *__restrict__ err2) {
  const int it = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (it >= max_iter) return;
  if (!valid[it]) {
    if (lane == 0) {
      inliers[it] = -1;
      err2[it] = 0.0;
    }
    return;
  }
  double T[12];
  for (int k = 0; k < 12; ++k) T[k] = Ts[(long long)it * 12 + k];
  int cnt = 0;
  double e2 = 0.0;
  for (int i = lane; i < n_corres; i += 64) {
    const V3 e = sub(apply(T, load3(src, i)), load3(dst, corres[i]));
    const double dis = sqrt(dot(e, e));
    if (dis < max_dist) {
      ++cnt;
      e2 += dis * dis;
    }
  }
  // fixed-order reduction (xor butterfly): every lane ends with the same sums
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    e2 += __shfl_xor(e2, o, 64);
  }
  if (lane == 0) {
    inliers[it] = cnt;
    err2[it] = e2;
  }
}

// winner: most inliers, then lowest RMSE, then the earliest hypothesis
__global__ __launch_bounds__(1024) void k_ransac_select(const int32_t *__restrict__ inliers, const double *__restrict__ err2,
                                                        int max_iter, const double *__restrict__ Ts, int n_corres,
                                                        double *__restrict__ out_T, int32_t *__restrict__ meta,
                                                        double *__restrict__ stats) {
  __shared__ int s_cnt[1024], s_it[1024];
  __shared__ double s_rm[1024];
  const int t = threadIdx.x;
  int bc = -1, bi = 0x7fffffff, nvalid = 0;
  double br = 0.0;
  for (int it = t; it < max_iter; it += 1024) {
    const int c = inliers[it];
    if (c < 0) continue;
    ++nvalid;
    const double rm = c > 0 ? sqrt(err2[it] / c) : 0.0;
    if (c > bc || (c == bc && rm < br)) {   // `it` ascends per thread: strict comparisons keep the earliest
      bc = c; br = rm; bi = it;
    }
  }
  s_cnt[t] = bc; s_rm[t] = br; s_it[t] = bi;
  __shared__ int s_nv[1024];
  s_nv[t] = nvalid;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (t < o) {
      const int c = s_cnt[t + o], i2 = s_it[t + o];
      const double r = s_rm[t + o];
      const bool better = c > s_cnt[t] || (c == s_cnt[t] && (r < s_rm[t] || (r == s_rm[t] && i2 < s_it[t])));
      if (better) { s_cnt[t] = c; s_rm[t] = r; s_it[t] = i2; }
      s_nv[t] += s_nv[t + o];
    }
    __syncthreads();
  }
  if (t == 0) {
    const bool any = s_cnt[0] > 0;       // Open3D starts from fitness 0: a hypothesis must have an inlier
    meta[0] = any ? s_it[0] : -1;
    meta[1] = any ? s_cnt[0] : 0;
    meta[2] = s_nv[0];
    stats[0] = any ? (double)s_cnt[0] / n_corres : 0.0;
    stats[1] = any ? s_rm[0] : 0.0;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) {
        double v = r == c ? 1.0 : 0.0;   // identity when nothing survives (Open3D's default result)
        if (any && r < 3) v = Ts[(long long)s_it[0] * 12 + 4 * r + c];
        out_T[4 * r + c] = v;
      }
  }
}

}  // namespace
}  // namespace imf

using namespace imf;

extern "C" {

size_t imf_ransac_workspace_bytes(int max_iter) {
  if (max_iter <= 0) return 0;
  return (size_t)max_iter * (12 * 8 + 8 + 4 + 1) + 256;
}

int imf_ransac_registration(const double *src, int64_t n_src, const double *dst, int64_t n_dst,
                            const int32_t *corres, int ransac_n, double max_corr_dist, double edge_similarity,
                            int max_iter, uint64_t seed, double *out_T, int32_t *out_meta, double *out_stats,
                            void *workspace, size_t workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IMF_REQUIRE(src && dst && corres && out_T && out_meta && out_stats && workspace, "imf_ransac_registration: null pointer");
  IMF_REQUIRE(n_src >= 1 && n_dst >= 1 && n_src < (1ll << 30), "imf_ransac_registration: n_src=%lld n_dst=%lld",
              (long long)n_src, (long long)n_dst);
  IMF_REQUIRE(ransac_n >= 3 && ransac_n <= kMaxSample, "imf_ransac_registration: ransac_n=%d (3 or 4)", ransac_n);
  IMF_REQUIRE(max_iter >= 1 && max_iter <= (1 << 24), "imf_ransac_registration: max_iter=%d", max_iter);
  IMF_REQUIRE(workspace_bytes >= imf_ransac_workspace_bytes(max_iter), "imf_ransac_registration: workspace %zu < %zu",
              workspace_bytes, imf_ransac_workspace_bytes(max_iter));
  double *Ts = (double *)workspace;
  double *err2 = Ts + (size_t)max_iter * 12;
  int32_t *inl = (int32_t *)(err2 + max_iter);
  uint8_t *valid = (uint8_t *)(inl + max_iter);
  k_ransac_hypotheses<<<(unsigned)div_up(max_iter, 256), 256, 0, stream>>>(src, dst, corres, (int)n_src, ransac_n,
                                                                         max_corr_dist, edge_similarity, seed, max_iter,
                                                                         Ts, valid);
  k_ransac_score<<<(unsigned)div_up(max_iter, 4), 256, 0, stream>>>(src, dst, corres, (int)n_src, max_corr_dist, max_iter,
                                                                    Ts, valid, inl, err2);
  k_ransac_select<<<1, 1024, 0, stream>>>(inl, err2, max_iter, Ts, (int)n_src, out_T, out_meta, out_stats);
  IMF_CHECK_LAUNCH("imf_ransac_registration");
  return IMF_OK;
}

}  // extern "C"
