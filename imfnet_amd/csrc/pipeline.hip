// Streaming pipeline of the harness span (util/misc.py:67-104 as called from scripts/generate_desc.py:99-123): host
// arrays in -> voxelise -> forward -> xyz_down and descriptors back on the host, for a STREAM of fragments.
//
// The reference moves every fragment through pageable copies on the one stream its kernels run on.  Here a fragment's
// inputs arrive in a pinned block and leave in a pinned block, and both transfers run in the IDLE HALVES of the two streams a
// forward already has besides its main stream -- HIP maps streams onto four hardware queues and a fifth / sixth stream would
// share a queue with one of these anyway, in an order nobody chose (measured: a download queued ahead of the next forward's
// coarse levels delayed that forward by the whole transfer; GPU_MAX_HW_QUEUES=8 made everything 2x slower) -- so the
// placement is explicit:
//
//     image stream: [image branch k ][ upload k+1 ]      [image branch k+1][ upload k+2 ]
//     main stream : [ forward k .................. ]      [ forward k+1 ................. ]
//     side stream : [levels, maps k ][ download k-1 ]    [levels, maps k+1][ download k  ]
//
// and the ~0.25 ms + ~0.3 ms of PCIe time per S50k pair sit under the neighbouring forwards' kernels.  Two transfer
// mechanisms (DESIGN.md 4e has the measurements):
//   * the copy engines (hipMemcpyAsync, IMF_PIPELINE_SDMA_COPIES): leave the CUs alone (forward 0.975 ms under both
//     transfers), but with the HIP runtime's default ROC_CPU_WAIT_FOR_SIGNAL=1 the call BLOCKS its thread behind the
//     stream's queued kernels (~0.9 ms); the Python package sets the variable to 0 at import and then selects this mode;
//   * copy kernels that address the pinned (device-visible) blocks directly over PCIe: never block the issuing thread and
//     move only the rows that exist (the download kernel reads the row count from the forward's meta block), but their
//     wavefronts wait ~2 us per access in the same CUs' vector-memory pipes as the convolutions' LDS-DMA and slow the
//     forward they run under by 25-50 % -- the fallback when the runtime was started without the variable.
//
// All HIP calls of a job are made by ONE worker thread of the pipeline (imf_pipeline_submit only queues the job
// descriptor): the ~150 launches of a forward overlap the caller's staging of the next fragment, and no interpreter is on
// the issue path.  imf_pipeline_wait polls the job's completion mark (k_signal below), never the runtime.
#include <immintrin.h>
#include <string.h>
#include <time.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace imf {
namespace {

// ---- copy kernels: 16 bytes per lane, four independent loads in flight per lane ---------------------------------------
constexpr int kCopyThreads = 256;
constexpr int kCopyUnroll = 4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // (the nontemporal builtins want a native vector)

__device__ __forceinline__ void copy_span(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16, size_t first,
                                          size_t stride) {
  size_t i = first;
  for (; i + (kCopyUnroll - 1) * stride < n16; i += kCopyUnroll * stride) {
    u32x4 v[kCopyUnroll];
#pragma unroll
    for (int u = 0; u < kCopyUnroll; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < kCopyUnroll; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
  }
  for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// host (pinned) -> device: one contiguous block, n16 units of 16 bytes
__global__ void __launch_bounds__(kCopyThreads) k_copy_in(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16) {
  copy_span(src, dst, n16, (size_t)blockIdx.x * kCopyThreads + threadIdx.x, (size_t)gridDim.x * kCopyThreads);
}

// device -> host (pinned): [meta | rows of segment A | rows of segment B], the row count read from meta[0] on the device
// (capped by rows_cap); every segment starts 16-byte aligned and has a row size that is a multiple of 8 bytes.
__global__ void __launch_bounds__(kCopyThreads) k_copy_out(const char *__restrict__ src, char *__restrict__ dst,
                                                           const int32_t *__restrict__ meta, int meta16, size_t off_a,
                                                           int row_a, size_t off_b, int row_b, int64_t rows_cap) {
  int64_t m = meta[0];
  m = m < 0 ? 0 : (m > rows_cap ? rows_cap : m);
  const size_t first = (size_t)blockIdx.x * kCopyThreads + threadIdx.x, stride = (size_t)gridDim.x * kCopyThreads;
  copy_span((const u32x4 *)src, (u32x4 *)dst, (size_t)meta16, first, stride);
  copy_span((const u32x4 *)(src + off_a), (u32x4 *)(dst + off_a), ((size_t)m * row_a + 15) / 16, first, stride);
  copy_span((const u32x4 *)(src + off_b), (u32x4 *)(dst + off_b), ((size_t)m * row_b + 15) / 16, first, stride);
}

// Completion mark: a one-lane kernel behind the download writes the job's sequence number into a pinned word the waiting
// thread polls.  (A host thread parked in hipEventSynchronize slows every other thread's launches 3-5x on this stack --
// measured in round 3 with the writer threads and again here, 1.46 vs 0.5 ms to issue one forward -- so imf_pipeline_wait
// never enters the runtime while the job is in flight.)
__global__ void k_signal(int32_t *flag, int32_t seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct Ticket {
  imf_job job;
  int32_t seq = 0;          // what the completion mark of this submission will read
  double t_submit = 0, t_pop = 0, t_fwd = 0, t_dl = 0, t_up0 = 0, t_up1 = 0, t_dl0 = 0;   // host clock: submitted, taken by the worker, forward issued, download issued
  int state = 0;            // 0 free, 1 queued, 2 forward issued, 3 everything issued (download, or an error)
  bool uploaded = false;    // the upload was issued ahead, while the previous job was being issued
  bool flush = false;       // somebody waits for this job: do not defer its download any further
  int rc = 0;
  char err[256] = "";
  hipEvent_t e_in0 = nullptr, e_in = nullptr, e_begin = nullptr, e_fwd = nullptr, e_done = nullptr;
};

double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

struct Pipeline {
  hipStream_t main = nullptr;
  int device = 0, flags = 0, copy_blocks = 64;
  int32_t *marks = nullptr;  // pinned, 16 words (one 64-byte line) per ticket
  hipEvent_t e_epoch = nullptr;   // recorded on the main stream at creation: origin of the device-side time stamps
  double t_epoch = 0;
  int32_t next_seq = 1;
  std::vector<Ticket> tickets;
  std::deque<int> queue;     // submitted, forward not issued yet
  std::deque<int> pending;   // forward issued, download deferred
  hipEvent_t last_fwd = nullptr;   // end of the last forward issued (a ticket's e_fwd; tickets live as long as the pipeline)
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::thread worker;
  bool stop = false;
};

// Upload: on the bucket's IMAGE stream.  In issue order it lands behind the previous job's image branch (done ~halfway
// through that forward) and ahead of this job's, i.e. in the idle second half of the image stream.
int issue_upload(Pipeline &p, Ticket &t) {
  const imf_job &j = t.job;
  t.t_up0 = now_s();
  struct Stamp { double &d; ~Stamp() { d = now_s(); } } stamp{t.t_up1};
  hipStream_t st = (hipStream_t)j.io->image_stream;
  IMF_CHECK_HIP(hipEventRecord(t.e_in0, st));
  if (j.in_bytes) {
    if (p.flags & IMF_PIPELINE_SDMA_COPIES) {
      IMF_CHECK_HIP(hipMemcpyAsync(j.dev_in, j.host_in, j.in_bytes, hipMemcpyHostToDevice, st));
    } else {
      const size_t n16 = (j.in_bytes + 15) / 16;
      const unsigned nb = (unsigned)(n16 / (kCopyThreads * kCopyUnroll) + 1);
      k_copy_in<<<nb < (unsigned)p.copy_blocks ? nb : (unsigned)p.copy_blocks, kCopyThreads, 0, st>>>(
          (const u32x4 *)j.host_in, (u32x4 *)j.dev_in, n16);
      IMF_CHECK_LAUNCH("k_copy_in");
    }
  }
  IMF_CHECK_HIP(hipEventRecord(t.e_in, st));
  return IMF_OK;
}

int issue_forward(Pipeline &p, Ticket &t) {
  const imf_job &j = t.job;
  if (j.io->head_on_side) {   // the head runs on the side stream, ahead of the main one: IT waits for the upload; the bucket
    j.io->inputs_event = t.e_in;      // is free (its previous job has been waited for: imf_pipeline_wait's contract)
    j.io->reuse_event = nullptr;
  } else {
    IMF_CHECK_HIP(hipStreamWaitEvent(p.main, t.e_in, 0));
  }
  // is the GPU idle?  (no earlier forward, or the last one issued has ended: the executor then issues its side chain piecewise)
  j.io->gpu_idle_hint = 1;
  if (p.last_fwd) {
    j.io->gpu_idle_hint = hipEventQuery(p.last_fwd) == hipSuccess ? 1 : 0;
    (void)hipGetLastError();                         // (hipErrorNotReady is the other answer, not an error)
  }
  IMF_CHECK_HIP(hipEventRecord(t.e_begin, p.main));
  j.io->main_stream = p.main;
  int rc = imf_fragment_forward(j.net, j.img, j.caps, j.io);
  j.io->gpu_idle_hint = 0;
  j.io->inputs_event = nullptr;       // (the ticket's event: not the caller's to keep)
  if (rc) return rc;
  if (j.sel) {
    rc = imf_gather_points(j.io->xyz, j.io->xyz_is_f64, (const int32_t *)j.io->levels[0].first_idx, j.io->meta,
                           j.caps->rows[0], j.sel, p.main);
    if (rc) return rc;
  }
  IMF_CHECK_HIP(hipEventRecord(t.e_fwd, p.main));
  p.last_fwd = t.e_fwd;
  return IMF_OK;
}

// Download: on the bucket's SIDE stream.  Deferred (job->defer_download) it is issued right after the NEXT job's launches,
// so in stream order it sits behind that job's coarse levels and rulebooks -- the side stream's idle second half -- instead
// of in front of them (where it would hold the next forward's critical path up by the whole transfer).
int issue_download(Pipeline &p, Ticket &t) {
  const imf_job &j = t.job;
  t.t_dl0 = now_s();
  hipStream_t st = (hipStream_t)j.io->side_stream;
  IMF_CHECK_HIP(hipStreamWaitEvent(st, t.e_fwd, 0));
  if (p.flags & IMF_PIPELINE_SDMA_COPIES) {
    IMF_CHECK_HIP(hipMemcpyAsync(j.host_out, j.dev_out, j.out_bytes, hipMemcpyDeviceToHost, st));
  } else {
    k_copy_out<<<p.copy_blocks, kCopyThreads, 0, st>>>((const char *)j.dev_out, (char *)j.host_out, j.io->meta,
                                                        (IMF_META_WORDS * 4 + 15) / 16, j.sel_offset, 24, j.out_offset,
                                                        j.out_row_bytes, j.caps->rows[0]);
    IMF_CHECK_LAUNCH("k_copy_out");
  }
  IMF_CHECK_HIP(hipEventRecord(t.e_done, st));
  k_signal<<<1, 1, 0, st>>>(p.marks + 16 * (&t - p.tickets.data()), t.seq);
  IMF_CHECK_LAUNCH("k_signal");
  return IMF_OK;
}

void fail(Ticket &t, int rc) {   // (worker thread: imf_last_error is its thread-local string)
  if (!t.rc) {
    t.rc = rc;
    strncpy(t.err, imf_last_error(), sizeof(t.err) - 1);
  }
  (void)hipGetLastError();
}

void work(Pipeline *p) {
  (void)hipSetDevice(p->device);
  std::unique_lock<std::mutex> lk(p->mu);
  auto flush_wanted = [&] {
    for (int id : p->pending)
      if (p->tickets[id].flush) return true;
    return false;
  };
  // issue the deferred downloads: all of them (`all`: a newer forward has just been issued behind them), or only up to
  // the newest one somebody waits for (downloads stay in submission order on the side stream)
  auto downloads = [&](bool all) {
    while (!p->pending.empty()) {
      if (!all && !flush_wanted()) break;
      Ticket &t = p->tickets[p->pending.front()];
      p->pending.pop_front();
      lk.unlock();
      const int rc = issue_download(*p, t);
      if (rc) fail(t, rc);
      t.t_dl = now_s();
      lk.lock();
      t.state = 3;
      p->cv_done.notify_all();
    }
  };
  while (true) {
    p->cv_work.wait(lk, [&] { return p->stop || !p->queue.empty() || flush_wanted(); });
    downloads(false);
    if (p->queue.empty()) {
      if (p->stop) return;
      continue;
    }
    const int id = p->queue.front();
    p->queue.pop_front();
    Ticket &t = p->tickets[id];
    lk.unlock();
    t.t_pop = now_s();
    // (t.rc: the upload-ahead of this job, issued while it was still queued, failed -- its inputs are stale: no forward)
    int rc = t.rc ? t.rc : (t.uploaded ? IMF_OK : issue_upload(*p, t));
    if (!rc) rc = issue_forward(*p, t);
    if (rc) fail(t, rc);
    t.t_fwd = t.t_dl = now_s();
    lk.lock();
    downloads(true);   // earlier jobs' transfers: now behind this forward's side-stream work
    if (!p->queue.empty()) {   // the next job is already staged: its upload goes out now, behind this job's image branch
      Ticket &nx = p->tickets[p->queue.front()];
      lk.unlock();
      const int rc2 = issue_upload(*p, nx);
      if (rc2) fail(nx, rc2);
      lk.lock();
      nx.uploaded = true;
    }
    if (t.rc) {
      t.state = 3;
      p->cv_done.notify_all();
    } else {
      t.state = 2;
      p->pending.push_back(id);
      if (!t.job.defer_download || t.flush || p->stop) downloads(true);
    }
  }
}

}  // namespace
}  // namespace imf

using namespace imf;

extern "C" {

void *imf_pipeline_create(void *main_stream, int depth, int flags) {
  if (!main_stream || depth < 1 || depth > 64) {
    set_error("imf_pipeline_create: a main stream and 1 <= depth <= 64");
    return nullptr;
  }
  Pipeline *p = new Pipeline();
  p->main = (hipStream_t)main_stream;
  p->flags = flags;
  const int blocks = (flags >> 8) & 0xFFF;
  if (blocks) p->copy_blocks = blocks;
  if (hipGetDevice(&p->device) != hipSuccess) p->device = 0;
  p->tickets.resize((size_t)depth);
  if (hipHostMalloc((void **)&p->marks, (size_t)depth * 64, hipHostMallocDefault) != hipSuccess) {
    set_error("imf_pipeline_create: hipHostMalloc of the completion marks failed");
    delete p;
    return nullptr;
  }
  memset(p->marks, 0, (size_t)depth * 64);
  for (Ticket &t : p->tickets) {
    bool ok = hipEventCreate(&t.e_in0) == hipSuccess && hipEventCreate(&t.e_in) == hipSuccess &&
              hipEventCreate(&t.e_begin) == hipSuccess && hipEventCreate(&t.e_fwd) == hipSuccess &&
              hipEventCreate(&t.e_done) == hipSuccess;
    if (!ok) {
      set_error("imf_pipeline_create: hipEventCreate failed");
      imf_pipeline_destroy(p);
      return nullptr;
    }
  }
  if (hipEventCreate(&p->e_epoch) != hipSuccess || hipEventRecord(p->e_epoch, p->main) != hipSuccess) {
    set_error("imf_pipeline_create: epoch event");
    imf_pipeline_destroy(p);
    return nullptr;
  }
  p->t_epoch = now_s();
  p->worker = std::thread(work, p);
  return p;
}

void imf_pipeline_destroy(void *handle) {
  Pipeline *p = (Pipeline *)handle;
  if (!p) return;
  {
    std::lock_guard<std::mutex> g(p->mu);
    p->stop = true;
    for (Ticket &t : p->tickets) t.flush = true;   // nothing stays deferred
  }
  p->cv_work.notify_all();
  if (p->worker.joinable()) p->worker.join();
  for (Ticket &t : p->tickets) {
    if (t.state == 3 && t.rc == 0 && t.e_done) (void)hipEventSynchronize(t.e_done);
    for (hipEvent_t e : {t.e_in0, t.e_in, t.e_begin, t.e_fwd, t.e_done})
      if (e) (void)hipEventDestroy(e);
  }
  if (p->e_epoch) (void)hipEventDestroy(p->e_epoch);
  if (p->marks) (void)hipHostFree(p->marks);
  delete p;
}

int imf_pipeline_submit(void *handle, const imf_job *job) {
  Pipeline *p = (Pipeline *)handle;
  IMF_REQUIRE(p && job, "imf_pipeline_submit: null pointer");
  IMF_REQUIRE(job->net && job->img && job->caps && job->io && job->dev_in && job->host_in && job->dev_out && job->host_out,
              "imf_pipeline_submit: incomplete job");
  IMF_REQUIRE(job->io->side_stream && job->io->image_stream && job->io->side_stream != job->io->image_stream &&
                  job->io->side_stream != (void *)p->main && job->io->image_stream != (void *)p->main,
              "imf_pipeline_submit: the bucket needs side / image streams distinct from the pipeline's main stream");
  IMF_REQUIRE(((uintptr_t)job->dev_in | (uintptr_t)job->host_in | (uintptr_t)job->dev_out | (uintptr_t)job->host_out) % 16 == 0 &&
                  job->sel_offset % 16 == 0 && job->out_offset % 16 == 0 && job->out_row_bytes % 8 == 0 && job->out_row_bytes > 0,
              "imf_pipeline_submit: blocks and segments must be 16-byte aligned");
  std::lock_guard<std::mutex> g(p->mu);
  IMF_REQUIRE(!p->stop, "imf_pipeline_submit: the pipeline is shutting down");
  for (size_t i = 0; i < p->tickets.size(); ++i) {
    Ticket &t = p->tickets[i];
    if (t.state != 0) continue;
    t.job = *job;
    t.state = 1;
    t.uploaded = t.flush = false;
    t.t_submit = now_s();
    t.seq = p->next_seq;
    p->next_seq = p->next_seq == 0x7FFFFFFF ? 1 : p->next_seq + 1;
    t.rc = 0;
    t.err[0] = 0;
    p->queue.push_back((int)i);
    p->cv_work.notify_one();
    return (int)i;
  }
  set_error("imf_pipeline_submit: all %zu tickets are in flight (imf_pipeline_wait releases one)", p->tickets.size());
  return IMF_EINVAL;
}

int imf_pipeline_wait(void *handle, int ticket, float *ms) {
  Pipeline *p = (Pipeline *)handle;
  IMF_REQUIRE(p && ticket >= 0 && (size_t)ticket < p->tickets.size(), "imf_pipeline_wait: bad ticket %d", ticket);
  Ticket &t = p->tickets[(size_t)ticket];
  const double t_wait = now_s();
  {
    std::unique_lock<std::mutex> lk(p->mu);
    IMF_REQUIRE(t.state != 0, "imf_pipeline_wait: ticket %d is not in flight", ticket);
    if (t.state != 3) {   // still queued, or its download deferred: the waiter wants it now
      t.flush = true;
      p->cv_work.notify_one();
    }
    p->cv_done.wait(lk, [&] { return t.state == 3; });
  }
  int rc = t.rc;
  if (rc) {
    set_error("%s", t.err);
    // launches of this job may already be queued on the bucket's three streams (a failure after the upload, or half-way
    // through the forward): the bucket goes back to its owner only once they have drained
    const imf_fragment_io *io = t.job.io;
    if (io) {
      (void)hipStreamSynchronize(p->main);
      if (io->side_stream) (void)hipStreamSynchronize((hipStream_t)io->side_stream);
      if (io->image_stream) (void)hipStreamSynchronize((hipStream_t)io->image_stream);
      (void)hipGetLastError();
    }
  } else {
    // poll the completion mark (see k_signal); once in a while ask the runtime whether the device is still alive
    volatile int32_t *mark = p->marks + 16 * ticket;
    hipError_t e = hipSuccess;
    for (uint64_t spins = 1; *mark != t.seq; ++spins) {
      if (spins < 100000) {   // ~1-2 ms of polling before the first sleep (a sleep costs >= 50 us of timer slack on wake-up)
        _mm_pause();
      } else {
        struct timespec ts = {0, 20000};
        nanosleep(&ts, nullptr);
        if (spins % 20000 == 0) {   // ~ every 1.5 s of sleeping
          e = hipEventQuery(t.e_done);
          if (e == hipSuccess) break;   // complete as far as the runtime knows: the mark is there or never will be
          if (e != hipErrorNotReady) break;
          e = hipSuccess;
        }
      }
    }
    if (e != hipSuccess) {
      set_error("imf_pipeline_wait: %s", hipGetErrorString(e));
      rc = IMF_ELAUNCH;
    } else if (ms) {
      // device: [0] upload, [1] upload done -> forward done (incl. queueing behind the previous forward), [2] -> download done
      // host:   [3] submit -> taken by the worker, [4] -> forward issued, [5] -> download issued, [6] -> seen complete, [7] this call
      const double t_end = now_s();
      ms[0] = ms[1] = ms[2] = -1.f;
      (void)hipEventElapsedTime(ms + 0, t.e_in0, t.e_in);
      (void)hipEventElapsedTime(ms + 1, t.e_in, t.e_fwd);
      (void)hipEventElapsedTime(ms + 2, t.e_fwd, t.e_done);
      ms[3] = (float)((t.t_pop - t.t_submit) * 1e3);
      ms[4] = (float)((t.t_fwd - t.t_pop) * 1e3);
      ms[5] = (float)((t.t_dl - t.t_fwd) * 1e3);
      ms[6] = (float)((t_end - t.t_dl) * 1e3);
      ms[7] = (float)((t_end - t_wait) * 1e3);
      // time stamps since the pipeline's creation -- device clock: [8] upload begins, [9] ends, [10] the forward begins,
      // [11] ends, [12] download ends; host clock: [13] submit, [14] forward issued, [15] this call returns
      hipEvent_t evs[5] = {t.e_in0, t.e_in, t.e_begin, t.e_fwd, t.e_done};
      for (int i = 0; i < 5; ++i) {
        ms[8 + i] = -1.f;
        (void)hipEventElapsedTime(ms + 8 + i, p->e_epoch, evs[i]);
      }
      ms[13] = (float)((t.t_submit - p->t_epoch) * 1e3);
      ms[14] = (float)((t.t_fwd - p->t_epoch) * 1e3);
      ms[15] = (float)((t_end - p->t_epoch) * 1e3);
      ms[16] = (float)((t.t_pop - p->t_epoch) * 1e3);
      ms[17] = (float)((t.t_up0 - p->t_epoch) * 1e3);
      ms[18] = (float)((t.t_up1 - p->t_epoch) * 1e3);
      ms[19] = (float)((t.t_dl0 - p->t_epoch) * 1e3);
      ms[20] = (float)((t.t_dl - p->t_epoch) * 1e3);
    }
  }
  std::lock_guard<std::mutex> g(p->mu);
  t.state = 0;
  return rc;
}

// ---- host-side staging of a point array into a pinned block --------------------------------------------------------------
// Open3D hands the reference float64 points that ARE float32 values (a PLY stores float32; np.array(pcd.points) widens,
// scripts/generate_desc.py:83-84).  When every value survives the round trip the fragment can cross PCIe as float32 --
// half the bytes -- and the voxeliser widens before its fp64 divide (bit-identical voxels: tests/test_gpu_parity.py
// ::test_voxelize_reference_head_map[float32]).  Returns 1 when narrowed, 0 when some value is not a float32 (dst then
// holds nothing useful: the caller stages the float64 rows), < 0 on a bad argument.
__attribute__((target("avx2"))) static int narrow_avx2(const double *src, int64_t n, float *dst) {
  __m256d bad = _mm256_setzero_pd();
  int64_t i = 0;
  for (; i + 8 <= n; i += 8) {
    const __m256d a = _mm256_loadu_pd(src + i), b = _mm256_loadu_pd(src + i + 4);
    const __m128 fa = _mm256_cvtpd_ps(a), fb = _mm256_cvtpd_ps(b);
    _mm_storeu_ps(dst + i, fa);
    _mm_storeu_ps(dst + i + 4, fb);
    // NEQ_OQ: ordered and not equal -- a NaN (narrowed to a NaN) does not count as a loss
    bad = _mm256_or_pd(bad, _mm256_or_pd(_mm256_cmp_pd(_mm256_cvtps_pd(fa), a, _CMP_NEQ_OQ),
                                         _mm256_cmp_pd(_mm256_cvtps_pd(fb), b, _CMP_NEQ_OQ)));
  }
  int ok = _mm256_movemask_pd(bad) == 0;
  for (; i < n; ++i) {
    const float f = (float)src[i];
    dst[i] = f;
    ok &= ((double)f == src[i]) | (src[i] != src[i]);
  }
  return ok;
}
static int narrow_generic(const double *src, int64_t n, float *dst) {
  int ok = 1;
  for (int64_t i = 0; i < n; ++i) {
    const float f = (float)src[i];
    dst[i] = f;
    ok &= ((double)f == src[i]) | (src[i] != src[i]);
  }
  return ok;
}

int imf_host_narrow_points(const double *src, int64_t n_values, float *dst) {
  IMF_REQUIRE(src && dst && n_values >= 0, "imf_host_narrow_points: null pointer");
  static const bool avx2 = __builtin_cpu_supports("avx2");
  // chunked, so a fragment that is NOT float32-valued is found out early
  const int64_t chunk = 1 << 16;
  for (int64_t at = 0; at < n_values; at += chunk) {
    const int64_t k = n_values - at < chunk ? n_values - at : chunk;
    if (!(avx2 ? narrow_avx2(src + at, k, dst + at) : narrow_generic(src + at, k, dst + at))) return 0;
  }
  return 1;
}

}  // extern "C"
