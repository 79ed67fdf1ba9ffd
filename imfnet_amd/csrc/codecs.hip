// Host-side codecs of the batch path (host code only, no kernels): what the reference does with
// Open3D / matplotlib / OpenCV / numpy around every fragment (scripts/generate_desc.py:83-97,118-123):
//   imf_ply_read_points   o3d.io.read_point_cloud + np.array(pcd.points)          (:83,102)
//   imf_png_read_f32      matplotlib.image.imread of a .png: float32 [0,1] HWC     (:92)
//   imf_resize_bilinear   cv2.resize(INTER_LINEAR) on float images (util/uio.py:33-40)
//   imf_npz_write         np.savez_compressed(points, xyz, feature)               (:118-123)
// The python equivalents (dataio.py: a numpy structured read, PIL, torch interpolate, numpy's zlib level 6) capped
// the CLI at ~40 fragments/s -- a 30th of the GPU's rate -- mostly in the level-6 deflate of ~10 MB per fragment.
// Here: one pass over the PLY body, an inflate + unfilter PNG decoder, and a ZIP writer that stores the arrays either
// uncompressed or with a fast raw deflate (zlib level 1); np.load reads both, the arrays are identical.
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "fast_crc32.h"
#include "fast_deflate.h"

namespace imf {
namespace {

struct File {
  FILE *f = nullptr;
  explicit File(const char *path, const char *mode) { f = fopen(path, mode); }
  ~File() { if (f) fclose(f); }
  bool close() {   // flush + close with the error checked (a full disk shows up here, not in fwrite)
    if (!f) return false;
    const bool ok = fflush(f) == 0 && !ferror(f);
    const bool closed = fclose(f) == 0;
    f = nullptr;
    return ok && closed;
  }
  long size() {
    if (fseek(f, 0, SEEK_END) != 0) return -1;
    const long s = ftell(f);
    return fseek(f, 0, SEEK_SET) == 0 ? s : -1;
  }
};

// Untrusted files: nothing a file says may escape an extern "C" function as a C++ exception (std::terminate).
template <typename R, typename F>
R guarded(const char *what, F &&body) {
  try {
    return body();
  } catch (const std::bad_alloc &) {
    set_error("%s: out of memory (corrupt size field?)", what);
  } catch (const std::exception &e) {
    set_error("%s: %s", what, e.what());
  }
  return (R)IMF_EINVAL;
}

int ply_type_size(const char *t, bool &is_float) {
  is_float = false;
  if (!strcmp(t, "char") || !strcmp(t, "uchar") || !strcmp(t, "int8") || !strcmp(t, "uint8")) return 1;
  if (!strcmp(t, "short") || !strcmp(t, "ushort") || !strcmp(t, "int16") || !strcmp(t, "uint16")) return 2;
  if (!strcmp(t, "int") || !strcmp(t, "uint") || !strcmp(t, "int32") || !strcmp(t, "uint32")) return 4;
  if (!strcmp(t, "float") || !strcmp(t, "float32")) { is_float = true; return 4; }
  if (!strcmp(t, "double") || !strcmp(t, "float64")) { is_float = true; return 8; }
  return 0;
}

inline uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Helper threads of the block-parallel deflate: ONE pool per process, grown on demand and kept.  A std::thread per call was
// measured to cap the box at ~300 files/s however the writers x threads were split (every thread start maps and unmaps an
// 8 MiB stack: the same address-space lock as the per-block buffers before them).
class DeflatePool {
 public:
  ~DeflatePool() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (std::thread &t : threads_) t.join();
  }
  // run `fn` on up to `helpers` pool threads beside the caller; returns when every started copy has returned
  void run(int helpers, const std::function<void()> &fn) {
    struct Latch { std::mutex m; std::condition_variable c; int left; } latch;
    latch.left = 0;
    if (helpers > 0) {
      std::lock_guard<std::mutex> g(mu_);
      const unsigned cap = std::max(1u, std::thread::hardware_concurrency());
      while ((int)threads_.size() < helpers && threads_.size() < cap) threads_.emplace_back([this] { loop(); });
      helpers = std::min<int>(helpers, (int)threads_.size());
      latch.left = helpers;
      for (int i = 0; i < helpers; ++i)
        queue_.push_back([&fn, &latch] {
          fn();
          std::lock_guard<std::mutex> g2(latch.m);
          if (--latch.left == 0) latch.c.notify_one();
        });
    }
    if (helpers > 0) cv_.notify_all();
    fn();
    std::unique_lock<std::mutex> lk(latch.m);
    latch.c.wait(lk, [&] { return latch.left == 0; });
  }

 private:
  void loop() {
    std::unique_lock<std::mutex> lk(mu_);
    while (true) {
      cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
      if (queue_.empty()) return;
      std::function<void()> job = std::move(queue_.front());
      queue_.pop_front();
      lk.unlock();
      job();
      lk.lock();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> queue_;
  std::vector<std::thread> threads_;
  bool stop_ = false;
};
DeflatePool &deflate_pool() {
  static DeflatePool pool;
  return pool;
}

// One deflate state per thread, kept between blocks, members and files (deflateInit2 allocates ~270 KiB: mmap territory).
struct ThreadDeflate {
  z_stream zs;
  bool live = false;
  ~ThreadDeflate() { if (live) deflateEnd(&zs); }
  int begin(int level, int strategy) {
    if (!live) {
      memset(&zs, 0, sizeof(zs));
      if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy) != Z_OK) return Z_MEM_ERROR;
      live = true;
      return Z_OK;
    }
    if (deflateReset(&zs) != Z_OK) return Z_STREAM_ERROR;
    zs.next_in = nullptr; zs.avail_in = 0;
    static unsigned char sink[16];
    zs.next_out = sink; zs.avail_out = sizeof(sink);
    return deflateParams(&zs, level, strategy);   // nothing consumed yet: only switches level / strategy
  }
};

void put16(std::vector<unsigned char> &v, uint32_t x) { v.push_back(x & 255); v.push_back((x >> 8) & 255); }
void put32(std::vector<unsigned char> &v, uint32_t x) { put16(v, x & 0xFFFF); put16(v, x >> 16); }

}  // namespace
}  // namespace imf

using namespace imf;

extern "C" {

/* Vertex count of a PLY file (header only), < 0 on error. */
int64_t imf_ply_vertex_count(const char *path) {
  File fh(path, "rb");
  if (!fh.f) { set_error("imf_ply_vertex_count: cannot open %s", path); return IMF_EINVAL; }
  char line[512];
  while (fgets(line, sizeof(line), fh.f)) {
    long long n;
    if (sscanf(line, "element vertex %lld", &n) == 1) return n;
    if (!strncmp(line, "end_header", 10)) break;
  }
  set_error("imf_ply_vertex_count: no vertex element in %s", path);
  return IMF_EINVAL;
}

/* x, y, z of every vertex as float64 [n,3] (Open3D widens float vertices to double).  ascii, binary_little_endian and
 * binary_big_endian; any scalar properties beside x/y/z are skipped; list properties in the vertex element are an error.
 * Returns the number of vertices written (<= capacity) or a negative IMF_E* code. */
int64_t imf_ply_read_points(const char *path, double *out, int64_t capacity) {
  return guarded<int64_t>("imf_ply_read_points", [&]() -> int64_t {
  IMF_REQUIRE(path && out, "imf_ply_read_points: null pointer");
  File fh(path, "rb");
  IMF_REQUIRE(fh.f, "imf_ply_read_points: cannot open %s (%s)", path, strerror(errno));
  char line[512], a[64], b[64], c[64];
  IMF_REQUIRE(fgets(line, sizeof(line), fh.f) && !strncmp(line, "ply", 3), "imf_ply_read_points: %s is not a PLY file", path);
  int fmt = -1;   // 0 ascii, 1 little, 2 big
  long long n = -1;
  bool in_vertex = false, header_done = false;
  struct Prop { int size; bool is_float; int axis; };
  std::vector<Prop> props;
  while (fgets(line, sizeof(line), fh.f)) {
    if (sscanf(line, "format %63s", a) == 1) {
      fmt = !strcmp(a, "ascii") ? 0 : (!strcmp(a, "binary_little_endian") ? 1 : (!strcmp(a, "binary_big_endian") ? 2 : -1));
    } else if (sscanf(line, "element %63s %63s", a, b) == 2) {
      in_vertex = !strcmp(a, "vertex");
      if (in_vertex) n = atoll(b);
    } else if (in_vertex && sscanf(line, "property %63s %63s %63s", a, b, c) >= 2) {
      IMF_REQUIRE(strcmp(a, "list"), "imf_ply_read_points: list property in the vertex element of %s", path);
      Prop p;
      p.size = ply_type_size(a, p.is_float);
      IMF_REQUIRE(p.size > 0, "imf_ply_read_points: unknown property type '%s' in %s", a, path);
      p.axis = !strcmp(b, "x") ? 0 : (!strcmp(b, "y") ? 1 : (!strcmp(b, "z") ? 2 : -1));
      props.push_back(p);
    } else if (!strncmp(line, "end_header", 10)) {
      header_done = true;
      break;
    }
  }
  IMF_REQUIRE(header_done && fmt >= 0 && n >= 0, "imf_ply_read_points: malformed header in %s", path);
  IMF_REQUIRE(n <= capacity, "imf_ply_read_points: %lld vertices exceed the capacity %lld", n, (long long)capacity);
  int have = 0, stride = 0;
  for (const Prop &p : props) {
    if (p.axis >= 0) {
      have |= 1 << p.axis;
      IMF_REQUIRE(p.is_float, "imf_ply_read_points: x/y/z must be float or double in %s", path);
    }
    stride += p.size;
  }
  IMF_REQUIRE(have == 7, "imf_ply_read_points: the vertex element of %s has no x/y/z", path);
  if (fmt == 0) {
    for (long long i = 0; i < n; ++i) {
      for (const Prop &p : props) {
        double v;
        IMF_REQUIRE(fscanf(fh.f, "%lf", &v) == 1, "imf_ply_read_points: truncated ascii body in %s", path);
        if (p.axis >= 0) out[3 * i + p.axis] = v;
      }
    }
    return n;
  }
  {   // the vertex count comes from the file: check it against what the file can hold before allocating
    const long at = ftell(fh.f), total = fh.size();
    IMF_REQUIRE(at >= 0 && total >= at && fseek(fh.f, at, SEEK_SET) == 0 && stride > 0 &&
                    (unsigned long long)n * (unsigned long long)stride <= (unsigned long long)(total - at),
                "imf_ply_read_points: %s declares %lld vertices of %d bytes but holds %ld body bytes", path, n, stride, total - at);
  }
  std::vector<unsigned char> body((size_t)n * stride);
  IMF_REQUIRE(fread(body.data(), 1, body.size(), fh.f) == body.size(), "imf_ply_read_points: truncated body in %s", path);
  const bool swap = fmt == 2;
  int off = 0;
  for (const Prop &p : props) {
    if (p.axis >= 0) {
      const unsigned char *src = body.data() + off;
      for (long long i = 0; i < n; ++i, src += stride) {
        unsigned char tmp[8];
        if (swap) for (int k = 0; k < p.size; ++k) tmp[k] = src[p.size - 1 - k];
        else memcpy(tmp, src, p.size);
        if (p.size == 4) { float v; memcpy(&v, tmp, 4); out[3 * i + p.axis] = (double)v; }
        else { double v; memcpy(&v, tmp, 8); out[3 * i + p.axis] = v; }
      }
    }
    off += p.size;
  }
  return n;
  });
}

/* PNG header: height, width, channels (1, 2, 3 or 4 after palette expansion).  0 on success. */
int imf_png_info(const char *path, int *h, int *w, int *channels) {
  IMF_REQUIRE(path && h && w && channels, "imf_png_info: null pointer");
  File fh(path, "rb");
  IMF_REQUIRE(fh.f, "imf_png_info: cannot open %s", path);
  unsigned char hd[33];
  IMF_REQUIRE(fread(hd, 1, 33, fh.f) == 33 && !memcmp(hd, "\x89PNG\r\n\x1a\n", 8) && !memcmp(hd + 12, "IHDR", 4),
              "imf_png_info: %s is not a PNG file", path);
  *w = (int)be32(hd + 16); *h = (int)be32(hd + 20);
  const int ct = hd[25];
  *channels = ct == 0 ? 1 : (ct == 2 ? 3 : (ct == 3 ? 3 : (ct == 4 ? 2 : 4)));
  return IMF_OK;
}

/* matplotlib.image.imread semantics for .png: float32 in [0,1], [H,W,C] (8-bit samples / 255, 16-bit / 65535; palette
 * images expanded to RGB).  Non-interlaced files with 8- or 16-bit samples (the data sets' colour images); anything
 * else returns IMF_EUNSUPPORTED and the caller falls back to its generic decoder. */
int imf_png_read_f32(const char *path, float *out, int64_t capacity_floats, int *h_out, int *w_out, int *c_out) {
  return guarded<int>("imf_png_read_f32", [&]() -> int {
  IMF_REQUIRE(path && out && h_out && w_out && c_out, "imf_png_read_f32: null pointer");
  File fh(path, "rb");
  IMF_REQUIRE(fh.f, "imf_png_read_f32: cannot open %s", path);
  const long size = fh.size();
  IMF_REQUIRE(size > 33, "imf_png_read_f32: %s is not a PNG file", path);
  std::vector<unsigned char> buf((size_t)size);
  IMF_REQUIRE( fread(buf.data(), 1, buf.size(), fh.f) == buf.size() && !memcmp(buf.data(), "\x89PNG\r\n\x1a\n", 8),
              "imf_png_read_f32: %s is not a PNG file", path);
  int W = 0, H = 0, depth = 0, ct = 0, interlace = 0;
  std::vector<unsigned char> idat, plte;
  size_t pos = 8;
  bool transparency = false;
  while (pos + 12 <= buf.size()) {
    const uint32_t len = be32(&buf[pos]);
    const unsigned char *type = &buf[pos + 4], *data = &buf[pos + 8];
    if ((unsigned long long)pos + 12 + len > buf.size()) break;
    if (!memcmp(type, "IHDR", 4)) {
      IMF_REQUIRE(len >= 13, "imf_png_read_f32: short IHDR chunk in %s", path);
      W = (int)be32(data); H = (int)be32(data + 4); depth = data[8]; ct = data[9]; interlace = data[12];
    }
    else if (!memcmp(type, "tRNS", 4)) transparency = true;
    else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
    else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
    else if (!memcmp(type, "IEND", 4)) break;
    pos += 12 + len;
  }
  // tRNS: matplotlib returns RGBA for such files; 16-bit RGB(A): matplotlib / PIL go through an 8-bit path -- neither is
  // reproduced here, the caller's generic decoder takes them
  if (interlace != 0 || (depth != 8 && depth != 16) || (ct == 3 && depth != 8) || transparency ||
      (depth == 16 && (ct == 2 || ct == 6))) {
    set_error("imf_png_read_f32: %s: interlaced / sub-byte / tRNS / 16-bit colour PNG not handled natively", path);
    return IMF_EUNSUPPORTED;
  }
  const int samples = ct == 0 ? 1 : (ct == 2 ? 3 : (ct == 3 ? 1 : (ct == 4 ? 2 : 4)));
  const int C = ct == 3 ? 3 : samples;
  IMF_REQUIRE(W > 0 && H > 0 && (ct == 0 || ct == 2 || ct == 3 || ct == 4 || ct == 6), "imf_png_read_f32: bad IHDR in %s", path);
  IMF_REQUIRE((int64_t)H * W * C <= capacity_floats, "imf_png_read_f32: %dx%dx%d exceeds the capacity", H, W, C);
  const int bpp = samples * depth / 8;                         // bytes per pixel
  const size_t row = (size_t)W * bpp;
  // deflate expands at most ~1032x: a header that promises more pixels than the IDAT stream can hold is corrupt
  IMF_REQUIRE((unsigned long long)(row + 1) * (unsigned long long)H <= 1040ull * (unsigned long long)idat.size() + 64,
              "imf_png_read_f32: %s declares %dx%d pixels but holds %zu bytes of image data", path, W, H, idat.size());
  std::vector<unsigned char> raw((row + 1) * H);
  uLongf raw_len = (uLongf)raw.size();
  IMF_REQUIRE(uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) == Z_OK && raw_len == raw.size(),
              "imf_png_read_f32: corrupt image data in %s", path);
  std::vector<unsigned char> prev(row, 0), cur(row);
  for (int y = 0; y < H; ++y) {
    const unsigned char *src = &raw[(row + 1) * y];
    const int ft = src[0];
    ++src;
    for (size_t i = 0; i < row; ++i) {
      const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
      int v = src[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: set_error("imf_png_read_f32: bad filter type in %s", path); return IMF_EINVAL;
      }
      cur[i] = (unsigned char)v;
    }
    float *dst = out + (size_t)y * W * C;
    if (ct == 3) {
      for (int x = 0; x < W; ++x) {
        const size_t e = (size_t)cur[x] * 3;
        for (int k = 0; k < 3; ++k) dst[3 * x + k] = e + k < plte.size() ? plte[e + k] / 255.f : 0.f;
      }
    } else if (depth == 8) {
      for (size_t i = 0; i < row; ++i) dst[i] = cur[i] / 255.f;   // a correctly rounded divide, as numpy's
    } else {
      for (size_t i = 0; i < (size_t)W * samples; ++i) dst[i] = (float)((cur[2 * i] << 8) | cur[2 * i + 1]) / 65535.f;
    }
    prev.swap(cur);
  }
  *h_out = H; *w_out = W; *c_out = C;
  return IMF_OK;
  });
}

/* cv2.resize(image, (W_out, H_out), INTER_LINEAR) for float images [H,W,C] -> [H_out,W_out,C]: bilinear with
 * half-pixel centres, edge clamp, no anti-aliasing (util/uio.py:33-40).  chw != 0 writes [C,H_out,W_out] (the
 * transposes of generate_desc.py:96-97 folded in). */
int imf_resize_bilinear_f32(const float *in, int H, int W, int C, float *out, int H_out, int W_out, int chw) {
  return guarded<int>("imf_resize_bilinear_f32", [&]() -> int {
  IMF_REQUIRE(in && out && H > 0 && W > 0 && C > 0 && H_out > 0 && W_out > 0, "imf_resize_bilinear_f32: bad argument");
  const float sy = (float)H / H_out, sx = (float)W / W_out;
  std::vector<int> x0(W_out), x1(W_out);
  std::vector<float> fx(W_out);
  for (int x = 0; x < W_out; ++x) {
    float f = (x + 0.5f) * sx - 0.5f;
    if (f < 0.f) f = 0.f;
    int i = (int)f;
    if (i > W - 1) i = W - 1;
    x0[x] = i; x1[x] = i + 1 < W ? i + 1 : W - 1; fx[x] = f - (float)i;
  }
  for (int y = 0; y < H_out; ++y) {
    float f = (y + 0.5f) * sy - 0.5f;
    if (f < 0.f) f = 0.f;
    int i = (int)f;
    if (i > H - 1) i = H - 1;
    const int y1 = i + 1 < H ? i + 1 : H - 1;
    const float fy = f - (float)i;
    const float *r0 = in + (size_t)i * W * C, *r1 = in + (size_t)y1 * W * C;
    for (int x = 0; x < W_out; ++x) {
      for (int c = 0; c < C; ++c) {
        const float a = r0[x0[x] * C + c], b = r0[x1[x] * C + c], d = r1[x0[x] * C + c], e = r1[x1[x] * C + c];
        const float top = a + (b - a) * fx[x], bot = d + (e - d) * fx[x];
        const float v = top + (bot - top) * fy;
        if (chw) out[((size_t)c * H_out + y) * W_out + x] = v;
        else out[((size_t)y * W_out + x) * C + c] = v;
      }
    }
  }
  return IMF_OK;
  });
}

/* np.savez / np.savez_compressed replacement: a ZIP archive of .npy members (format 1.0 headers, C order).
 * names[i]: member name without ".npy"; dtype[i]: numpy descr string ("<f8", "<f4", "<i4", ...); shape: ndim[i] dims each,
 * concatenated; data[i]: host pointers; level: 0 = stored (np.savez), 2..9 = raw deflate at that zlib level
 * (np.savez_compressed uses 6), 1 = FAST: the library's own deflate producers (csrc/fast_deflate.h) -- 8-byte-item members
 * (point arrays) with matches at value granularity + dynamic Huffman, LZ77-proof members (float32 descriptors) with a
 * byte-wise Huffman code, anything else zlib level 1 -- 2.5x zlib level 1's speed at a slightly smaller file on descriptor
 * files (IMFNET_NPZ_ZLIB=1: zlib at level 1 too, for A/B).  np.load reads all of them; the arrays are identical.  Members
 * must stay below 4 GiB (no ZIP64).
 * threads > 1: BLOCK-PARALLEL deflate -- a member's bytes are cut into 256 KiB blocks, every block becomes an independent
 * raw-deflate segment that ends on a byte boundary (Z_SYNC_FLUSH; the last one Z_FINISH), the segments are concatenated
 * (a valid deflate stream: what pigz writes) and the CRC-32s of the blocks are combined (crc32_combine).  The file's bytes
 * are a function of the arrays and `level` only -- the blocks and the per-member strategy never depend on `threads`.  The
 * reference's np.savez_compressed of ~14 MB per fragment is ~0.5 s of one core -- eight GPUs' worth of fragments need it
 * spread. */
int imf_npz_write_mt(const char *path, int n_arrays, const char *const *names, const char *const *dtype, const int32_t *ndim,
                     const int64_t *shape, const void *const *data, int level, int threads) {
  return guarded<int>("imf_npz_write", [&]() -> int {
  IMF_REQUIRE(path && names && dtype && ndim && shape && data && n_arrays > 0 && level >= 0 && level <= 9,
              "imf_npz_write: bad argument");
  threads = threads < 1 ? 1 : (threads > 256 ? 256 : threads);
  constexpr size_t kBlock = 256 << 10;
  // Level 1 (the CLI's default) is served by the library's own producers (fast_deflate.h) where the member's shape allows:
  // 8-byte items -> matches at value granularity + dynamic Huffman (point arrays: 0.13 of the input, zlib level 1: 0.167,
  // several times its speed); members in which zlib's probe finds no LZ77 matches -> byte-wise Huffman.  Everything else,
  // every other level, and IMFNET_NPZ_ZLIB=1 (A/B): zlib.
  enum { kZlib = 0, kValues64 = 1, kHuffman = 2 };
  static const bool zlib_only = getenv("IMFNET_NPZ_ZLIB") && atoi(getenv("IMFNET_NPZ_ZLIB")) != 0;
  struct Member {
    std::vector<unsigned char> head;   // the .npy header
    const unsigned char *src; size_t nbytes; uint32_t usize; int strategy; size_t first_block, n_blocks; int producer;
  };
  struct Block { int member; size_t lo, len, out_at, out_cap, used; uLong crc; size_t in_len; int rc; };
  std::vector<Member> mem((size_t)n_arrays);
  std::vector<Block> blocks;
  size_t arena_bytes = 0;
  const int64_t *sh = shape;

  // ---- phase 1 (this thread): headers, the per-member strategy, the block list ---------------------------------------
  for (int i = 0; i < n_arrays; ++i) {
    // numeric / bool descrs only ("<f8", "<i4", "|u1", "|b1" ...): the digits are the item size.  Strings ("<U7": 4 bytes
    // per character, "|S3") and anything else are refused -- the caller writes such arrays with numpy
    IMF_REQUIRE(dtype[i] && strlen(dtype[i]) >= 3 && strchr("<|=", dtype[i][0]) && strchr("fiub", dtype[i][1]),
                "imf_npz_write: dtype '%s' (numeric little-endian descrs only)", dtype[i] ? dtype[i] : "(null)");
    const int itemsize = atoi(dtype[i] + 2);
    IMF_REQUIRE(itemsize == 1 || itemsize == 2 || itemsize == 4 || itemsize == 8, "imf_npz_write: dtype '%s'", dtype[i]);
    size_t count = 1;
    std::string shp = "(";
    for (int d = 0; d < ndim[i]; ++d) {
      count *= (size_t)sh[d];
      shp += std::to_string((long long)sh[d]) + (ndim[i] == 1 || d + 1 < ndim[i] ? "," : "");
      if (d + 1 < ndim[i]) shp += " ";
    }
    shp += ")";
    sh += ndim[i];
    Member &m = mem[(size_t)i];
    m.nbytes = count * (size_t)itemsize;
    IMF_REQUIRE(m.nbytes < (1ull << 32) - 256, "imf_npz_write: member %s too large", names[i]);
    IMF_REQUIRE(data[i] || m.nbytes == 0, "imf_npz_write: member %s has no data", names[i]);
    std::string hdr = std::string("{'descr': '") + dtype[i] + "', 'fortran_order': False, 'shape': " + shp + ", }";
    const size_t unpadded = 10 + hdr.size() + 1;
    hdr.append((64 - unpadded % 64) % 64, ' ');
    hdr += '\n';
    m.head = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    put16(m.head, (uint32_t)hdr.size());
    m.head.insert(m.head.end(), hdr.begin(), hdr.end());
    m.usize = (uint32_t)(m.head.size() + m.nbytes);
    m.src = (const unsigned char *)data[i];
    // Strategy per member, decided by the data alone (never by `threads`: the file's bytes do not depend on it): a probe
    // deflates the array's first 64 KiB; when LZ77 finds next to nothing (float32 descriptors: 0.93 of the input at 25 MB/s
    // per core) the member is Huffman-coded only (the same 0.93 at ~95 MB/s); point arrays (0.2 at ~110 MB/s) keep it.
    m.strategy = Z_DEFAULT_STRATEGY;
    const bool values64 = level == 1 && !zlib_only && itemsize == 8;   // (no probe needed: the producer is chosen by the item size)
    if (level > 0 && m.nbytes >= (64 << 10) && !values64) {
      z_stream ps;
      memset(&ps, 0, sizeof(ps));
      IMF_REQUIRE(deflateInit2(&ps, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK, "imf_npz_write: deflateInit2");
      std::vector<unsigned char> tmp(deflateBound(&ps, 64 << 10) + 64);
      ps.next_in = const_cast<unsigned char *>(m.src); ps.avail_in = 64 << 10;
      ps.next_out = tmp.data(); ps.avail_out = (uInt)tmp.size();
      const int prc = deflate(&ps, Z_FINISH);
      const size_t got = (size_t)ps.total_out;
      deflateEnd(&ps);
      if (prc == Z_STREAM_END && got * 100 > (size_t)(64 << 10) * 85) m.strategy = Z_HUFFMAN_ONLY;
    }
    m.producer = kZlib;
    if (values64) m.producer = kValues64;
    else if (level == 1 && !zlib_only && m.strategy == Z_HUFFMAN_ONLY) m.producer = kHuffman;
    // block b covers [b * kBlock, ...) of the array's bytes; block 0 is preceded by the .npy header
    m.first_block = blocks.size();
    m.n_blocks = m.nbytes ? (m.nbytes + kBlock - 1) / kBlock : 1;
    for (size_t bi = 0; bi < m.n_blocks; ++bi) {
      Block bl;
      memset(&bl, 0, sizeof(bl));
      bl.member = i;
      bl.lo = bi * kBlock;
      bl.len = m.nbytes - bl.lo < kBlock ? m.nbytes - bl.lo : kBlock;
      bl.in_len = bl.len + (bi == 0 ? m.head.size() : 0);
      bl.out_cap = level > 0 ? bl.in_len + bl.in_len / 512 + 256 : 0;   // >= deflateBound for raw deflate + the sync marker
      bl.out_at = arena_bytes;
      arena_bytes += bl.out_cap;
      blocks.push_back(bl);
    }
  }

  // ---- phase 2 (all threads): CRC-32 and one raw-deflate segment per block, into slices of ONE arena --------------------
  // The arena belongs to the calling thread and is kept between calls (writer threads call this file after file): a fresh
  // 14 MB buffer per file, or a 256 KiB one per block, is an mmap + page faults + munmap per allocation, and with a hundred
  // threads doing it the kernel's address-space lock -- not zlib -- set the rate (measured: 128 writers x 1 thread 8x slower
  // per file than one writer alone).
  static thread_local std::vector<unsigned char> arena;
  if (arena.size() < arena_bytes) arena.resize(arena_bytes + arena_bytes / 4);
  unsigned char *const out0 = arena.data();
  std::atomic<size_t> next{0};
  auto work = [&]() {
    static thread_local ThreadDeflate td;
    z_stream &zs = td.zs;
    for (size_t b = next.fetch_add(1); b < blocks.size(); b = next.fetch_add(1)) {
      Block &bl = blocks[b];
      const Member &m = mem[(size_t)bl.member];
      const bool first = bl.lo == 0, last = b + 1 == m.first_block + m.n_blocks;
      uLong crc = crc32(0L, Z_NULL, 0);
      if (first) crc = fast_crc32(crc, m.head.data(), m.head.size());
      if (bl.len) crc = fast_crc32(crc, m.src + bl.lo, bl.len);        // carry-less multiplies: ~15x zlib 1.2.11's table loop
      bl.crc = crc;
      bl.rc = Z_OK;
      if (level == 0) continue;
      if (m.producer != kZlib) {
        const unsigned char *head = first ? m.head.data() : nullptr;
        const size_t head_len = first ? m.head.size() : 0;
        bl.used = m.producer == kValues64
                      ? fdef::deflate_values64(head, head_len, m.src + bl.lo, bl.len, last, out0 + bl.out_at, bl.out_cap)
                      : fdef::deflate_huffman(head, head_len, m.src + bl.lo, bl.len, last, out0 + bl.out_at, bl.out_cap);
        bl.rc = bl.used ? Z_OK : Z_BUF_ERROR;
        continue;
      }
      int rc = td.begin(level, m.strategy);
      if (rc != Z_OK) { bl.rc = rc; continue; }
      zs.next_out = out0 + bl.out_at; zs.avail_out = (uInt)bl.out_cap;
      if (first) {
        zs.next_in = const_cast<unsigned char *>(m.head.data()); zs.avail_in = (uInt)m.head.size();
        rc = deflate(&zs, Z_NO_FLUSH);
      }
      zs.next_in = const_cast<unsigned char *>(m.src + bl.lo); zs.avail_in = (uInt)bl.len;
      if (rc == Z_OK) rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
      bl.rc = (last ? rc == Z_STREAM_END : (rc == Z_OK && zs.avail_in == 0 && zs.avail_out > 0)) ? Z_OK : Z_STREAM_ERROR;
      bl.used = (size_t)((out0 + bl.out_at + bl.out_cap - zs.avail_out) - (out0 + bl.out_at));
    }
  };
  deflate_pool().run((int)std::min<size_t>(blocks.size(), (size_t)threads) - 1, work);

  // ---- phase 3 (this thread): the ZIP file -----------------------------------------------------------------------------
  File fh(path, "wb");
  IMF_REQUIRE(fh.f, "imf_npz_write: cannot create %s (%s)", path, strerror(errno));
  std::vector<unsigned char> central;
  uint32_t offset = 0;
  for (int i = 0; i < n_arrays; ++i) {
    const Member &m = mem[(size_t)i];
    uLong crc = 0;
    size_t csize64 = 0;
    for (size_t b = m.first_block; b < m.first_block + m.n_blocks; ++b) {
      IMF_REQUIRE(blocks[b].rc == Z_OK, "imf_npz_write: deflate failed in block %zu of %s", b - m.first_block, names[i]);
      crc = b == m.first_block ? blocks[b].crc : crc32_combine(crc, blocks[b].crc, (z_off_t)blocks[b].in_len);
      csize64 += blocks[b].used;
    }
    IMF_REQUIRE(level == 0 || csize64 < (1ull << 32) - 256, "imf_npz_write: member %s too large", names[i]);
    const uint32_t csize = level > 0 ? (uint32_t)csize64 : m.usize;
    const std::string fname = std::string(names[i]) + ".npy";
    std::vector<unsigned char> local;
    put32(local, 0x04034b50); put16(local, 20); put16(local, 0); put16(local, level > 0 ? 8 : 0);
    put16(local, 0); put16(local, 0x21);                                  // time / date (1980-01-01)
    put32(local, (uint32_t)crc); put32(local, csize); put32(local, m.usize);
    put16(local, (uint32_t)fname.size()); put16(local, 0);
    local.insert(local.end(), fname.begin(), fname.end());
    bool ok = fwrite(local.data(), 1, local.size(), fh.f) == local.size();
    if (level > 0) {
      for (size_t b = m.first_block; b < m.first_block + m.n_blocks && ok; ++b)
        ok = fwrite(out0 + blocks[b].out_at, 1, blocks[b].used, fh.f) == blocks[b].used;
    } else {
      ok = ok && fwrite(m.head.data(), 1, m.head.size(), fh.f) == m.head.size() &&
           (m.nbytes == 0 || fwrite(m.src, 1, m.nbytes, fh.f) == m.nbytes);
    }
    IMF_REQUIRE(ok, "imf_npz_write: short write to %s", path);
    put32(central, 0x02014b50); put16(central, 20); put16(central, 20); put16(central, 0); put16(central, level > 0 ? 8 : 0);
    put16(central, 0); put16(central, 0x21);
    put32(central, (uint32_t)crc); put32(central, csize); put32(central, m.usize);
    put16(central, (uint32_t)fname.size()); put16(central, 0); put16(central, 0); put16(central, 0); put16(central, 0);
    put32(central, 0); put32(central, offset);
    central.insert(central.end(), fname.begin(), fname.end());
    offset += (uint32_t)local.size() + csize;
  }
  std::vector<unsigned char> end;
  put32(end, 0x06054b50); put16(end, 0); put16(end, 0); put16(end, (uint32_t)n_arrays); put16(end, (uint32_t)n_arrays);
  put32(end, (uint32_t)central.size()); put32(end, offset); put16(end, 0);
  IMF_REQUIRE(fwrite(central.data(), 1, central.size(), fh.f) == central.size() && fwrite(end.data(), 1, end.size(), fh.f) == end.size(),
              "imf_npz_write: short write to %s", path);
  IMF_REQUIRE(fh.close(), "imf_npz_write: flush / close of %s failed (%s)", path, strerror(errno));
  return IMF_OK;
  });
}

int imf_npz_write(const char *path, int n_arrays, const char *const *names, const char *const *dtype, const int32_t *ndim,
                  const int64_t *shape, const void *const *data, int level) {
  return imf_npz_write_mt(path, n_arrays, names, dtype, ndim, shape, data, level, 1);
}

}  // extern "C"
