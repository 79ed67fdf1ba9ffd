// Native JPEG decode for the `.jpg` branch of the harness (scripts/generate_desc.py:88-92: matplotlib.image.imread of a
// JPEG = PIL = libjpeg(-turbo) with its defaults -> uint8 [H, W, 3]).  Host code only.
//
// What libjpeg computes with its defaults is fully specified integer arithmetic, restated here from the IJG algorithm
// descriptions (baseline sequential Huffman JPEG, ITU T.81; libjpeg's documented defaults JDCT_ISLOW + fancy upsampling):
//   * coefficient x quantiser, then the "slow-but-accurate" integer inverse DCT (Loeffler-Ligtenberg-Moschytz, 13-bit
//     constants, 2 extra bits after the column pass), + 128, clamped;
//   * chroma of 2x1 / 2x2 sub-sampled images: "fancy" triangle-filter upsampling (3/4 nearer + 1/4 further sample,
//     roundings 1 / 2 horizontally and 8 / 7 in the two-dimensional case, edge samples replicated);
//   * YCbCr -> RGB in 16-bit fixed point (1.40200, 1.77200, 0.71414, 0.34414).
// Pinned bit for bit against PIL in tests/test_cabi_and_host.py (qualities 30-100, 4:4:4 / 4:2:2 / 4:2:0, odd sizes, restart
// intervals).  Anything else a file may hold (progressive, arithmetic coding, 12-bit, CMYK / Adobe transforms, RGB component
// ids, other sampling factors, several scans) answers IMF_EUNSUPPORTED and the caller (imfnet_amd/dataio.py) decodes with PIL.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <exception>
#include <new>
#include <vector>

#include "common.h"

namespace imf {
namespace {

// Untrusted files: nothing a file says may escape an extern "C" function as a C++ exception (as in codecs.hip)
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

constexpr int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  bool set = false;
  uint8_t bits[17] = {0}, vals[256] = {0};
  int mincode[17], maxcode[18], valptr[17];
  void build() {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = code;
      code += bits[l];
      k += bits[l];
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
  }
};

struct Bits {   // entropy-coded segment reader: 0xFF00 -> 0xFF, any other marker ends the data (zeros are fed from there)
  const uint8_t *p, *end;
  uint32_t acc = 0;
  int n = 0;
  int marker = 0;
  void fill() {
    while (n <= 24) {
      int b = 0;
      if (!marker && p < end) {
        b = *p++;
        if (b == 0xFF) {
          int m = p < end ? *p : 0xD9;
          if (m == 0) ++p;
          else { marker = m; b = 0; --p; }
        }
      }
      acc |= (uint32_t)b << (24 - n);
      n += 8;
    }
  }
  int get(int k) {   // k <= 16
    if (k == 0) return 0;
    if (n < k) fill();
    const int v = (int)(acc >> (32 - k));
    acc <<= k;
    n -= k;
    return v;
  }
  int decode(const Huff &h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
      code = (code << 1) | get(1);
      if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
  }
  bool restart(int expect) {   // byte-align, consume RST`expect`
    acc = 0; n = 0;
    if (!marker) {   // skip fill bytes up to the marker
      while (p + 1 < end && !(p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF)) ++p;
      if (p + 1 >= end) return false;
      marker = p[1];
    }
    if (marker != 0xD0 + expect) return false;
    p += 2;
    marker = 0;
    return true;
  }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// the "islow" inverse DCT: coefficients (natural order) x quantisers -> 8 x 8 samples, stride `stride`
void idct_islow(const int16_t *coef, const uint16_t *q, uint8_t *out, int stride) {
  constexpr int CB = 13, P1 = 2;
  constexpr int F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299,
                F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
  int ws[64];
  auto descale = [](long long x, int n) { return (int)((x + (1ll << (n - 1))) >> n); };
  for (int c = 0; c < 8; ++c) {
    const int16_t *in = coef + c;
    const uint16_t *qq = q + c;
    int *w = ws + c;
    if (!in[8] && !in[16] && !in[24] && !in[32] && !in[40] && !in[48] && !in[56]) {
      const int dc = (int)in[0] * qq[0] * (1 << P1);
      for (int r = 0; r < 8; ++r) w[8 * r] = dc;
      continue;
    }
    long long z2 = (long long)in[16] * qq[16], z3 = (long long)in[48] * qq[48];
    long long z1 = (z2 + z3) * F0541;
    long long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    z2 = (long long)in[0] * qq[0]; z3 = (long long)in[32] * qq[32];
    long long tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
    const long long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = (long long)in[56] * qq[56]; tmp1 = (long long)in[40] * qq[40];
    tmp2 = (long long)in[24] * qq[24]; tmp3 = (long long)in[8] * qq[8];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    long long z4 = tmp1 + tmp3;
    const long long z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    w[0] = descale(tmp10 + tmp3, CB - P1); w[56] = descale(tmp10 - tmp3, CB - P1);
    w[8] = descale(tmp11 + tmp2, CB - P1); w[48] = descale(tmp11 - tmp2, CB - P1);
    w[16] = descale(tmp12 + tmp1, CB - P1); w[40] = descale(tmp12 - tmp1, CB - P1);
    w[24] = descale(tmp13 + tmp0, CB - P1); w[32] = descale(tmp13 - tmp0, CB - P1);
  }
  // range limit: the IJG table indexed by the low 10 bits: a signed 10-bit value + 128, clamped to 0 .. 255
  auto limit = [](int x) {
    int v = x & 1023;
    if (v >= 512) v -= 1024;
    v += 128;
    return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
  };
  for (int r = 0; r < 8; ++r) {
    const int *w = ws + 8 * r;
    uint8_t *o = out + (size_t)r * stride;
    if (!w[1] && !w[2] && !w[3] && !w[4] && !w[5] && !w[6] && !w[7]) {
      const uint8_t dc = limit(descale(w[0], P1 + 3));
      for (int c = 0; c < 8; ++c) o[c] = dc;
      continue;
    }
    long long z2 = w[2], z3 = w[6];
    long long z1 = (z2 + z3) * F0541;
    long long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    long long tmp0 = ((long long)w[0] + w[4]) * (1 << CB), tmp1 = ((long long)w[0] - w[4]) * (1 << CB);
    const long long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    long long z4 = tmp1 + tmp3;
    const long long z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    constexpr int S = CB + P1 + 3;
    o[0] = limit(descale(tmp10 + tmp3, S)); o[7] = limit(descale(tmp10 - tmp3, S));
    o[1] = limit(descale(tmp11 + tmp2, S)); o[6] = limit(descale(tmp11 - tmp2, S));
    o[2] = limit(descale(tmp12 + tmp1, S)); o[5] = limit(descale(tmp12 - tmp1, S));
    o[3] = limit(descale(tmp13 + tmp0, S)); o[4] = limit(descale(tmp13 - tmp0, S));
  }
}

struct Comp {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int bw = 0, bh = 0;            // plane size in blocks (padded to whole MCUs)
  int dw = 0, dh = 0;            // real (down-sampled) width / height in samples
  int pred = 0;
  std::vector<uint8_t> plane;    // [8 bh][8 bw]
};

struct Jpeg {
  int W = 0, H = 0, nc = 0;
  Comp c[3];
  uint16_t qt[4][64];
  bool qset[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  int restart = 0;
  const uint8_t *scan = nullptr, *end = nullptr;
};

// markers up to the first SOS.  IMF_OK / IMF_EUNSUPPORTED (a valid file this decoder does not cover) / IMF_EINVAL (corrupt)
int parse(const uint8_t *d, size_t n, Jpeg &j, const char *what) {
  if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) { set_error("%s: not a JPEG file", what); return IMF_EINVAL; }
  size_t p = 2;
  bool sof = false;
  while (p + 4 <= n) {
    if (d[p] != 0xFF) { set_error("%s: marker expected at byte %zu", what, p); return IMF_EINVAL; }
    while (p < n && d[p] == 0xFF) ++p;
    if (p >= n) break;
    const int m = d[p++];
    if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    if (p + 2 > n) break;
    const size_t len = ((size_t)d[p] << 8) | d[p + 1];
    if (len < 2 || p + len > n) { set_error("%s: segment %02X overruns the file", what, m); return IMF_EINVAL; }
    const uint8_t *s = d + p + 2;
    const size_t sl = len - 2;
    if (m == 0xC0 || m == 0xC1) {
      if (sl < 6 || s[0] != 8) { set_error("%s: %d-bit samples", what, sl ? s[0] : 0); return IMF_EUNSUPPORTED; }
      j.H = (s[1] << 8) | s[2]; j.W = (s[3] << 8) | s[4]; j.nc = s[5];
      if (j.H <= 0 || j.W <= 0 || sl < 6 + 3 * (size_t)j.nc) { set_error("%s: bad frame header", what); return IMF_EINVAL; }
      if ((long long)j.H * j.W > (1ll << 26)) { set_error("%s: %d x %d pixels (corrupt header?)", what, j.W, j.H); return IMF_EUNSUPPORTED; }
      if (j.nc != 3) { set_error("%s: %d components (3 handled natively)", what, j.nc); return IMF_EUNSUPPORTED; }
      for (int i = 0; i < 3; ++i) {
        j.c[i].id = s[6 + 3 * i]; j.c[i].h = s[7 + 3 * i] >> 4; j.c[i].v = s[7 + 3 * i] & 15; j.c[i].tq = s[8 + 3 * i];
        if (j.c[i].tq > 3) { set_error("%s: bad quantiser index", what); return IMF_EINVAL; }
      }
      sof = true;
    } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      set_error("%s: SOF%d (progressive / lossless / arithmetic) is not handled natively", what, m - 0xC0);
      return IMF_EUNSUPPORTED;
    } else if (m == 0xCC) {
      set_error("%s: arithmetic coding", what);
      return IMF_EUNSUPPORTED;
    } else if (m == 0xDB) {
      size_t o = 0;
      while (o < sl) {
        const int pq = s[o] >> 4, tq = s[o] & 15;
        ++o;
        if (tq > 3 || pq > 1 || o + (pq ? 128 : 64) > sl) { set_error("%s: bad DQT", what); return IMF_EINVAL; }
        for (int k = 0; k < 64; ++k) {
          j.qt[tq][kZigzag[k]] = pq ? (uint16_t)((s[o] << 8) | s[o + 1]) : s[o];
          o += pq ? 2 : 1;
        }
        j.qset[tq] = true;
      }
    } else if (m == 0xC4) {
      size_t o = 0;
      while (o < sl) {
        const int tc = s[o] >> 4, th = s[o] & 15;
        ++o;
        if (tc > 1 || th > 3 || o + 16 > sl) { set_error("%s: bad DHT", what); return IMF_EINVAL; }
        Huff &h = tc ? j.ac[th] : j.dc[th];
        int tot = 0;
        for (int l = 1; l <= 16; ++l) { h.bits[l] = s[o + l - 1]; tot += h.bits[l]; }
        o += 16;
        if (tot > 256 || o + tot > sl) { set_error("%s: bad DHT", what); return IMF_EINVAL; }
        memcpy(h.vals, s + o, tot);
        o += tot;
        h.build();
        h.set = true;
      }
    } else if (m == 0xDD) {
      if (sl < 2) { set_error("%s: bad DRI", what); return IMF_EINVAL; }
      j.restart = (s[0] << 8) | s[1];
    } else if (m == 0xEE) {   // Adobe: transform 1 = YCbCr is what we convert; anything else is another colour space
      if (sl >= 12 && !memcmp(s, "Adobe", 5) && s[11] != 1) { set_error("%s: Adobe colour transform %d", what, s[11]); return IMF_EUNSUPPORTED; }
    } else if (m == 0xDA) {
      if (!sof) { set_error("%s: scan before the frame header", what); return IMF_EINVAL; }
      if (sl < 1 || s[0] != j.nc || sl < 1 + 2 * (size_t)j.nc + 3) { set_error("%s: a scan of %d of %d components", what, sl ? s[0] : 0, j.nc); return IMF_EUNSUPPORTED; }
      for (int i = 0; i < j.nc; ++i) {
        if (s[1 + 2 * i] != j.c[i].id) { set_error("%s: scan component order", what); return IMF_EUNSUPPORTED; }
        j.c[i].td = s[2 + 2 * i] >> 4; j.c[i].ta = s[2 + 2 * i] & 15;
        if (j.c[i].td > 3 || j.c[i].ta > 3 || !j.dc[j.c[i].td].set || !j.ac[j.c[i].ta].set || !j.qset[j.c[i].tq]) {
          set_error("%s: scan refers to a table the file does not define", what);
          return IMF_EINVAL;
        }
      }
      j.scan = d + p + len;
      j.end = d + n;
      break;
    }
    p += len;
  }
  if (!j.scan) { set_error("%s: no scan", what); return IMF_EINVAL; }
  if (j.c[0].id == 'R' && j.c[1].id == 'G' && j.c[2].id == 'B') { set_error("%s: RGB component ids", what); return IMF_EUNSUPPORTED; }
  const bool s444 = j.c[0].h == 1 && j.c[0].v == 1, s422 = j.c[0].h == 2 && j.c[0].v == 1, s420 = j.c[0].h == 2 && j.c[0].v == 2;
  if (!(s444 || s422 || s420) || j.c[1].h != 1 || j.c[1].v != 1 || j.c[2].h != 1 || j.c[2].v != 1) {
    set_error("%s: sampling factors %dx%d / %dx%d / %dx%d (4:4:4, 4:2:2 and 4:2:0 are handled natively)", what, j.c[0].h, j.c[0].v,
              j.c[1].h, j.c[1].v, j.c[2].h, j.c[2].v);
    return IMF_EUNSUPPORTED;
  }
  return IMF_OK;
}

int decode(Jpeg &j, uint8_t *rgb, const char *what) {
  const int hmax = j.c[0].h, vmax = j.c[0].v;
  const int mcux = (j.W + 8 * hmax - 1) / (8 * hmax), mcuy = (j.H + 8 * vmax - 1) / (8 * vmax);
  for (int i = 0; i < 3; ++i) {
    Comp &c = j.c[i];
    c.bw = mcux * c.h; c.bh = mcuy * c.v;
    c.dw = (j.W * c.h + hmax - 1) / hmax; c.dh = (j.H * c.v + vmax - 1) / vmax;
    c.plane.assign((size_t)64 * c.bw * c.bh, 0);
    c.pred = 0;
  }
  Bits br{j.scan, j.end};
  int16_t coef[64];
  int next_rst = 0, left = j.restart;
  for (int my = 0; my < mcuy; ++my)
    for (int mx = 0; mx < mcux; ++mx) {
      if (j.restart && left == 0) {
        if (!br.restart(next_rst)) { set_error("%s: restart marker %d missing", what, next_rst); return IMF_EINVAL; }
        next_rst = (next_rst + 1) & 7;
        left = j.restart;
        for (int i = 0; i < 3; ++i) j.c[i].pred = 0;
      }
      --left;
      for (int i = 0; i < 3; ++i) {
        Comp &c = j.c[i];
        for (int by = 0; by < c.v; ++by)
          for (int bx = 0; bx < c.h; ++bx) {
            memset(coef, 0, sizeof(coef));
            const int t = br.decode(j.dc[c.td]);
            if (t < 0 || t > 11) { set_error("%s: corrupt entropy-coded data", what); return IMF_EINVAL; }
            c.pred += t ? extend(br.get(t), t) : 0;
            coef[0] = (int16_t)c.pred;
            for (int k = 1; k < 64; ++k) {
              const int rs = br.decode(j.ac[c.ta]);
              if (rs < 0) { set_error("%s: corrupt entropy-coded data", what); return IMF_EINVAL; }
              const int r = rs >> 4, s = rs & 15;
              if (s == 0) {
                if (r != 15) break;
                k += 15;
                continue;
              }
              k += r;
              if (k > 63) { set_error("%s: corrupt entropy-coded data", what); return IMF_EINVAL; }
              coef[kZigzag[k]] = (int16_t)extend(br.get(s), s);
            }
            const int stride = 8 * c.bw;
            idct_islow(coef, j.qt[c.tq], c.plane.data() + (size_t)(8 * (my * c.v + by)) * stride + 8 * (mx * c.h + bx), stride);
          }
      }
    }

  // colour conversion tables (16-bit fixed point)
  int cr_r[256], cb_b[256], cr_g[256], cb_g[256];
  for (int i = 0; i < 256; ++i) {
    const int x = i - 128;
    cr_r[i] = (int)((91881ll * x + 32768) >> 16);
    cb_b[i] = (int)((116130ll * x + 32768) >> 16);
    cr_g[i] = -46802 * x;
    cb_g[i] = -22554 * x + 32768;
  }
  auto clamp = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
  const Comp &Y = j.c[0], &Cb = j.c[1], &Cr = j.c[2];
  const int ys = 8 * Y.bw, cs = 8 * Cb.bw;
  const int cw = Cb.dw, ch = Cb.dh;
  std::vector<uint8_t> ub((size_t)2 * cw + 8), ur((size_t)2 * cw + 8);
  auto up_h2v1 = [&](const uint8_t *in, uint8_t *out) {   // one row, cw samples -> 2 cw
    if (cw <= 2) {   // (libjpeg filters only components more than two samples wide: plain replication otherwise)
      for (int x = 0; x < cw; ++x) out[2 * x] = out[2 * x + 1] = in[x];
      return;
    }
    out[0] = in[0];
    out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    for (int x = 1; x < cw - 1; ++x) {
      const int v = in[x] * 3;
      out[2 * x] = (uint8_t)((v + in[x - 1] + 1) >> 2);
      out[2 * x + 1] = (uint8_t)((v + in[x + 1] + 2) >> 2);
    }
    out[2 * cw - 2] = (uint8_t)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2);
    out[2 * cw - 1] = in[cw - 1];
  };
  auto up_h2v2 = [&](const uint8_t *near, const uint8_t *far, uint8_t *out) {   // one output row from two input rows
    if (cw <= 2) {
      for (int x = 0; x < cw; ++x) out[2 * x] = out[2 * x + 1] = near[x];
      return;
    }
    int thiscol = near[0] * 3 + far[0], nextcol = near[1] * 3 + far[1], lastcol;
    out[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
    out[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
    lastcol = thiscol; thiscol = nextcol;
    for (int x = 1; x < cw - 1; ++x) {
      nextcol = near[x + 1] * 3 + far[x + 1];
      out[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
      out[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
      lastcol = thiscol; thiscol = nextcol;
    }
    out[2 * cw - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
    out[2 * cw - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
  };
  for (int y = 0; y < j.H; ++y) {
    const uint8_t *yr = Y.plane.data() + (size_t)y * ys;
    const uint8_t *b, *r;
    if (hmax == 1) {
      b = Cb.plane.data() + (size_t)y * cs;
      r = Cr.plane.data() + (size_t)y * cs;
    } else if (vmax == 1) {
      up_h2v1(Cb.plane.data() + (size_t)y * cs, ub.data());
      up_h2v1(Cr.plane.data() + (size_t)y * cs, ur.data());
      b = ub.data(); r = ur.data();
    } else {
      const int cy = y >> 1;
      int fy = (y & 1) ? cy + 1 : cy - 1;          // the further row: above for even output rows, below for odd ones
      fy = fy < 0 ? 0 : fy > ch - 1 ? ch - 1 : fy; // (the first / last real row stands in beyond the edges)
      up_h2v2(Cb.plane.data() + (size_t)cy * cs, Cb.plane.data() + (size_t)fy * cs, ub.data());
      up_h2v2(Cr.plane.data() + (size_t)cy * cs, Cr.plane.data() + (size_t)fy * cs, ur.data());
      b = ub.data(); r = ur.data();
    }
    uint8_t *o = rgb + (size_t)y * j.W * 3;
    for (int x = 0; x < j.W; ++x) {
      const int yy = yr[x], cb = b[x], cr = r[x];
      o[3 * x] = clamp(yy + cr_r[cr]);
      o[3 * x + 1] = clamp(yy + ((cb_g[cb] + cr_g[cr]) >> 16));
      o[3 * x + 2] = clamp(yy + cb_b[cb]);
    }
  }
  return IMF_OK;
}

int slurp(const char *path, std::vector<uint8_t> &buf, const char *fn) {
  FILE *f = fopen(path, "rb");
  if (!f) { set_error("%s: cannot open %s", fn, path); return IMF_EINVAL; }
  fseek(f, 0, SEEK_END);
  const long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (sz <= 0 || sz > (1l << 30)) { fclose(f); set_error("%s: %s: bad file size", fn, path); return IMF_EINVAL; }
  buf.resize((size_t)sz);
  const size_t got = fread(buf.data(), 1, (size_t)sz, f);
  fclose(f);
  if (got != (size_t)sz) { set_error("%s: short read of %s", fn, path); return IMF_EINVAL; }
  return IMF_OK;
}

}  // namespace
}  // namespace imf

using namespace imf;

extern "C" {

int imf_jpeg_info(const char *path, int *h, int *w, int *channels) {
  return guarded<int>("imf_jpeg_info", [&]() -> int {
    IMF_REQUIRE(path && h && w && channels, "imf_jpeg_info: null pointer");
    std::vector<uint8_t> buf;
    int rc = slurp(path, buf, "imf_jpeg_info");
    if (rc) return rc;
    Jpeg j;
    if ((rc = parse(buf.data(), buf.size(), j, path))) return rc;
    *h = j.H; *w = j.W; *channels = 3;
    return IMF_OK;
  });
}

int imf_jpeg_read_u8(const char *path, uint8_t *out, int64_t capacity_bytes, int *h_out, int *w_out, int *c_out) {
  return guarded<int>("imf_jpeg_read_u8", [&]() -> int {
    IMF_REQUIRE(path && out && h_out && w_out && c_out, "imf_jpeg_read_u8: null pointer");
    std::vector<uint8_t> buf;
    int rc = slurp(path, buf, "imf_jpeg_read_u8");
    if (rc) return rc;
    Jpeg j;
    if ((rc = parse(buf.data(), buf.size(), j, path))) return rc;
    IMF_REQUIRE((int64_t)j.H * j.W * 3 <= capacity_bytes, "imf_jpeg_read_u8: %dx%dx3 exceeds the capacity", j.H, j.W);
    if ((rc = decode(j, out, path))) return rc;
    *h_out = j.H; *w_out = j.W; *c_out = 3;
    return IMF_OK;
  });
}

}  // extern "C"
