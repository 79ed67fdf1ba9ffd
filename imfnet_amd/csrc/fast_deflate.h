// Two raw-DEFLATE (RFC 1951) producers for the descriptor files' members (host code; used by imf_npz_write_mt at level 1).
//
// The reference writes every fragment with np.savez_compressed (scripts/generate_desc.py:118-123): zlib level 6 over
// ~14 MB, 0.5 s of one core.  Round 4's writer cut the members into independent 256 KiB segments and ran zlib level 1 on
// them in parallel -- 66 ms of one core per file, and the CLI was bound by exactly that on the GPU box's 16-CPU quota
// (205-237 fragments/s against ~2 000 of GPU capacity).  What the members ARE makes a general LZ77 search
// unnecessary:
//   * `points` / `xyz` are float64 arrays of float32-VALUED coordinates of a TSDF-fused mesh: two of a vertex's three
//     coordinates lie on grid lines, so 98-99 % of the 8-byte values of the in-tree fragments repeat an earlier value
//     EXACTLY within the 32 KiB window.  deflate_values64 therefore looks for matches at 8-byte granularity only: ONE hash
//     probe per value (zlib level 1: one per byte), matches of 8 k bytes at distances of 8 d bytes, everything else as
//     literals; then one dynamic-Huffman block per segment.  On the in-tree fragments: 0.13 of the input (zlib level 1:
//     0.167, level 6: 0.135).
//   * `feature` is float32, L2-normalised: LZ77 finds nothing (zlib: 0.93 of the input at 25 MB/s, Z_HUFFMAN_ONLY the same
//     0.93 at 95 MB/s).  deflate_huffman is a plain byte histogram -> canonical code -> table-driven encoder.
// Both write ONE dynamic-Huffman block per segment (a stored block when that is not smaller), then either the final-block
// bit or zlib's Z_SYNC_FLUSH marker (an empty stored block: the next segment starts on a byte boundary) -- the segments of
// a member concatenate into one valid raw deflate stream, exactly as round 4's zlib segments did.  Pure functions of
// their input: the file's bytes do not depend on the thread count.  zlib remains the decoder's reference:
// tests/test_cabi_and_host.py inflates every stream with zlib and compares.
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace imf {
namespace fdef {

struct BitWriter {
  unsigned char *p, *end;
  uint64_t acc = 0;
  int n = 0;
  bool overflow = false;
  BitWriter(unsigned char *out, size_t cap) : p(out), end(out + cap) {}
  inline void put(uint32_t bits, int len) {           // LSB-first; len <= 32
    acc |= (uint64_t)bits << n;
    n += len;
    while (n >= 8) {
      if (p < end) *p++ = (unsigned char)acc; else overflow = true;
      acc >>= 8;
      n -= 8;
    }
  }
  inline void align() { if (n) put(0, 8 - n); }
  inline void byte(unsigned v) { if (p < end) *p++ = (unsigned char)v; else overflow = true; }
};

// Code lengths of an optimal prefix code limited to `maxbits`: plain Huffman (two-queue merge over the sorted
// frequencies); when the tree is too deep the frequencies are flattened (f -> (f + 1) / 2) and the code rebuilt -- a few
// rounds at most, the cost is a fraction of a percent of the segment.  len[s] = 0 for unused symbols; one used symbol
// gets length 1.
inline void code_lengths(const uint32_t *freq, int n, int maxbits, uint8_t *len) {
  std::vector<uint32_t> f(freq, freq + n);
  memset(len, 0, (size_t)n);
  while (true) {
    struct Node { uint64_t w; int l, r; };
    std::vector<Node> nodes;
    std::vector<int> leaves;
    for (int s = 0; s < n; ++s)
      if (f[s]) { leaves.push_back((int)nodes.size()); nodes.push_back({f[s], -1 - s, -1}); }
    if (leaves.empty()) return;
    if (leaves.size() == 1) { len[-1 - nodes[0].l] = 1; return; }
    std::sort(leaves.begin(), leaves.end(), [&](int a, int b) {
      return nodes[a].w != nodes[b].w ? nodes[a].w < nodes[b].w : nodes[a].l > nodes[b].l;   // (ties by symbol: deterministic)
    });
    std::vector<int> q2;
    size_t i1 = 0, i2 = 0;
    auto pop = [&]() {
      const bool from1 = i1 < leaves.size() && (i2 >= q2.size() || nodes[leaves[i1]].w <= nodes[q2[i2]].w);
      return from1 ? leaves[i1++] : q2[i2++];
    };
    const size_t total = leaves.size();
    for (size_t k = 0; k + 1 < total; ++k) {
      const int a = pop(), b = pop();
      nodes.push_back({nodes[a].w + nodes[b].w, a, b});
      q2.push_back((int)nodes.size() - 1);
    }
    // depths, root = last node
    std::vector<int> depth(nodes.size(), 0);
    int deepest = 0;
    for (int k = (int)nodes.size() - 1; k >= 0; --k) {
      if (nodes[k].r >= 0 || nodes[k].l >= 0) {          // internal (leaf: l < 0, r == -1)
        if (nodes[k].r >= 0) {
          depth[nodes[k].l] = depth[nodes[k].r] = depth[k] + 1;
        }
      }
      if (nodes[k].r < 0) deepest = std::max(deepest, depth[k]);
    }
    if (deepest <= maxbits) {
      for (size_t k = 0; k < nodes.size(); ++k)
        if (nodes[k].r < 0) len[-1 - nodes[k].l] = (uint8_t)depth[k];
      return;
    }
    for (int s = 0; s < n; ++s)
      if (f[s]) f[s] = (f[s] + 1) / 2;
  }
}

inline uint32_t reverse_bits(uint32_t v, int len) {
  uint32_t r = 0;
  for (int i = 0; i < len; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
  return r;
}

// canonical codes (RFC 1951 3.2.2), already bit-reversed for the LSB-first writer
inline void canonical_codes(const uint8_t *len, int n, uint16_t *code) {
  int bl_count[16] = {0}, next[16] = {0};
  for (int s = 0; s < n; ++s) bl_count[len[s]]++;
  bl_count[0] = 0;
  int c = 0;
  for (int b = 1; b < 16; ++b) { c = (c + bl_count[b - 1]) << 1; next[b] = c; }
  for (int s = 0; s < n; ++s) code[s] = len[s] ? (uint16_t)reverse_bits((uint32_t)next[len[s]]++, len[s]) : 0;
}

constexpr int kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
constexpr int kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
constexpr int kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
constexpr int kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct SymbolTables {                                  // length 3 .. 258 -> 0 .. 28; distance 1 .. 32768 -> 0 .. 29 (zlib's two-level table)
  uint8_t len[259], dist[512];
  SymbolTables() {
    for (int l = 0; l < 259; ++l) { int s = 28; while (s > 0 && kLenBase[s] > l) --s; len[l] = (uint8_t)s; }
    for (int d = 1; d <= 256; ++d) { int s = 29; while (kDistBase[s] > d) --s; dist[d - 1] = (uint8_t)s; }
    for (int k = 2; k < 256; ++k) { const int d = (k << 7) + 1; int s = 29; while (kDistBase[s] > d) --s; dist[256 + k] = (uint8_t)s; }
    dist[256] = dist[257] = 0;                          // (unused: distances <= 256 take the first half)
  }
};
inline const SymbolTables &symbol_tables() { static const SymbolTables t; return t; }
inline int length_symbol(int l) { return symbol_tables().len[l]; }
inline int dist_symbol(int d) {                        // codes >= 16 start at 2^k + 1 and 3 * 2^(k-1) + 1, k >= 8: (d - 1) >> 7 decides
  const SymbolTables &t = symbol_tables();
  return d <= 256 ? t.dist[d - 1] : t.dist[256 + ((d - 1) >> 7)];
}

// A segment as tokens: literal = the byte; match = 0x80000000 | (length << 16) | (distance - 1)   (length <= 258, distance <= 32768)
inline bool is_match(uint32_t t) { return (t & 0x80000000u) != 0; }

// One dynamic-Huffman block for `tok` (+ end-of-block), or stored blocks of `raw` when those are not larger; then the
// segment's tail: final-block bit set on the (last) block, or the sync-flush marker.  Returns bytes written, 0 = overflow.
inline size_t write_segment(const std::vector<uint32_t> &tok, const unsigned char *const *raw_parts, const size_t *raw_len,
                            int n_parts, bool last, unsigned char *out, size_t cap) {
  uint32_t lf[286] = {0}, df[30] = {0};
  uint64_t extra_bits = 0;
  for (uint32_t t : tok) {
    if (is_match(t)) {
      const int ls = length_symbol((int)((t >> 16) & 0x1FF)), ds = dist_symbol((int)(t & 0xFFFF) + 1);
      lf[257 + ls]++; df[ds]++;
      extra_bits += (uint64_t)(kLenExtra[ls] + kDistExtra[ds]);
    } else {
      lf[t]++;
    }
  }
  lf[256] = 1;
  uint8_t ll[286], dl[30];
  code_lengths(lf, 286, 15, ll);
  code_lengths(df, 30, 15, dl);
  int hlit = 286, hdist = 30;
  while (hlit > 257 && ll[hlit - 1] == 0) --hlit;
  while (hdist > 1 && dl[hdist - 1] == 0) --hdist;
  // the code-length sequence, run-length coded with symbols 16 / 17 / 18 (RFC 1951 3.2.7)
  std::vector<uint8_t> seq(ll, ll + hlit);
  seq.insert(seq.end(), dl, dl + hdist);
  struct CL { uint8_t sym, extra; };
  std::vector<CL> cl;
  for (size_t i = 0; i < seq.size();) {
    size_t j = i;
    while (j < seq.size() && seq[j] == seq[i]) ++j;
    size_t run = j - i;
    if (seq[i] == 0) {
      while (run >= 11) { const size_t r = std::min<size_t>(run, 138); cl.push_back({18, (uint8_t)(r - 11)}); run -= r; }
      if (run >= 3) { cl.push_back({17, (uint8_t)(run - 3)}); run = 0; }
      while (run--) cl.push_back({0, 0});
    } else {
      cl.push_back({seq[i], 0});
      --run;
      while (run >= 3) { const size_t r = std::min<size_t>(run, 6); cl.push_back({16, (uint8_t)(r - 3)}); run -= r; }
      while (run--) cl.push_back({seq[i], 0});
    }
    i = j;
  }
  uint32_t cf[19] = {0};
  for (const CL &c : cl) cf[c.sym]++;
  uint8_t cll[19];
  code_lengths(cf, 19, 7, cll);
  static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  int hclen = 19;
  while (hclen > 4 && cll[order[hclen - 1]] == 0) --hclen;
  uint16_t lc[286], dc[30], cc[19];
  canonical_codes(ll, 286, lc);
  canonical_codes(dl, 30, dc);
  canonical_codes(cll, 19, cc);
  uint64_t bits = 3 + 5 + 5 + 4 + 3ull * hclen + extra_bits;
  for (const CL &c : cl) bits += cll[c.sym] + (c.sym == 16 ? 2 : c.sym == 17 ? 3 : c.sym == 18 ? 7 : 0);
  for (int s = 0; s < 286; ++s) bits += (uint64_t)lf[s] * ll[s];
  for (int s = 0; s < 30; ++s) bits += (uint64_t)df[s] * dl[s];
  size_t raw_total = 0;
  for (int i = 0; i < n_parts; ++i) raw_total += raw_len[i];
  const uint64_t stored_bytes = raw_total + 5 * ((raw_total + 65534) / 65535 + (raw_total == 0));

  BitWriter w(out, cap);
  if ((bits + 7) / 8 >= stored_bytes) {                 // incompressible: stored blocks (<= 65 535 bytes each)
    std::vector<unsigned char> flat;
    flat.reserve(raw_total);
    for (int i = 0; i < n_parts; ++i) flat.insert(flat.end(), raw_parts[i], raw_parts[i] + raw_len[i]);
    size_t at = 0;
    do {
      const size_t n = std::min<size_t>(65535, flat.size() - at);
      const bool fin = last && at + n == flat.size();
      w.put(fin ? 1 : 0, 1); w.put(0, 2); w.align();
      w.byte(n & 255); w.byte(n >> 8); w.byte(~n & 255); w.byte((~n >> 8) & 255);
      for (size_t k = 0; k < n; ++k) w.byte(flat[at + k]);
      at += n;
    } while (at < flat.size());
  } else {
    w.put(last ? 1 : 0, 1); w.put(2, 2);
    w.put((uint32_t)(hlit - 257), 5); w.put((uint32_t)(hdist - 1), 5); w.put((uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; ++i) w.put(cll[order[i]], 3);
    for (const CL &c : cl) {
      w.put(cc[c.sym], cll[c.sym]);
      if (c.sym == 16) w.put(c.extra, 2); else if (c.sym == 17) w.put(c.extra, 3); else if (c.sym == 18) w.put(c.extra, 7);
    }
    for (uint32_t t : tok) {
      if (is_match(t)) {
        const int l = (int)((t >> 16) & 0x1FF), d = (int)(t & 0xFFFF) + 1;
        const int ls = length_symbol(l), ds = dist_symbol(d);
        w.put(lc[257 + ls], ll[257 + ls]);
        if (kLenExtra[ls]) w.put((uint32_t)(l - kLenBase[ls]), kLenExtra[ls]);
        w.put(dc[ds], dl[ds]);
        if (kDistExtra[ds]) w.put((uint32_t)(d - kDistBase[ds]), kDistExtra[ds]);
      } else {
        w.put(lc[t], ll[t]);
      }
    }
    w.put(lc[256], ll[256]);
  }
  if (last) {
    w.align();
  } else {                                             // Z_SYNC_FLUSH: an empty stored block ends the segment on a byte boundary
    w.put(0, 3); w.align();
    w.byte(0); w.byte(0); w.byte(0xFF); w.byte(0xFF);
  }
  return w.overflow ? 0 : (size_t)(w.p - out);
}

// `head` (the .npy header in front of a member's first segment: literals) + `n_values` 8-byte values.
inline size_t deflate_values64(const unsigned char *head, size_t head_len, const unsigned char *src, size_t n_bytes, bool last,
                               unsigned char *out, size_t cap) {
  const size_t n = n_bytes / 8;
  std::vector<uint32_t> tok;
  tok.reserve(head_len + n_bytes / 4 + 16);
  for (size_t i = 0; i < head_len; ++i) tok.push_back(head[i]);
  constexpr int kHashBits = 14;
  std::vector<uint32_t> table(1u << kHashBits, 0);      // value index + 1 of the latest occurrence of a hash
  auto value = [&](size_t i) { uint64_t v; memcpy(&v, src + 8 * i, 8); return v; };
  size_t i = 0;
  while (i < n) {
    const uint64_t v = value(i);
    const uint32_t h = (uint32_t)((v * 0x9E3779B97F4A7C15ull) >> (64 - kHashBits));
    const uint32_t cand = table[h];
    table[h] = (uint32_t)i + 1;
    if (cand && i + 1 - cand <= 4096 && value(cand - 1) == v) {
      const size_t j = cand - 1;
      size_t m = 1;
      while (m < 32 && i + m < n && value(i + m) == value(j + m)) {
        const uint64_t vm = value(i + m);
        table[(uint32_t)((vm * 0x9E3779B97F4A7C15ull) >> (64 - kHashBits))] = (uint32_t)(i + m) + 1;
        ++m;
      }
      tok.push_back(0x80000000u | ((uint32_t)(8 * m) << 16) | (uint32_t)(8 * (i - j) - 1));
      i += m;
    } else {
      for (int b = 0; b < 8; ++b) tok.push_back(src[8 * i + b]);
      ++i;
    }
  }
  for (size_t b = 8 * n; b < n_bytes; ++b) tok.push_back(src[b]);
  const unsigned char *parts[2] = {head, src};
  const size_t lens[2] = {head_len, n_bytes};
  return write_segment(tok, parts, lens, 2, last, out, cap);
}

// `head` + bytes, Huffman-coded literals only.  The byte loop is the whole cost (6.6 MB of descriptors per S50k fragment), so
// it does not go through the token vector: four interleaved histograms, one (code | length << 16) table, and the codes of
// four bytes (<= 60 bits) gathered in a register before they reach the writer.
inline size_t deflate_huffman(const unsigned char *head, size_t head_len, const unsigned char *src, size_t n_bytes, bool last,
                              unsigned char *out, size_t cap) {
  uint32_t h4[4][256];
  memset(h4, 0, sizeof(h4));
  size_t i = 0;
  for (; i + 4 <= n_bytes; i += 4) { h4[0][src[i]]++; h4[1][src[i + 1]]++; h4[2][src[i + 2]]++; h4[3][src[i + 3]]++; }
  for (; i < n_bytes; ++i) h4[0][src[i]]++;
  for (size_t k = 0; k < head_len; ++k) h4[0][head[k]]++;
  uint32_t lf[286] = {0};
  for (int b = 0; b < 256; ++b) lf[b] = h4[0][b] + h4[1][b] + h4[2][b] + h4[3][b];
  lf[256] = 1;
  uint8_t ll[286];
  code_lengths(lf, 286, 15, ll);
  uint16_t lc[286];
  canonical_codes(ll, 286, lc);
  uint64_t bits = 0;
  for (int b = 0; b < 257; ++b) bits += (uint64_t)lf[b] * ll[b];
  const size_t raw_total = head_len + n_bytes;
  const uint64_t stored_bytes = raw_total + 5 * ((raw_total + 65534) / 65535 + (raw_total == 0));
  if ((bits + 7) / 8 + 80 >= stored_bytes) {            // (+ ~80 bytes of block header) not smaller: the generic path stores it
    std::vector<uint32_t> none;
    const unsigned char *parts[2] = {head, src};
    const size_t lens[2] = {head_len, n_bytes};
    uint32_t flat_freq = 0;
    (void)flat_freq;
    // a token list that cannot win makes write_segment choose stored blocks: hand it the literals
    none.reserve(raw_total);
    for (size_t k = 0; k < head_len; ++k) none.push_back(head[k]);
    for (size_t k = 0; k < n_bytes; ++k) none.push_back(src[k]);
    return write_segment(none, parts, lens, 2, last, out, cap);
  }
  // block header: no distance codes (HDIST = 1 with a zero length), literal / length lengths run-length coded
  int hlit = 257;
  std::vector<uint8_t> seq(ll, ll + hlit);
  seq.push_back(0);
  struct CL { uint8_t sym, extra; };
  std::vector<CL> cl;
  for (size_t a = 0; a < seq.size();) {
    size_t j = a;
    while (j < seq.size() && seq[j] == seq[a]) ++j;
    size_t run = j - a;
    if (seq[a] == 0) {
      while (run >= 11) { const size_t r = std::min<size_t>(run, 138); cl.push_back({18, (uint8_t)(r - 11)}); run -= r; }
      if (run >= 3) { cl.push_back({17, (uint8_t)(run - 3)}); run = 0; }
      while (run--) cl.push_back({0, 0});
    } else {
      cl.push_back({seq[a], 0});
      --run;
      while (run >= 3) { const size_t r = std::min<size_t>(run, 6); cl.push_back({16, (uint8_t)(r - 3)}); run -= r; }
      while (run--) cl.push_back({seq[a], 0});
    }
    a = j;
  }
  uint32_t cf[19] = {0};
  for (const CL &c : cl) cf[c.sym]++;
  uint8_t cll[19];
  code_lengths(cf, 19, 7, cll);
  static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  int hclen = 19;
  while (hclen > 4 && cll[order[hclen - 1]] == 0) --hclen;
  uint16_t cc[19];
  canonical_codes(cll, 19, cc);
  BitWriter w(out, cap);
  w.put(last ? 1 : 0, 1); w.put(2, 2);
  w.put((uint32_t)(hlit - 257), 5); w.put(0, 5); w.put((uint32_t)(hclen - 4), 4);
  for (int k = 0; k < hclen; ++k) w.put(cll[order[k]], 3);
  for (const CL &c : cl) {
    w.put(cc[c.sym], cll[c.sym]);
    if (c.sym == 16) w.put(c.extra, 2); else if (c.sym == 17) w.put(c.extra, 3); else if (c.sym == 18) w.put(c.extra, 7);
  }
  for (size_t k = 0; k < head_len; ++k) w.put(lc[head[k]], ll[head[k]]);
  // body: needs (bits + 7) / 8 + slack bytes of room -- checked once, then unchecked 8-byte stores
  if ((size_t)(w.end - w.p) < (size_t)((bits + 7) / 8) + 32) return 0;
  uint32_t tab[256];
  for (int b = 0; b < 256; ++b) tab[b] = (uint32_t)lc[b] | ((uint32_t)ll[b] << 16);
  uint64_t acc = w.acc;
  int nb = w.n;
  unsigned char *p = w.p;
  auto flush = [&]() {                                  // nb < 64 on entry; writes whole bytes, keeps the rest
    memcpy(p, &acc, 8);
    const int by = nb >> 3;
    p += by;
    acc = by == 8 ? 0 : acc >> (8 * by);
    nb &= 7;
  };
  size_t k = 0;
  for (; k + 4 <= n_bytes; k += 4) {
    const uint32_t t0 = tab[src[k]], t1 = tab[src[k + 1]], t2 = tab[src[k + 2]], t3 = tab[src[k + 3]];
    uint64_t g = t0 & 0xFFFF;
    int gl = (int)(t0 >> 16);
    g |= (uint64_t)(t1 & 0xFFFF) << gl; gl += (int)(t1 >> 16);
    g |= (uint64_t)(t2 & 0xFFFF) << gl; gl += (int)(t2 >> 16);
    g |= (uint64_t)(t3 & 0xFFFF) << gl; gl += (int)(t3 >> 16);      // <= 60 bits
    if (nb + gl > 64) flush();                          // nb <= 7 afterwards: 7 + 60 > 64 is possible, so split the group then
    if (nb + gl > 64) {
      acc |= g << nb;
      const int took = 64 - nb;
      nb = 64;
      flush();
      acc = g >> took;
      nb = gl - took;
    } else {
      acc |= nb < 64 ? g << nb : 0;
      nb += gl;
    }
  }
  if (nb >= 8) flush();
  w.p = p; w.acc = acc; w.n = nb;
  for (; k < n_bytes; ++k) w.put(lc[src[k]], ll[src[k]]);
  w.put(lc[256], ll[256]);
  if (last) {
    w.align();
  } else {
    w.put(0, 3); w.align();
    w.byte(0); w.byte(0); w.byte(0xFF); w.byte(0xFF);
  }
  return w.overflow ? 0 : (size_t)(w.p - out);
}

}  // namespace fdef
}  // namespace imf
