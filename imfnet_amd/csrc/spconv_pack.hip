// Weight image of variant 6 ("split-f16": fp32 arithmetic emulated on the f16 matrix pipe).
//
// The fp32 MFMA (v_mfma_f32_16x16x4_f32) issues once per 32 cycles for 2 048 FLOP; the f16 MFMA
// (v_mfma_f32_16x16x32_f16) once per ~17 cycles for 16 384 FLOP.  Every fp32 operand is written as
// x = hi + lo with hi = f16(x), lo = f16(x - hi) (22-23 significant bits; f16 subnormals are kept by
// the converts and by the MFMA on gfx950, tools/ubench/mfma_f16_denorm.hip), and the product is
// formed as lo_a*hi_w + hi_a*lo_w + hi_a*hi_w with fp32 accumulation inside the MFMA: three f16
// MFMAs per 32 input channels instead of eight fp32 ones (~5x fewer matrix-pipe cycles) at fp32-class
// accuracy (the dropped lo*lo term is 2^-22 relative; measured end to end: max |dF| 3e-7 against an
// fp64-accumulated network, plain fp32 is 2e-7).  Inputs must stay below the f16 range (65 504) --
// guarded by IMF_FLAG_RANGE; the weights are split once, here, at pack time.  The kernels that consume the image:
// spconv_g.hip (LDS-DMA staging, level 0 and the image branch), spconv_w.hip (wave-split, coarse levels), head.hip,
// and -- diagnostic builds only -- the register-staged spconv_h3.hip.
#include "spconv_shared.h"

namespace imf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Packed image (same size as the fp32 one: two halves per weight):
//   [y][k][cc][q = 2 cb + h][lane][t],  ci = 32 cc + 16 (t>>2) + 4 (lane>>4) + (t&3),  co = y CW + 16 cb + (lane&15),
//   (the contraction index of the MFMA is free to permute: lane group q holds channels 4q..4q+3 and 16+4q..16+4q+3 so that
//   each of the two 16-byte gathers of a row is one contiguous 64-byte segment across the four lanes of that row)
//   h = 0: hi halves, h = 1: lo halves; one (q, lane) entry = 8 halves = one float4.
// max |w| of the kernel as float bits (non-negative floats order like unsigned integers)
__global__ void __launch_bounds__(256)
k_absmax_bits(const float *__restrict__ w, long long total, unsigned *__restrict__ out) {
  __shared__ unsigned part[4];
  unsigned m = 0u;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const unsigned b = __float_as_uint(fabsf(w[i]));
    if (b < 0x7F800000u) m = b > m ? b : m;          // ignore inf / NaN
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned v = __shfl_xor(m, o, 64);
    m = v > m ? v : m;
  }
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 1; q < 4; ++q) m = part[q] > m ? part[q] : m;
    atomicMax(out, m);
  }
}

// Power-of-two pre-scaling of the weight image: with s = 13 - floor(log2 max|w|) the scaled kernel peaks in
// [2^13, 2^14), so lo = f16(w' - hi) is a NORMAL f16 number for every |w| >= 2^-17 max|w| (unscaled, a typical
// trained kernel of magnitude 1e-2 has subnormal lo halves: absolute error 3e-8 per weight = 18 bits).  The scaling
// is exact; the kernels multiply the fp32 accumulators by 2^-s (trailer[1]), also exact.
__device__ __forceinline__ int weight_shift(unsigned absmax_bits) {
  if (absmax_bits == 0u) return 0;
  const int e = (int)(absmax_bits >> 23) - 127;          // floor(log2 max|w|) (subnormal maxima: e = -127)
  int s = 13 - e;
  return s < -40 ? -40 : (s > 100 ? 100 : s);
}

__global__ void __launch_bounds__(256)
k_pack_weights_h3(const float *__restrict__ w, int kvol, int cin, int cout, _Float16 *__restrict__ packed,
                  float *__restrict__ trailer) {
  const long long total = (long long)kvol * cin * cout;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int shift = weight_shift(__float_as_uint(trailer[0]));
  if (idx == 0) trailer[1] = ldexpf(1.f, -shift);
  const int CB = co_blk_of(cout), CW = 16 * CB, ncc = cin / 32;
  long long r = idx;
  const int t = r & 7; r >>= 3;
  const int lane = r & 63; r >>= 6;
  const int cb = r % CB; r /= CB;
  const int cc = r % ncc; r /= ncc;
  const int k = r % kvol; r /= kvol;
  const int y = (int)r;
  const int ci = cc * 32 + 16 * (t >> 2) + 4 * (lane >> 4) + (t & 3);
  const int co = y * CW + 16 * cb + (lane & 15);
  const float v = ldexpf(w[((long long)k * cin + ci) * cout + co], shift);
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  const long long q0 = ((((long long)y * kvol + k) * ncc + cc) * (2 * CB) + 2 * cb) * 64 + lane;
  packed[q0 * 8 + t] = hi;
  packed[(q0 + 64) * 8 + t] = lo;
}

// ---- variant 3 ("bf16x3"): fp32 operands carried EXACTLY by three bf16 parts ---------------------------------------
// w = p0 + p1 + p2 with p0 = bf16(w), p1 = bf16(w - p0), p2 = bf16(w - p0 - p1), round-to-nearest-even: the residuals are
// exact fp32 differences with <= 16 and <= 8 significant bits, so the third part closes the sum -- three 8-bit significands
// carry fp32's 24, and bf16 has fp32's exponent range (no pre-scaling, no range guard, unlike the split-f16 image).  The
// kernels multiply with six v_mfma_f32_16x16x32_bf16 per 32 channels (a0 w2, a1 w1, a2 w0, a0 w1, a1 w0, a0 w0: every
// term down to 2^-16 relative; the three dropped ones are <= 2^-26 |a| |w| together), fp32 accumulation in the MFMA.
// Image: [y][k][cc][q = 3 cb + part][lane][t] -- the split-f16 image's lane / channel mapping with three parts per
// column block: ci = 32 cc + 16 (t >> 2) + 4 (lane >> 4) + (t & 3), co = y CW + 16 cb + (lane & 15); 1.5 x the fp32 size.
__global__ void __launch_bounds__(256)
k_pack_weights_b3(const float *__restrict__ w, int kvol, int cin, int cout, __bf16 *__restrict__ packed) {
  const long long total = (long long)kvol * cin * cout;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int CB = co_blk_of(cout), CW = 16 * CB, ncc = cin / 32;
  long long r = idx;
  const int t = r & 7; r >>= 3;
  const int lane = r & 63; r >>= 6;
  const int cb = r % CB; r /= CB;
  const int cc = r % ncc; r /= ncc;
  const int k = r % kvol; r /= kvol;
  const int y = (int)r;
  const int ci = cc * 32 + 16 * (t >> 2) + 4 * (lane >> 4) + (t & 3);
  const int co = y * CW + 16 * cb + (lane & 15);
  const float v = w[((long long)k * cin + ci) * cout + co];
  const __bf16 p0 = (__bf16)v;
  const float r1 = v - (float)p0;
  const __bf16 p1 = (__bf16)r1;
  const __bf16 p2 = (__bf16)(r1 - (float)p1);
  const long long q0 = ((((long long)y * kvol + k) * ncc + cc) * (3 * CB) + 3 * cb) * 64 + lane;
  packed[q0 * 8 + t] = p0;
  packed[(q0 + 64) * 8 + t] = p1;
  packed[(q0 + 128) * 8 + t] = p2;
}

}  // namespace imf

using namespace imf;

extern "C" int64_t imf_packed_weight_floats_bf16x3(int kvol, int cin, int cout) {
  return (int64_t)kvol * cin * cout / 2 * 3;
}

extern "C" int imf_pack_weights_bf16x3(const float *w, int kvol, int cin, int cout, float *packed, void *stream) {
  IMF_REQUIRE(w && packed, "imf_pack_weights_bf16x3: null pointer");
  IMF_REQUIRE(kvol >= 1 && kvol <= IMF_MAX_KVOL, "imf_pack_weights_bf16x3: kvol=%d", kvol);
  IMF_REQUIRE(cin > 0 && cin % 32 == 0 && cout > 0 && cout % 32 == 0,
              "imf_pack_weights_bf16x3: cin=%d cout=%d must be multiples of 32", cin, cout);
  const long long total = (long long)kvol * cin * cout;
  k_pack_weights_b3<<<(unsigned)div_up(total, 256), 256, 0, (hipStream_t)stream>>>(w, kvol, cin, cout,
                                                                                   reinterpret_cast<__bf16 *>(packed));
  IMF_CHECK_LAUNCH("k_pack_weights_b3");
  return IMF_OK;
}

extern "C" int imf_pack_weights_split16(const float *w, int kvol, int cin, int cout, float *packed,
                                        void *stream) {
  IMF_REQUIRE(w && packed, "imf_pack_weights_split16: null pointer");
  IMF_REQUIRE(kvol >= 1 && kvol <= IMF_MAX_KVOL, "imf_pack_weights_split16: kvol=%d", kvol);
  IMF_REQUIRE(cin > 0 && cin % 32 == 0 && cout > 0 && cout % 32 == 0,
              "imf_pack_weights_split16: cin=%d cout=%d must be multiples of 32", cin, cout);
  const long long total = (long long)kvol * cin * cout;
  hipStream_t st = (hipStream_t)stream;
  float *trailer = packed + total;                       // [0] max |w| (bits), [1] 2^-shift; 64 floats reserved
  IMF_CHECK_HIP(hipMemsetAsync(trailer, 0, 64 * sizeof(float), st));
  const long long nb = div_up(total, 256 * 8);
  k_absmax_bits<<<(unsigned)(nb > 1024 ? 1024 : nb), 256, 0, st>>>(w, total, reinterpret_cast<unsigned *>(trailer));
  k_pack_weights_h3<<<(unsigned)div_up(total, 256), 256, 0, st>>>(w, kvol, cin, cout,
                                                                  reinterpret_cast<_Float16 *>(packed), trailer);
  IMF_CHECK_LAUNCH("k_pack_weights_h3");
  return IMF_OK;
}
