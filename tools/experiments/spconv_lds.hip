// EXPERIMENT -- not part of the library build (round 2; see DESIGN.md section 6 "what the counters say").
// Result: 45.9 us per 32->32 level-0 layer, the same as the streaming kernel k_spconv_h3<2> (43-45 us) -- so neither
// the per-stage barriers nor the weight re-reads set that kernel's time.  Ablations of THIS kernel (tools/kernel_ab.sh,
// -DIMF_LDS_ABL=..): no MFMA 30.8 us, no gathers 27.8 us, no split 44.0 us, no epilogue 41.2 us, random rows instead
// of the neighbour table 74.5 us; prefetch depth 2 / 4 / 8: 45.9 / 45.9 / 51.9 us; XCD-contiguous blocks on/off: same.
// Kept as a record; to try it again add it to SRCS and call launch_spconv_h3_lds from imf_spconv_fwd.
// Sparse convolution, variant 6, for the layers whose WHOLE split-f16 kernel fits the LDS of a CU:
// 32 -> 32 channels, up to 27 offsets = 27 x 4 KiB = 108 KiB of the 160 KiB.
//
// k_spconv_h3 (spconv_h3.hip) streams 16 KiB weight stages through a double buffer: every stage costs a
// workgroup barrier, and with only 32 output columns a stage is 4 offsets x 6 MFMAs per wavefront -- the
// barriers and the weight re-reads (one full kernel per 64-row tile) set the time, not the matrix pipe or
// the gather.  Here one persistent workgroup per CU loads the kernel ONCE, and after that single barrier
// its 16 wavefronts never meet again: each takes 16-row blocks (tile, quarter) round-robin, reads its 27
// neighbour indices, and runs a 4-deep software pipeline of row gathers against B fragments read from LDS.
//
// Arithmetic: identical MFMA sequence per (row block, offset) as k_spconv_h3<2> -- lo*hi, hi*lo, hi*hi per
// 16-column block, offsets ascending -- so the two kernels agree bit for bit (offsets absent from a tile add
// exact zeros here and are skipped there).  Same epilogue (spconv_shared.h).
#include "spconv_shared.h"

#ifndef IMF_LDS_ABL
#define IMF_LDS_ABL 0   // timing experiments only (tools/lds_ablations.sh): 1 no MFMA, 2 no split, 4 no gathers, 8 no index loads, 16 no LDS reads, 32 no epilogue
#endif

namespace imf {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int kLdsKvol = 27;
constexpr int kLdsWaves = 16;
#ifndef IMF_LDS_DEPTH
#define IMF_LDS_DEPTH 4
#endif
constexpr int kDepth = IMF_LDS_DEPTH;   // row gathers in flight per wavefront

__device__ __forceinline__ void split8(const float4 &x0, const float4 &x1, f16x8 &hi, f16x8 &lo) {
  const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const _Float16 h = (_Float16)v[t];
    hi[t] = h;
    lo[t] = (_Float16)(v[t] - (float)h);
  }
}

}  // namespace

template <int USE>
__global__ void __launch_bounds__(64 * kLdsWaves)
k_spconv_h3_lds(const ConvParams p) {
  extern __shared__ float4 wl[];                     // [kvol][q = 2 cb + h][lane]: the packed image verbatim
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  long long slots_act = p.n_slots;
  if (p.n_out_dev) {                                 // capacity mode: only the tiles that hold rows
    slots_act = conv_slots(p, conv_rows(p));
    if (p.dyn_split_kvol && p.err && blockIdx.x == 0 && tid == 0 &&
        auto_split_rule(slots_act, p.cout, p.dyn_split_kvol, p.split_min_blocks, p.split_target) > 1)
      atomicOr(p.err, 16);                           // the exact path would have split this launch: flagged
  }
  int n_blk = (int)(slots_act / 16);                 // 16-row blocks
  int first = blockIdx.x * kLdsWaves + wave, step = gridDim.x * kLdsWaves, blk0 = 0;
  if (!p.no_xcd_swizzle && (gridDim.x & 7) == 0) {
    // Workgroup b runs on XCD b % 8, and every XCD has its own 4 MiB L2.  Rows are ordered by voxel key, so a
    // contiguous eighth of the row blocks gathers (mostly) from a contiguous eighth of the input matrix: XCD x
    // takes blocks [x n/8, (x+1) n/8) and its 32 workgroups interleave inside that range.
    const int xcd = blockIdx.x & 7, per = ((n_blk + 31) / 32) * 4;      // tile-aligned share
    blk0 = xcd * per;
    n_blk = n_blk < blk0 + per ? n_blk : blk0 + per;
    first = blk0 + (blockIdx.x >> 3) * kLdsWaves + wave;
    step = (gridDim.x >> 3) * kLdsWaves;
    if (blk0 + (int)(blockIdx.x >> 3) * kLdsWaves >= n_blk) return;
  } else if (blockIdx.x * kLdsWaves >= n_blk) {
    return;
  }

  {
    const float4 *src = reinterpret_cast<const float4 *>(p.w_packed);
    const int total = p.kvol * 256;
    constexpr int kPer = (kLdsKvol * 256 + 64 * kLdsWaves - 1) / (64 * kLdsWaves);   // 7: all loads in flight together
    float4 v[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int e = tid + i * 64 * kLdsWaves;
      v[i] = e < total ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int e = tid + i * 64 * kLdsWaves;
      if (e < total) wl[e] = v[i];
    }
  }
  __syncthreads();
  const float un = p.w_unscale ? *p.w_unscale : 1.f;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.in_a), (short)0,
                                                                      0x7FFFFFFF, 0x00020000);

#pragma unroll 1
  for (int blk = first; blk < n_blk; blk += step) {
    const int tile = blk >> 2, sub = blk & 3;
    const uint32_t mask = p.tile_mask[tile * IMF_MASK_WORDS];
    if (mask == 0u) continue;                        // padding tile
    const long long slot = (long long)blk * 16 + r16;

    // all neighbour indices of the block first (27 independent loads), then the pipelined gathers
    unsigned voff[kLdsKvol];
#pragma unroll
    for (int k = 0; k < kLdsKvol; ++k) {
      int irow = -1;
      if (IMF_LDS_ABL & 8) {
        if (k < p.kvol && ((mask >> k) & 1u)) irow = (int)((slot * 2654435761u + k * 40503u) % (unsigned)p.n_out);
      } else if (k < p.kvol && ((mask >> k) & 1u)) irow = p.nbr[(long long)k * p.n_slots + slot];
      // a row without an input at this offset reads beyond the 2 GiB window: zeros, no branch
      voff[k] = irow >= 0 ? (unsigned)irow * 128u + 32u * q4 : 0x80000000u;
    }

    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    float4 a[kDepth][2];
#pragma unroll
    for (int k = 0; k < kLdsKvol + kDepth - 1; ++k) {
      if (k < kLdsKvol) {
        if (IMF_LDS_ABL & 4) {
          a[k % kDepth][0] = make_float4(__uint_as_float(voff[k]), 1.f, 2.f, 3.f);
          a[k % kDepth][1] = make_float4(__uint_as_float(voff[k] + 1u), 1.f, 2.f, 3.f);
        } else {
        a[k % kDepth][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[k], 0, 0));
        a[k % kDepth][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[k] + 16u, 0, 0));
        }
      }
      const int c = k - (kDepth - 1);
      if (c < 0) continue;
      if (__ballot(voff[c] != 0x80000000u) == 0ull) continue;   // nothing under this offset for the 16 rows
      f16x8 ah, al;
      if (IMF_LDS_ABL & 2) {
        ah = __builtin_bit_cast(f16x8, a[c % kDepth][0]);
        al = __builtin_bit_cast(f16x8, a[c % kDepth][1]);
      } else {
        split8(a[c % kDepth][0], a[c % kDepth][1], ah, al);
      }
      const float4 *wk = wl + c * 256 + lane;
      f16x8 bh0, bl0, bh1, bl1;
      if (IMF_LDS_ABL & 16) {
        bh0 = bl0 = bh1 = bl1 = __builtin_bit_cast(f16x8, make_float4(un, (float)c, un, un));
      } else {
        bh0 = *reinterpret_cast<const f16x8 *>(wk), bl0 = *reinterpret_cast<const f16x8 *>(wk + 64);
        bh1 = *reinterpret_cast<const f16x8 *>(wk + 128), bl1 = *reinterpret_cast<const f16x8 *>(wk + 192);
      }
      if (IMF_LDS_ABL & 1) {   // keep the operands alive
        acc[0][0] += (float)bh0[0] + (float)bl0[1] + (float)ah[0] + (float)al[1];
        acc[1][0] += (float)bh1[0] + (float)bl1[1] + (float)ah[2] + (float)al[3];
        continue;
      }
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh1, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl1, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh1, acc[1], 0, 0, 0);
    }
    if (IMF_LDS_ABL & 32) {
      if (acc[0][0] + acc[1][1] == 12345.f) p.out[slot] = acc[0][2];
    } else {
      conv_epilogue<2>(p, acc, tile, 0, sub, r16, q4, un);
    }
  }
}

bool spconv_h3_lds_applies(const ConvParams &p, int split) {
  const bool off = getenv("IMF_NO_LDS_CONV") != nullptr;   // A/B switch, read per launch so a test can flip it
  return !off && split == 1 && p.c_a == 32 && p.c_b == 0 && p.cout == 32 && p.kvol > 1 && p.kvol <= kLdsKvol &&
         p.nbr && p.tile_mask && !p.tail_split && !p.tickets && p.n_slots / IMF_TILE_ROWS >= 512;
}

int launch_spconv_h3_lds(const ConvParams &p, hipStream_t st, int use) {
  const size_t lds = (size_t)p.kvol * 256 * sizeof(float4);
  static bool configured = false;
  if (!configured) {
    IMF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spconv_h3_lds<0>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kLdsKvol * 4096));
    IMF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spconv_h3_lds<1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kLdsKvol * 4096));
    configured = true;
  }
  int cus = 256;
  {
    static int cached = 0;
    if (!cached) {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) == hipSuccess &&
          hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
        cached = n;
      else
        cached = 256;
    }
    cus = cached;
  }
  const long long n_blk = p.n_slots / 16;
  const long long want = (n_blk + kLdsWaves - 1) / kLdsWaves;
  const unsigned grid = (unsigned)(want < cus ? want : cus);
  if (use == 1) k_spconv_h3_lds<1><<<grid, 64 * kLdsWaves, lds, st>>>(p);
  else          k_spconv_h3_lds<0><<<grid, 64 * kLdsWaves, lds, st>>>(p);
  return IMF_OK;
}

}  // namespace imf
