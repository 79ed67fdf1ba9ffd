// Shared helpers for libimfnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/imfnet_hip.h"

namespace imf {

void set_error(const char *fmt, ...);

#define IMF_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      imf::set_error(__VA_ARGS__);      \
      return IMF_EINVAL;                \
    }                                   \
  } while (0)

#define IMF_CHECK_LAUNCH(what)                                                   \
  do {                                                                           \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) {                                                      \
      imf::set_error("%s: %s", what, hipGetErrorString(e_));                     \
      return IMF_ELAUNCH;                                                        \
    }                                                                            \
  } while (0)

#define IMF_CHECK_HIP(expr)                                                      \
  do {                                                                           \
    hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess) {                                                      \
      imf::set_error("%s: %s", #expr, hipGetErrorString(e_));                    \
      return IMF_ELAUNCH;                                                        \
    }                                                                            \
  } while (0)

constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
constexpr int kCoordBits = 18;
constexpr int kCoordLim = 1 << (kCoordBits - 1);  // coordinates in [-2^17, 2^17)

__device__ __forceinline__ uint64_t pack_key(int b, int x, int y, int z) {
  return ((uint64_t)(uint32_t)b << (3 * kCoordBits)) | ((uint64_t)(x & 0x3FFFF) << (2 * kCoordBits)) |
         ((uint64_t)(y & 0x3FFFF) << kCoordBits) | (uint64_t)(z & 0x3FFFF);
}

__device__ __forceinline__ bool coord_in_range(int x, int y, int z) {
  return x >= -kCoordLim && x < kCoordLim && y >= -kCoordLim && y < kCoordLim && z >= -kCoordLim &&
         z < kCoordLim;
}

__device__ __forceinline__ uint32_t hash64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (uint32_t)k;
}

// First slot of a voxel key in the table of a level whose coordinates are multiples of 2^shift: LOCALITY-PRESERVING (round
// 3).  The 4 x 4 x 4 block of voxels a key belongs to is hashed to a 64-slot (1 KiB) window, the voxel's place inside the
// block picks the slot in the window (z fastest), collisions leave it (hash_step).  A rulebook thread's 27 neighbour
// probes then fall into a few 128-byte lines of one or two windows instead of 27 random lines, and neighbouring rows of a
// wavefront probe the same windows.  Which slot a voxel lands in has no bearing on any result (rows are ordered by first
// occurrence, not by slot).
__device__ __forceinline__ uint32_t hash_slot(uint64_t key, int shift, uint32_t capmask) {
  const uint64_t low = (3ull << shift) * (1ull | (1ull << kCoordBits) | (1ull << (2 * kCoordBits)));
  const uint32_t local = (uint32_t)(((key >> (2 * kCoordBits + shift)) & 3ull) << 4 | ((key >> (kCoordBits + shift)) & 3ull) << 2 |
                                    ((key >> shift) & 3ull));
  return ((hash64(key & ~low) << 6) + local) & capmask;
}

// A collision leaves the window with a key-dependent odd stride (double hashing): windows are filled in regular patterns
// (surface voxels are adjacent), so walking on linearly runs through long occupied stretches -- measured: the level-0 map
// of the pair 30 -> 93 us with linear probing on this layout.  Insert and find walk the same sequence.
__device__ __forceinline__ uint32_t hash_step(uint64_t key) { return (hash64(key) >> 3) | 1u; }

// Returns the slot that holds `key` after the call (inserting it if absent).
__device__ __forceinline__ uint32_t hash_insert(imf_slot *tab, uint32_t capmask, uint64_t key, int shift) {
  uint32_t s = hash_slot(key, shift, capmask);
  uint32_t step = 0;
  while (true) {
    unsigned long long prev =
        atomicCAS(reinterpret_cast<unsigned long long *>(&tab[s].key), (unsigned long long)kEmptyKey,
                  (unsigned long long)key);
    if (prev == kEmptyKey || prev == key) return s;
    if (!step) step = hash_step(key);
    s = (s + step) & capmask;
  }
}

// key-only table (keypoint membership sets)
__device__ __forceinline__ uint32_t hash_insert_key(uint64_t *keys, uint32_t capmask, uint64_t key) {
  uint32_t s = hash64(key) & capmask;
  while (true) {
    unsigned long long prev =
        atomicCAS(reinterpret_cast<unsigned long long *>(keys + s), (unsigned long long)kEmptyKey,
                  (unsigned long long)key);
    if (prev == kEmptyKey || prev == key) return s;
    s = (s + 1) & capmask;
  }
}

// One 16-byte load per probe: key and row of a slot arrive together (a hit costs no second random line).
__device__ __forceinline__ int hash_find(const imf_slot *__restrict__ tab, uint32_t capmask, uint64_t key, int shift) {
  uint32_t s = hash_slot(key, shift, capmask);
  uint32_t step = 0;
  while (true) {
    const uint4 v = *reinterpret_cast<const uint4 *>(tab + s);
    const uint64_t k = ((uint64_t)v.y << 32) | v.x;
    if (k == key) return (int)v.z;
    if (k == kEmptyKey) return -1;
    if (!step) step = hash_step(key);
    s = (s + step) & capmask;
  }
}

inline int64_t div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace imf
