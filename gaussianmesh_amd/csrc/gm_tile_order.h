// gm_tile_order.h -- dispatch order of the blend kernels: list tiles by descending list length (longest-processing-time first;
// with index order the last 40 % of the forward kernel ran on a few hundred late starters).  One workgroup: counting sort on
// min(length, 8191) / 32.  Runs as the extra workgroup of the tile pass's scatter launch (gm_bucket.hip) or, when the ranges
// come from tile_ranges_kernel, as its own launch (gm_render.hip).  The order inside a bucket is whatever the atomics give:
// it only moves work in time, results do not depend on it.
#pragma once
#include "gm_common.h"

namespace gm {

// key(t): work measure of list tile t (list length for the forward, the deepest contributor of its pixels for the backward)
template <int THREADS, class Key>
__device__ __forceinline__ void tile_order_by(Key key, int tiles, uint32_t* __restrict__ order,
                                              uint32_t* cnt /*[256] shared*/, uint32_t* wsum /*[THREADS / 64] shared*/) {
  static_assert(THREADS >= 256 && THREADS % 64 == 0, "tile_order_block: 256 or more threads");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 256) cnt[threadIdx.x] = 0;
  __syncthreads();
  auto bucket = [&](int t) { return 255u - min(key(t) >> 5, 255u); };   // bucket 0 = most work
  for (int t = threadIdx.x; t < tiles; t += THREADS) atomicAdd(&cnt[bucket(t)], 1u);
  __syncthreads();
  uint32_t v = threadIdx.x < 256 ? cnt[threadIdx.x] : 0u, incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t u = __shfl_up(incl, d);
    if (lane >= d) incl += u;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wave; w++) woff += wsum[w];
  __syncthreads();
  if (threadIdx.x < 256) cnt[threadIdx.x] = woff + incl - v;          // exclusive start of each bucket
  __syncthreads();
  for (int t = threadIdx.x; t < tiles; t += THREADS) order[atomicAdd(&cnt[bucket(t)], 1u)] = (uint32_t)t;
}

template <int THREADS>
__device__ __forceinline__ void tile_order_block(const uint2* __restrict__ ranges, int tiles, uint32_t* __restrict__ order,
                                                 uint32_t* cnt /*[256] shared*/, uint32_t* wsum /*[THREADS / 64] shared*/) {
  tile_order_by<THREADS>([&](int t) { const uint2 r = ranges[t]; return r.y - r.x; }, tiles, order, cnt, wsum);
}

}  // namespace gm
