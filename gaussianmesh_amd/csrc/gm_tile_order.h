// gm_tile_order.h -- dispatch order of the blend kernels: list tiles by descending list length (longest-processing-time first;
// with index order the last 40 % of the forward kernel ran on a few hundred late starters).  One workgroup: counting sort on
// min(length, 8191) / 32.  Runs as the extra workgroup of the tile pass's scatter launch (gm_bucket.hip) or, when the ranges
// come from tile_ranges_kernel, as its own launch (gm_render.hip).  The order inside a bucket is whatever the atomics give:
// it only moves work in time, results do not depend on it.
#pragma once
#include "gm_common.h"

namespace gm {

// key(t): work measure of list tile t (list length for the forward, the deepest contributor of its pixels for the backward).
// scratch [tiles] (may be null ONLY when key(t) cannot change while this runs): a tile's bucket and its rank inside the bucket are
// taken ONCE and parked there between the counting and the placement pass.  With the work hint the key is read from memory that
// the blend kernels of other frames update concurrently; evaluated twice, the two passes could disagree, and a counting sort whose
// passes disagree is no permutation - list tiles dispatched twice and others never (found by the two-rank bench test once the loop
// ran two first halves ahead: a frame in ~10 came back with a few tiles unrendered).
template <int THREADS, class Key>
__device__ __forceinline__ void tile_order_by(Key key, int tiles, uint32_t* __restrict__ order,
                                              uint32_t* cnt /*[256] shared*/, uint32_t* wsum /*[THREADS / 64] shared*/,
                                              uint32_t* __restrict__ scratch = nullptr) {
  static_assert(THREADS >= 256 && THREADS % 64 == 0, "tile_order_block: 256 or more threads");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 256) cnt[threadIdx.x] = 0;
  __syncthreads();
  auto bucket = [&](int t) { return 255u - min(key(t) >> 5, 255u); };   // bucket 0 = most work
  for (int t = threadIdx.x; t < tiles; t += THREADS) {
    const uint32_t b = bucket(t), r = atomicAdd(&cnt[b], 1u);
    if (scratch) scratch[t] = b | (r << 8);                             // (tiles <= 65536: the rank fits 24 bits)
  }
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
  if (scratch) {                                                        // same thread, same t as in the counting pass: its own words
    for (int t = threadIdx.x; t < tiles; t += THREADS) { const uint32_t p = scratch[t]; order[cnt[p & 255u] + (p >> 8)] = (uint32_t)t; }
  } else {
    for (int t = threadIdx.x; t < tiles; t += THREADS) order[atomicAdd(&cnt[bucket(t)], 1u)] = (uint32_t)t;
  }
}

// Work hint (optional, gm_forward_1_geom's work_hint): list length is a poor predictor of a tile's blend time - a 12 k-entry
// list under an opaque surface saturates after ~200 entries while a 1.5 k-entry silhouette list is walked to its end - and
// the launch ends on the heavy tiles that started late.  What a tile cost in a RECENT frame of the same view stream is a
// good one.  hint[0] counts frames; hint[1 + t] = (frame & 0xFFF) << 20 | work of list tile t, written by the forward blend
// with atomicMax (work = entries evaluated by the busiest of the tile's waves).  Entries older than GM_HINT_MAX_AGE frames are
// ignored and cleared.  Frames in flight on other streams read and write the same buffer concurrently: whatever they see only
// moves work in time - as long as every tile's key is read ONCE (tile_order_by's scratch).
#define GM_HINT_MAX_AGE 64u
#define GM_HINT_WORK_MASK 0xFFFFFu
template <int THREADS>
__device__ __forceinline__ void tile_order_block(const uint2* __restrict__ ranges, int tiles, uint32_t* __restrict__ order,
                                                 uint32_t* cnt /*[256] shared*/, uint32_t* wsum /*[THREADS / 64] shared*/,
                                                 uint32_t* __restrict__ hint = nullptr, uint32_t* __restrict__ epoch_out = nullptr,
                                                 uint32_t* __restrict__ scratch = nullptr /*[tiles], required with a hint*/) {
  if (!hint) {
    tile_order_by<THREADS>([&](int t) { const uint2 r = ranges[t]; return r.y - r.x; }, tiles, order, cnt, wsum);
    return;
  }
  if (threadIdx.x == 0) wsum[0] = (atomicAdd(&hint[0], 1u) + 1u) & 0xFFFu;
  __syncthreads();
  const uint32_t now = wsum[0];
  __syncthreads();
  if (threadIdx.x == 0) *epoch_out = now;                    // the blend kernel of this frame tags its entries with it
  for (int t = threadIdx.x; t < tiles; t += THREADS) {       // forget what is too old to mean anything
    const uint32_t v = hint[1 + t];
    if (v != 0u && ((now - (v >> 20)) & 0xFFFu) > GM_HINT_MAX_AGE) hint[1 + t] = 0u;
  }
  __syncthreads();
  // key: 8 x (work + 1) for a tile with a hint (bucket = work / 4, saturating at ~1000 entries); min(length, 2040) without one
  // (a tile that has just come into view: ranked like a tile of length / 8 entries of work)
  tile_order_by<THREADS>([&](int t) -> uint32_t {
    const uint2 r = ranges[t];
    const uint32_t len = r.y - r.x, v = hint[1 + t];
    if (len == 0u) return 0u;
    return v != 0u ? min(((v & GM_HINT_WORK_MASK) + 1u) << 3, 8191u) : min(len, 2040u);
  }, tiles, order, cnt, wsum, scratch);
}

}  // namespace gm
