// gm_bucket.hip -- ordering of a forward: (depth, id) order of the visible Gaussians and the stable sort of the instance
// stream by list tile, in nine launches instead of the eighteen of round 1's 8-bit LSD passes (gm_sort.hip, which now
// only serves simple_knn's Morton sort).
//
// Replaces (reference, RAST = gaussian_renderer/diff_gaussian_rasterizater/cuda_rasterizer):
//   RAST/rasterizer_impl.cu:478-483  cub::DeviceRadixSort::SortPairs on (tile << 32 | depth) keys
//   RAST/rasterizer_impl.cu:407      cub::DeviceScan::InclusiveSum over tiles_touched (instance offsets)
//   RAST/rasterizer_impl.cu:116-138, :485  identifyTileRanges + the memset of ranges
// The reference's ordering (ascending tile, then ascending depth bits, ties in ascending Gaussian id - what a stable
// sort of instances emitted in id order gives) is produced exactly; tests compare point_list bit for bit.
//
// A. Depth order of the P Gaussians (keys = float bits of view-space z, 0xFFFFFFFF for culled ones):
//    the preprocess kernel leaves the instance total, the visible count, the range of coarse bins in use (256 atomic slots,
//    one per cache line) and a coarse histogram of the visible keys (key >> 20; GeomState::coarse);
//    1. MSD partition into <= 2048 buckets through a table built from that histogram (DepthMap below: buckets in proportion
//       to the keys a coarse bin holds, so the partition stays balanced when a background or a floater stretches the depth
//       range): bk_hist -> bk_scan -> bk_scatter (stable, visible keys only);
//    2. bucket_sort_kernel: one workgroup per bucket sorts its few hundred to few thousand entries by the key bits inside
//       the bucket entirely in LDS, one word per entry ((key - first key of the bucket) << 12 | position: unique, so no
//       pass has to be stable): a counting split on the top <= 10 bits of the bucket's range with LDS atomics, then every
//       entry counts the smaller words of its own bin (a handful); bins above 48 entries (piles of equal depths) fall back
//       to stable LSD passes of <= 8 bits (ballot matching on registers + one LDS copy).  It writes the final order and the
//       emission records in that order and adds the instance counts to the per-run totals the emission reads.  A bucket
//       that does not fit the LDS (> 4096 entries) takes a slow, still exact, path through global memory.
//    Four launches and two passes over 8 B/Gaussian instead of twelve launches and four passes.
// B. Instances (emitted by duplicate_kernel in that order, keyed by list tile id | child mask << 16):
//    list tiles <= 2048 (1080p with 32-px parents: 2040; 4K with 64-px parents: 2040): ONE stable pass on an 11-bit digit,
//    bk_hist -> bk_scan -> bk_scatter; the scan's exclusive digit bases ARE the tile ranges, so identifyTileRanges and
//    its memset disappear.  More list tiles: two 8-bit passes + tile_ranges_kernel.
//
// Streams are (key, value) pairs interleaved as uint2 (one 8-byte access per entry everywhere).
//
// One pass = three kernels, no workgroup ever waits for another one of the same launch:
//   bk_hist     workgroup b counts the digits of its 4096 keys -> hist[b][digit] (one contiguous row), and adds the row to
//               chunk_total[b / 32][digit] and its 256-digit group sums to one of 64 slots per group with atomics (<= 32
//               resp. ~12 atomics per address; the accumulators are zeroed by the launch before: the forward's first memset
//               for the depth partition, duplicate_kernel for the tile pass)
//   bk_scan     workgroup (c, g): digits [256 g, 256 g + 256), rows [32 c, 32 c + 32): digit totals and the base of chunk c
//               from the chunk totals (<= n / 131072 coalesced loads), exclusive scan over the group's digits, group base
//               from the slots, then the rows' prefixes -> hist[row][digit] = ABSOLUTE first output position; workgroups
//               with c == 0 also publish the digit bases (bucket starts / tile ranges)
//   bk_scatter  workgroup b ranks its keys stably (match-any from DB ballots + per-wave digit counters in LDS), sorts the
//               tile by digit inside LDS (the counters' memory is reused as the stage) and streams it out: digit d's run
//               goes to hist[b][d] + (position in run), so neighbouring lanes store neighbouring pairs.  Measured: 3 M
//               pairs stored one lane at a time (64 different lines per store instruction) run into the L2's transaction
//               rate (25-34 us per pass); larger tiles make longer runs (tile / 2048 pairs per digit).
#include "gm_common.h"
#include "gm_tile_order.h"

namespace gm {

#define BK_THREADS 256
#define BK_WAVES 4
#define BK_ROUNDS_MAX 16
#define BK_CHUNK GM_BK_CHUNK
#ifndef GM_TILE_PASS_WAVES
#define GM_TILE_PASS_WAVES 8        // workgroup of the one-pass tile sort: 8 waves x 1024 keys
#endif
#define BS_ROUNDS GM_BS_ROUNDS
#define BS_CAP (256 * BS_ROUNDS)     // entries a bucket may have for the in-LDS sort (256 threads x BS_ROUNDS)
#define BS_BINS 1024                 // bins of the in-LDS sort's counting split (= the words of wcnt)
static_assert(BS_BINS == BK_WAVES * 256, "the counting split's bins live in wcnt");
#define BS_BIN_MAX 48u               // largest bin the counting split accepts before the stable radix passes take over

struct DigitSpec { uint32_t sub, shift, mask; };       // digit(k) = ((k - sub) >> shift) & mask

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
  return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
  return v;
}

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* wsum /*[4] shared*/, uint32_t& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const uint32_t s = wsum[w];
    woff += (w < wave) ? s : 0;
    tot += s;
  }
  total = tot;
  return woff + incl - v;
}

// agent-scope accesses to memory that kernels of OTHER streams read or write while this one runs (the view stream's DepthPlan): they
// bypass the XCD-local L2s, which are not coherent with each other inside a launch
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Depth-bucket mapping (gm_common.h, GM_COARSE_*): coarse bin c = key >> 20 owns buckets [base[c], base[c] + nb[c]) and
// subdivides its 2^20 key values linearly among them.  Monotone in the key, so (bucket, key inside the bucket, id) order is
// (key, id) order.
struct DepthMap { uint16_t base[GM_COARSE_BINS]; uint16_t nb[GM_COARSE_BINS]; };     // 8 KiB of LDS
__device__ __forceinline__ uint32_t depth_bucket(const DepthMap& m, uint32_t key) {
  const uint32_t c = (key >> GM_COARSE_SHIFT) & (GM_COARSE_BINS - 1);
  return (uint32_t)m.base[c] + (((key & ((1u << GM_COARSE_SHIFT) - 1u)) * (uint32_t)m.nb[c]) >> GM_COARSE_SHIFT);
}
// Built by every workgroup of the partition's histogram launch (256 threads; workgroup 0 also publishes it: dmap[c] = base <<
// 16 | nb, bmap[bucket] = c, counters[VISIBLE / NBUCKETS / RENDERED]).  Each non-empty coarse bin gets
// max(1, count * GM_BUCKET_BUDGET / visible) buckets; should more than 2048 come out (hundreds of sparsely filled bins), every
// non-empty bin gets exactly one.  Returns the number of buckets (0: nothing visible).
__device__ __forceinline__ uint32_t block_build_depth_map(const uint32_t* __restrict__ slots, const uint32_t* __restrict__ coarse, DepthMap& m,
                                                          uint32_t* wsum /*[4]*/, uint32_t* s_tmp /*[4]*/, uint32_t* __restrict__ dmap,
                                                          uint32_t* __restrict__ bmap, uint32_t* __restrict__ counters, int publish) {
  // publish: 0 nothing, 1 table + bucket ranges + counters (the partition's histogram launch), 2 the table alone (direct placement:
  // the table is for LATER frames, this frame's counters come from direct_plan_kernel)
  constexpr int PER = GM_COARSE_BINS / BK_THREADS;                  // 8 consecutive coarse bins per thread
  {                                    // instance total (num_rendered), visible count, first / last coarse bin in use from the slots
    static_assert(GM_SLOTS == BK_THREADS, "one slot per thread");
    const uint4 sl = *reinterpret_cast<const uint4*>(slots + GM_SLOT_STRIDE * threadIdx.x);
    const uint32_t inst = wave_sum_u32(sl.x), vis = wave_sum_u32(sl.y), ncmin = wave_max_u32(sl.z), cmax = wave_max_u32(sl.w);
    uint4* part = reinterpret_cast<uint4*>(&m);                     // (the map's memory is free until the table is written)
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = make_uint4(inst, vis, ncmin, cmax);
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint4 a = part[0], b = part[1], c = part[2], d = part[3];
      s_tmp[0] = a.x + b.x + c.x + d.x; s_tmp[1] = a.y + b.y + c.y + d.y;
      s_tmp[2] = (GM_COARSE_BINS - 1u) - max(max(a.z, b.z), max(c.z, d.z)); s_tmp[3] = max(max(a.w, b.w), max(c.w, d.w));
    }
  }
  __syncthreads();
  const uint32_t cmin = s_tmp[2], cmax = s_tmp[3];
  uint32_t cnt[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) cnt[j] = 0;
  if (threadIdx.x * PER + PER > cmin && threadIdx.x * PER <= cmax) {   // only the threads whose bins can hold anything read the histogram
#pragma unroll
    for (int k = 0; k < GM_COARSE_COPIES; k++)
#pragma unroll
      for (int j = 0; j < PER; j++) cnt[j] += coarse[coarse_index(threadIdx.x * PER + j, k)];
  }
  // a bin next to an empty or much sparser one is usually only partly covered by the keys (the depth range of an object begins
  // or ends inside it, and surfaces pile up at their depth extremes): its keys sit in a fraction of its range, so it gets four
  // times its share.  Neighbour counts travel through the map's own memory.
  uint32_t* scratch = reinterpret_cast<uint32_t*>(&m);
#pragma unroll
  for (int j = 0; j < PER; j++) scratch[threadIdx.x * PER + j] = cnt[j];
  __syncthreads();
  const uint32_t visible = s_tmp[1];
  const uint32_t left = threadIdx.x ? scratch[threadIdx.x * PER - 1] : 0u;
  const uint32_t right = threadIdx.x + 1 < BK_THREADS ? scratch[threadIdx.x * PER + PER] : 0u;
  __syncthreads();                                                   // (the scratch is dead: the map may be written)
  // share of the bucket budget in proportion to the weights: deterministic float arithmetic in a fixed order (every workgroup
  // of every launch computes the same table)
  float wgt[PER], wpart = 0.f;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const uint32_t lo = j ? cnt[j - 1] : left, hi = j + 1 < PER ? cnt[j + 1] : right;
    wgt[j] = (float)cnt[j] * ((lo < cnt[j] / 8u || hi < cnt[j] / 8u) ? 4.0f : 1.0f);
    wpart += wgt[j];
  }
  {
    float ws = wpart;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) ws += __shfl_xor(ws, d);
    if ((threadIdx.x & 63) == 0) reinterpret_cast<float*>(wsum)[threadIdx.x >> 6] = ws;
  }
  __syncthreads();
  const float wtotal = ((reinterpret_cast<float*>(wsum)[0] + reinterpret_cast<float*>(wsum)[1]) + reinterpret_cast<float*>(wsum)[2]) +
                       reinterpret_cast<float*>(wsum)[3];
  __syncthreads();
  const float scale = (float)GM_BUCKET_BUDGET / (wtotal > 0.f ? wtotal : 1.f);
  uint32_t nbk[PER], nsum = 0, nonempty = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) {
    nbk[j] = cnt[j] ? max(1u, (uint32_t)(wgt[j] * scale)) : 0u;
    nsum += nbk[j]; nonempty += cnt[j] ? 1u : 0u;
  }
  (void)visible;
  uint32_t total_b;
  uint32_t excl = block_exclusive_scan_256(nsum, wsum, total_b);
  __syncthreads();
  if (total_b > (1u << GM_BUCKET_BITS)) {                            // (uniform) fall back to one bucket per non-empty coarse bin
#pragma unroll
    for (int j = 0; j < PER; j++) nbk[j] = cnt[j] ? 1u : 0u;
    excl = block_exclusive_scan_256(nonempty, wsum, total_b);
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const uint32_t c = threadIdx.x * PER + j;
    m.base[c] = (uint16_t)excl; m.nb[c] = (uint16_t)nbk[j];
    if (publish) st_agent(dmap + c, (excl << 16) | nbk[j]);
    excl += nbk[j];
  }
  __syncthreads();
  if (publish == 1) {
    // bucket -> coarse bin: the last bin whose first bucket is <= b (empty bins share their successor's first bucket)
    for (uint32_t bk = threadIdx.x; bk < total_b; bk += BK_THREADS) {
      uint32_t lo = 0, hi = GM_COARSE_BINS;                          // invariant: base[lo] <= bk, (hi == BINS or base[hi] > bk)
      while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint32_t)m.base[mid] <= bk) lo = mid; else hi = mid;
      }
      // the bucket's key range: sub-range j of coarse bin lo, [lo << 20 | ceil(j 2^20 / nb), lo << 20 | ceil((j + 1) 2^20 / nb));
      // bucket_sort_kernel sorts on (key - first key) with as many bits as the range needs
      const uint32_t cn = max((uint32_t)m.nb[lo], 1u), j = bk - (uint32_t)m.base[lo];
      const uint32_t k0 = (uint32_t)((((unsigned long long)j << GM_COARSE_SHIFT) + cn - 1u) / cn);
      const uint32_t k1 = (uint32_t)((((unsigned long long)(j + 1u) << GM_COARSE_SHIFT) + cn - 1u) / cn);
      const uint32_t width = k1 > k0 ? k1 - k0 : 1u;
      reinterpret_cast<uint2*>(bmap)[bk] = make_uint2((lo << GM_COARSE_SHIFT) + k0, width > 1u ? 32u - (uint32_t)__clz((int)(width - 1u)) : 0u);
    }
    if (threadIdx.x == 0) {
      counters[GM_CNT_VISIBLE] = visible; counters[GM_CNT_NBUCKETS] = total_b; counters[GM_CNT_RENDERED] = s_tmp[0];
      counters[GM_CNT_CMIN] = cmin; counters[GM_CNT_CMAX] = cmax;
    }
  }
  return visible ? total_b : 0u;
}
__device__ __forceinline__ void block_load_depth_map(const uint32_t* __restrict__ dmap, const uint32_t* __restrict__ counters, DepthMap& m) {
  const uint32_t cmin = counters[GM_CNT_CMIN], cmax = min(counters[GM_CNT_CMAX], (uint32_t)GM_COARSE_BINS - 1u);
  uint32_t* z = reinterpret_cast<uint32_t*>(&m);
  for (uint32_t i = threadIdx.x; i < sizeof(DepthMap) / 4; i += blockDim.x) z[i] = 0u;     // (culled keys index the last bin: bucket 0, unused)
  __syncthreads();
  for (uint32_t c = cmin + threadIdx.x; c <= cmax; c += blockDim.x) {       // no visible key maps outside the bins in use
    const uint32_t e = dmap[c];
    m.base[c] = (uint16_t)(e >> 16); m.nb[c] = (uint16_t)(e & 0xFFFFu);
  }
  __syncthreads();
}
template <bool MSD> struct MapStorage { DepthMap m; };
template <> struct MapStorage<false> { char unused; };

// key of entry idx: MSD input is the plain key array of the preprocess kernel, everything else is a pair stream
template <bool MSD>
__device__ __forceinline__ uint32_t load_key(const void* __restrict__ in, uint32_t idx) {
  return MSD ? reinterpret_cast<const uint32_t*>(in)[idx] : reinterpret_cast<const uint2*>(in)[idx].x;
}

// ---------------------------------------------------------------------------------------------
// acc: [GM_ACC_SLOTS] group slots ([group][64]) followed by chunk_total [chunks][ND]; zero on entry.
// WAVES waves of 64 threads per workgroup, 1024 keys per wave: the depth partition (1 M keys) uses 4-wave workgroups so
// that the launch covers the chip; the tile pass (millions of instances) 16-wave workgroups: 4x fewer histogram rows,
// atomics and scan work per key.
template <bool MSD, int DB, int WAVES, int ROUNDS>
__global__ __launch_bounds__(WAVES * 64) void bk_hist_kernel(const void* __restrict__ in, uint32_t n_host,
                                                              const uint32_t* __restrict__ n_dev, DigitSpec ds,
                                                              const uint32_t* __restrict__ slots, uint32_t* __restrict__ hist,
                                                              uint32_t* __restrict__ acc, uint32_t* __restrict__ counters,
                                                              const uint32_t* __restrict__ coarse, uint32_t* __restrict__ dmap,
                                                              uint32_t* __restrict__ bmap, const FrameOfs go, const FrameOfs bo) {
  constexpr int ND = 1 << DB, THREADS = WAVES * 64, TILE = WAVES * ROUNDS * 64;
  // frame blockIdx.z of a batch (gm_common.h FrameOfs): the depth partition lives in the geometry buffer, the tile pass in the binning buffer
  in = frame_ptr(in, MSD ? go : bo); hist = frame_ptr(hist, MSD ? go : bo); acc = frame_ptr(acc, MSD ? go : bo);
  n_dev = frame_ptr(n_dev, go); slots = frame_ptr(slots, go); counters = frame_ptr(counters, go); coarse = frame_ptr(coarse, go);
  dmap = frame_ptr(dmap, go); bmap = frame_ptr(bmap, go);
  __shared__ uint32_t h[ND];
  __shared__ uint32_t s_tmp[4];
  __shared__ uint32_t s_w[4];
  __shared__ MapStorage<MSD> s_map;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t n = n_host;
  if (n_dev) n = min(n, *n_dev);
  if constexpr (MSD) {               // (the launch's extra, last workgroup only publishes the mapping, the bucket count and num_rendered)
    if (block_build_depth_map(slots, coarse, s_map.m, s_w, s_tmp, dmap, bmap, counters, blockIdx.x == gridDim.x - 1 ? 1 : 0) == 0u) return;
  }
  auto digit = [&](uint32_t key) -> uint32_t {
    if constexpr (MSD) return depth_bucket(s_map.m, key);
    else return ((key - ds.sub) >> ds.shift) & ds.mask;
  };
  const uint32_t nblk = (n + TILE - 1) / TILE;
  if (blockIdx.x >= nblk) return;
  for (int d = threadIdx.x; d < ND; d += THREADS) h[d] = 0;
  __syncthreads();
  const uint32_t wbase = blockIdx.x * TILE + wave * (ROUNDS * 64);
  uint32_t k[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const uint32_t idx = wbase + r * 64 + lane;
    k[r] = idx < n ? load_key<MSD>(in, idx) : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const uint32_t idx = wbase + r * 64 + lane;
    const bool valid = idx < n && (!MSD || k[r] != 0xFFFFFFFFu);
    if (valid) atomicAdd(&h[digit(k[r])], 1u);
  }
  __syncthreads();
  uint32_t* chunk_total = acc + GM_ACC_SLOTS + (size_t)(blockIdx.x / BK_CHUNK) * ND;
  for (int d = threadIdx.x; d < ND; d += THREADS) {   // a wave's 64 digits lie in one 256-digit group
    const uint32_t c = h[d];
    hist[(size_t)blockIdx.x * ND + d] = c;
    if (c) atomicAdd(&chunk_total[d], c);
    const uint32_t gs = wave_sum_u32(c);
    if (lane == 0 && gs) atomicAdd(&acc[(d >> 8) * 64 + ((blockIdx.x * WAVES + wave) & 63)], gs);
  }
}

// ---------------------------------------------------------------------------------------------
// grid = (chunks, ND / 256).  base_out (optional): [ND + 1] exclusive digit bases; ranges_out (optional): [nranges] uint2
// {first, one past last} per digit value, {0, 0} for an empty one.
template <bool MSD, int DB>
__global__ __launch_bounds__(BK_THREADS) void bk_scan_kernel(uint32_t* __restrict__ hist, uint32_t tile, uint32_t n_host, const uint32_t* __restrict__ n_dev,
                                                              const uint32_t* __restrict__ acc,
                                                              uint32_t* __restrict__ base_out, const uint32_t* __restrict__ counters,
                                                              uint2* __restrict__ ranges_out, uint32_t nranges, const FrameOfs go,
                                                              const FrameOfs bo, const FrameOfs io) {
  constexpr int ND = 1 << DB, NG = ND / 256;
  hist = frame_ptr(hist, MSD ? go : bo); acc = frame_ptr(acc, MSD ? go : bo);
  n_dev = frame_ptr(n_dev, go); base_out = frame_ptr(base_out, go); counters = frame_ptr(counters, go); ranges_out = frame_ptr(ranges_out, io);
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t gsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t n = n_host;
  if (n_dev) n = min(n, *n_dev);
  if (MSD && counters[GM_CNT_VISIBLE] == 0u) n = 0;          // (written by the histogram launch before this one)
  const uint32_t nblk = (n + tile - 1) / tile;
  const uint32_t nchunks = (nblk + BK_CHUNK - 1) / BK_CHUNK;
  const uint32_t c = blockIdx.x, g = blockIdx.y;
  const uint32_t d = g * 256 + threadIdx.x;
  if (nblk == 0) {                                  // nothing to sort: every list is empty
    if (c == 0) {
      if (base_out) { base_out[d] = 0; if (d == 0) base_out[ND] = 0; }
      if (ranges_out && d < nranges) ranges_out[d] = make_uint2(0u, 0u);
    }
    return;
  }
  if (c >= nchunks) return;
  const uint32_t* chunk_total = acc + GM_ACC_SLOTS;
  uint32_t tot = 0, cbase = 0;
  for (uint32_t c0 = 0; c0 < nchunks; c0 += 8) {      // batches of independent loads
    uint32_t t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) t[j] = c0 + j < nchunks ? chunk_total[(size_t)(c0 + j) * ND + d] : 0u;
#pragma unroll
    for (int j = 0; j < 8; j++) { tot += t[j]; cbase += (c0 + j < c) ? t[j] : 0u; }
  }
  uint32_t below = 0;                                 // keys in the digit groups before this one
  {
    uint32_t part = 0;
#pragma unroll
    for (int j = 0; j < (NG * 64 + 255) / 256; j++) {
      const uint32_t i = j * 256 + threadIdx.x;
      if (i < NG * 64 && (i >> 6) < g) part += acc[i];
    }
    part = wave_sum_u32(part);
    if (lane == 0) gsum[wave] = part;
  }
  uint32_t gtotal;
  const uint32_t excl = block_exclusive_scan_256(tot, wsum, gtotal);    // (its barrier also publishes gsum)
  below = gsum[0] + gsum[1] + gsum[2] + gsum[3];
  const uint32_t base = below + excl;
  if (c == 0) {
    if (base_out) { base_out[d] = base; if (d == ND - 1) base_out[ND] = base + tot; }
    if (ranges_out && d < nranges) {
      const bool refused = !MSD && counters[GM_CNT_REFUSED] != 0u;      // emission refused: every list stays empty
      ranges_out[d] = (refused || tot == 0u) ? make_uint2(0u, 0u) : make_uint2(base, base + tot);   // empty lists: {0, 0} as in the reference
    }
  }
  const uint32_t r0 = c * BK_CHUNK;
  const uint32_t nr = min((uint32_t)BK_CHUNK, nblk - r0);
  uint32_t v[BK_CHUNK];
#pragma unroll
  for (int r = 0; r < BK_CHUNK; r++) v[r] = (uint32_t)r < nr ? hist[(size_t)(r0 + r) * ND + d] : 0u;
  uint32_t run = base + cbase;
#pragma unroll
  for (int r = 0; r < BK_CHUNK; r++) {
    if ((uint32_t)r < nr) hist[(size_t)(r0 + r) * ND + d] = run;
    run += v[r];
  }
}

// ---------------------------------------------------------------------------------------------
// zero_acc / zero_words: accumulators of the NEXT pass (two-pass tile sort), cleared here because no earlier launch can
template <bool MSD, int DB, int WAVES, int ROUNDS>
__global__ __launch_bounds__(WAVES * 64, 4) void bk_scatter_kernel(const void* __restrict__ in, uint2* __restrict__ out,
                                                                 uint32_t n_host, const uint32_t* __restrict__ n_dev, DigitSpec ds,
                                                                 const uint32_t* __restrict__ hist,
                                                                 uint32_t* __restrict__ zero_acc, uint32_t zero_words,
                                                                 const uint2* __restrict__ ord_ranges, int ord_tiles,
                                                                 uint32_t* __restrict__ ord_out, const uint32_t* __restrict__ dmap,
                                                                 const uint32_t* __restrict__ counters, uint32_t* __restrict__ ord_hint,
                                                                 uint32_t* __restrict__ ord_epoch, unsigned long long* __restrict__ trace,
                                                                 uint32_t* __restrict__ ord_scratch, const FrameOfs go, const FrameOfs bo,
                                                                 const FrameOfs io) {
  constexpr int ND = 1 << DB, THREADS = WAVES * 64, TILE = WAVES * ROUNDS * 64;
  in = frame_ptr(in, MSD ? go : bo); out = frame_ptr(out, MSD ? go : bo); hist = frame_ptr(hist, MSD ? go : bo); zero_acc = frame_ptr(zero_acc, bo);
  n_dev = frame_ptr(n_dev, go); dmap = frame_ptr(dmap, go); counters = frame_ptr(counters, go);
  ord_ranges = frame_ptr(ord_ranges, io); ord_out = frame_ptr(ord_out, io); ord_epoch = frame_ptr(ord_epoch, io); ord_scratch = frame_ptr(ord_scratch, io);
  const unsigned long long t_begin = trace ? wall_clock64() : 0ull;      // (tools/pipeline_trace.py: per-workgroup start / end)
  if (!MSD && ord_out && blockIdx.x == gridDim.x - 1) {      // the launch's extra workgroup: dispatch order of the blend kernels
    __shared__ uint32_t o_cnt[256];                          // from the ranges the scan kernel has just published
    __shared__ uint32_t o_wsum[WAVES];
    tile_order_block<THREADS>(ord_ranges, ord_tiles, ord_out, o_cnt, o_wsum, ord_hint, ord_epoch, ord_scratch);
    return;
  }
  // wcnt: per-wave running digit counts (<= 1024) -> per-wave exclusive offsets (< TILE <= 16384); once every key knows its
  // position in the digit-sorted tile the same memory stages the tile (one 8-byte pair per key)
  constexpr int SMEM = (WAVES * ND * 2 > TILE * 8) ? WAVES * ND * 2 : TILE * 8;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  uint16_t (*wcnt)[ND] = reinterpret_cast<uint16_t (*)[ND]>(smem);
  uint2* stage = reinterpret_cast<uint2*>(smem);
  __shared__ uint32_t gbase[ND];             // first output position of each digit for this workgroup
  __shared__ uint16_t dstart[ND];            // start of each digit's run inside the digit-sorted tile (< TILE <= 16384)
  __shared__ uint32_t wsum[WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (zero_acc)
    for (uint32_t i = blockIdx.x * THREADS + threadIdx.x; i < zero_words; i += gridDim.x * THREADS) zero_acc[i] = 0u;
  uint32_t n = n_host;
  if (n_dev) n = min(n, *n_dev);
  __shared__ MapStorage<MSD> s_map;
  if constexpr (MSD) {
    if (counters[GM_CNT_VISIBLE] == 0u) return;
    block_load_depth_map(dmap, counters, s_map.m);
  }
  auto digit = [&](uint32_t key) -> uint32_t {
    if constexpr (MSD) return depth_bucket(s_map.m, key);
    else return ((key - ds.sub) >> ds.shift) & ds.mask;
  };
  const uint32_t nblk = (n + TILE - 1) / TILE;
  if (blockIdx.x >= nblk) return;
  for (int d = threadIdx.x; d < ND; d += THREADS) {
#pragma unroll
    for (int w = 0; w < WAVES; w++) wcnt[w][d] = 0;
    gbase[d] = hist[(size_t)blockIdx.x * ND + d];
  }
  __syncthreads();

  const uint32_t wbase = blockIdx.x * TILE + wave * (ROUNDS * 64);
  uint32_t key[ROUNDS], val[ROUNDS], rank[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const uint32_t idx = wbase + r * 64 + lane;
    if (MSD) {
      key[r] = idx < n ? reinterpret_cast<const uint32_t*>(in)[idx] : 0xFFFFFFFFu;
      val[r] = idx;
    } else {
      const uint2 kv = idx < n ? reinterpret_cast<const uint2*>(in)[idx] : make_uint2(0xFFFFFFFFu, 0u);
      key[r] = kv.x; val[r] = kv.y;
    }
  }
  uint32_t vmask = 0;                          // bit r: this lane's key of round r takes part
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const uint32_t idx = wbase + r * 64 + lane;
    const bool valid = idx < n && (!MSD || key[r] != 0xFFFFFFFFu);
    vmask |= valid ? (1u << r) : 0u;
    const uint32_t d = digit(key[r]);
    uint64_t peers = __ballot(valid);          // match-any over the digit bits among the valid lanes
#pragma unroll
    for (int b = 0; b < DB; b++) {
      const uint64_t bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t before = lanes_below(peers);
    const int leader = __ffsll((unsigned long long)peers) - 1;
    uint32_t old = 0;
    if (valid && lane == leader) {
      old = wcnt[wave][d];
      wcnt[wave][d] = (uint16_t)(old + __popcll(peers));
    }
    old = __shfl(old, leader < 0 ? 0 : leader);
    rank[r] = old + before;
    __builtin_amdgcn_wave_barrier();           // keep the per-wave LDS counter updates of successive rounds in order
  }
  __syncthreads();
  uint32_t tile_n;
  {  // per digit: per-wave counts -> per-wave exclusive offsets; digit counts -> start of each digit's run in the sorted tile
    constexpr int DPT = (ND + THREADS - 1) / THREADS;            // consecutive digits per thread (ND >= THREADS here or DPT == 1)
    uint32_t dc[DPT], sum = 0;
#pragma unroll
    for (int j = 0; j < DPT; j++) {
      const int d = threadIdx.x * DPT + j;
      uint32_t run = 0;
      if (d < ND) {
#pragma unroll
        for (int w = 0; w < WAVES; w++) {
          const uint32_t c = wcnt[w][d];
          wcnt[w][d] = (uint16_t)run;
          run += c;
        }
      }
      dc[j] = run; sum += run;
    }
    uint32_t incl = sum;                                       // workgroup-wide exclusive scan of `sum`
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
      const uint32_t t = __shfl_up(incl, dd);
      if (lane >= dd) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) {
      const uint32_t t = wsum[w];
      woff += (w < wave) ? t : 0;
      tot += t;
    }
    tile_n = tot;
    uint32_t excl = woff + incl - sum;
#pragma unroll
    for (int j = 0; j < DPT; j++) {
      const int d = threadIdx.x * DPT + j;
      if (d < ND) dstart[d] = (uint16_t)excl;
      excl += dc[j];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {        // position of each key inside the digit-sorted tile
    const uint32_t d = digit(key[r]);
    rank[r] += ((vmask >> r) & 1u) ? dstart[d] + wcnt[wave][d] : 0u;
  }
  __syncthreads();                             // wcnt is dead from here: its memory becomes the stage
#pragma unroll
  for (int r = 0; r < ROUNDS; r++)
    if ((vmask >> r) & 1u) stage[rank[r]] = make_uint2(key[r], val[r]);
  __syncthreads();
  // digit d's run goes to gbase[d] + (position in run): neighbouring lanes store neighbouring pairs
  for (uint32_t i = threadIdx.x; i < tile_n; i += THREADS) {
    const uint2 kv = stage[i];
    const uint32_t d = digit(kv.x);
    out[gbase[d] + (i - dstart[d])] = kv;
  }
  if (trace && threadIdx.x == 0) { trace[3 * blockIdx.x] = t_begin; trace[3 * blockIdx.x + 1] = wall_clock64(); trace[3 * blockIdx.x + 2] = tile_n; }
}

// ---------------------------------------------------------------------------------------------
// Instances of 64 consecutive sorted positions (first one `pos0`, lane l holds position pos0 + l's count) into the per-run
// totals duplicate_kernel builds its output offsets from: the 64 positions touch at most two runs of GM_SCAN_ITEMS.
__device__ __forceinline__ void chunk_add(uint32_t* __restrict__ chunk_inst, uint32_t pos0, int lane, uint32_t count) {
  const uint32_t ch0 = pos0 / GM_SCAN_ITEMS;
  const bool first = (pos0 + (uint32_t)lane) / GM_SCAN_ITEMS == ch0;
  const uint32_t s0 = wave_sum_u32(first ? count : 0u), s1 = wave_sum_u32(first ? 0u : count);
  if (lane == 0) {
    if (s0) atomicAdd(chunk_inst + ch0, s0);
    if (s1) atomicAdd(chunk_inst + ch0 + 1, s1);
  }
}

// One workgroup per bucket of the MSD partition: (key, id) pairs [start, end) of p1 -> final order.
// Outputs: order0[start..end) = ids in (key, id) order, bin_sorted[start..end) = emission records of those ids,
// chunk_inst[run] += instance counts of the sorted positions of that run.  p0[start..end) is scratch for the slow path.
// DIRECT (direct depth placement): bucket b's entries are the first (end - start) of its slab, p1 + b * cap and bins + b * cap, in
// ARRIVAL order (the preprocess kernel's atomics), so position says nothing about the id: the bucket's key range is taken from the
// entries themselves (the table that placed them is an earlier frame's), equal keys are ordered by comparing ids, and what the
// one-word sort cannot do - a key range above 20 bits, a pile of more than BS_BIN_MAX equal keys - refuses the frame
// (counters[GM_CNT_DIRECT_FAIL]; the caller renders it again on the partition path).  The record is read from the workgroup's own
// slab: contiguous memory, every fetched line used.
template <bool DIRECT>
__global__ __launch_bounds__(BK_THREADS) void bucket_sort_kernel(uint32_t* __restrict__ counters,
                                                                  const uint32_t* __restrict__ bmap, const uint32_t* __restrict__ bucket_start,
                                                                  uint2* __restrict__ p1, uint2* __restrict__ p0, uint32_t* __restrict__ order0,
                                                                  const uint32_t* __restrict__ tiles, const uint4* __restrict__ bins,
                                                                  uint4* __restrict__ bin_sorted, uint32_t* __restrict__ chunk_inst,
                                                                  unsigned long long* __restrict__ trace, uint32_t cap,
                                                                  const uint32_t* __restrict__ slots, const uint32_t* __restrict__ coarse,
                                                                  const uint32_t* __restrict__ hdr, uint32_t* __restrict__ plan, const FrameOfs go) {
  const unsigned long long t_begin = trace ? wall_clock64() : 0ull;
  if (!DIRECT) {                                 // frame blockIdx.z of a batch: everything lives in the geometry buffer (the direct placement is not batched)
    counters = frame_ptr(counters, go); bmap = frame_ptr(bmap, go); bucket_start = frame_ptr(bucket_start, go); p1 = frame_ptr(p1, go); p0 = frame_ptr(p0, go);
    order0 = frame_ptr(order0, go); tiles = frame_ptr(tiles, go); bins = frame_ptr(bins, go); bin_sorted = frame_ptr(bin_sorted, go);
    chunk_inst = frame_ptr(chunk_inst, go);
  }
  __shared__ uint32_t wcnt[BK_WAVES][256];
  __shared__ uint32_t dstart[256];
  __shared__ uint32_t lkey[BS_CAP];                                 // (key - first key) << 12 | position in the bucket: 16 KiB
  __shared__ uint32_t wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (DIRECT && blockIdx.x == gridDim.x - 1) {
    // the launch's extra workgroup, beside the sorting ones: this frame's coarse histogram -> the table the stream's LATER frames place
    // their Gaussians with (slot = this frame's sequence number % slots; stamp 0 while the words change, the sequence number after)
    static_assert(sizeof(DepthMap) <= sizeof(lkey), "the builder's map lives in the sort's key array");
    DepthMap& m = *reinterpret_cast<DepthMap*>(lkey);
    const uint32_t seq = hdr[0], slot = seq % GM_PLAN_SLOTS;
    uint32_t* stamp = plan + 2 + slot;
    if (threadIdx.x == 0) st_agent(stamp, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // (the store has left before the table words do)
    __syncthreads();
    block_build_depth_map(slots, coarse, m, wsum, dstart, plan + 2 + GM_PLAN_SLOTS + (size_t)slot * GM_COARSE_BINS, nullptr, nullptr, 2);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
      st_agent(stamp, seq);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      atomicMax(plan, seq);
    }
    return;
  }
  const uint32_t nb = counters[GM_CNT_VISIBLE] ? counters[GM_CNT_NBUCKETS] : 0u;
  const uint32_t b = blockIdx.x;
  uint32_t start = 0, end = 0;
  if (b < nb) { start = bucket_start[b]; end = bucket_start[b + 1]; }
  const uint32_t n = end - start;
  if (n == 0u) return;
  if (DIRECT) {
    if (counters[GM_CNT_DIRECT_FAIL] != 0u) return;                 // (the frame is refused already)
    p1 += (size_t)b * cap; bins += (size_t)b * cap; start = 0;      // the bucket's slab; `end - start` stays n, outputs go to obase + p
  }
  const uint32_t obase = DIRECT ? bucket_start[b] : start;
  uint2 krange = make_uint2(0u, 0u);
  if (!DIRECT) krange = reinterpret_cast<const uint2*>(bmap)[b];    // {first key of the bucket, bits of (key - first key) inside it}
  if (DIRECT) {                                                     // the range of the keys that arrived
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    for (uint32_t p = threadIdx.x; p < n; p += BK_THREADS) { const uint32_t k = p1[p].x; kmin = min(kmin, k); kmax = max(kmax, k); }
    kmin = ~wave_max_u32(~kmin); kmax = wave_max_u32(kmax);
    if (lane == 0) { dstart[wave] = kmin; dstart[4 + wave] = kmax; }
    __syncthreads();
    kmin = min(min(dstart[0], dstart[1]), min(dstart[2], dstart[3]));
    kmax = max(max(dstart[4], dstart[5]), max(dstart[6], dstart[7]));
    __syncthreads();
    const uint32_t width = kmax - kmin;
    krange = make_uint2(kmin, width ? 32u - (uint32_t)__clz((int)width) : 0u);
    if (krange.y > 20u) {                                           // (workgroup-uniform) the word has 20 bits for the key
      if (threadIdx.x == 0) counters[GM_CNT_DIRECT_FAIL] = 1u;
      return;
    }
  }
  DigitSpec ds;
  ds.sub = krange.x; ds.shift = 0; ds.mask = 0xFFFFFFFFu;
  const uint32_t low_bits = krange.y;
  const uint32_t npass = (low_bits + 7u) / 8u;
  const uint32_t pb = npass ? (low_bits + npass - 1u) / npass : 0u;
  if (n <= BS_CAP) {
    // wave w owns positions [w * rounds * 64, (w + 1) * rounds * 64) in rounds of 64: rank order == position order
    const uint32_t rounds = (n + 255u) / 256u;
    // What is sorted is ONE word per entry: the key relative to the bucket's first key (< 2^20: a bucket lies inside one coarse
    // bin) above the entry's position in the bucket (< 4096).  Half the LDS and half the traffic of (key, id) pairs; the id is
    // picked up from the bucket's own pair range at the end.  Equal keys keep their position order = ascending id.
    uint32_t key[BS_ROUNDS], rank[BS_ROUNDS];
#pragma unroll
    for (int r = 0; r < BS_ROUNDS; r++) {
      const uint32_t p = (wave * rounds + r) * 64u + lane;
      const bool valid = (uint32_t)r < rounds && p < n;
      key[r] = valid ? ((p1[start + p].x - ds.sub) << 12) | p : 0u;
    }
    // Fast path: the words are UNIQUE (position in the low bits), so any correct sort of them is THE order and no pass has to be
    // stable.  One counting split on the top <= 10 bits of the bucket's key range with LDS atomics (a key's slot inside its bin is
    // whatever the atomic returned), then every key counts the smaller words of its own bin - a bucket's keys are spread over its
    // range (the bucket map divides coarse bins linearly), so a bin holds a handful.  ~50 vector instructions per round of 64 keys
    // instead of ~220 per 8-bit ballot-matching pass.  Bins above BS_BIN_MAX entries (piles of equal depths): the stable passes below.
    bool done = false;
    {
      uint32_t* __restrict__ hist = &wcnt[0][0];                      // BS_BINS bins, then their first positions
      const uint32_t hb = min(low_bits, 10u), hs = 12u + low_bits - hb;
#pragma unroll
      for (int j = 0; j < BS_BINS / BK_THREADS; j++) hist[threadIdx.x + BK_THREADS * j] = 0u;
      if (threadIdx.x == 0) dstart[0] = 0u;
      __syncthreads();
#pragma unroll
      for (int r = 0; r < BS_ROUNDS; r++) {
        if ((uint32_t)r < rounds) {                   // wave-uniform
          const uint32_t p = (wave * rounds + r) * 64u + lane;
          if (p < n) rank[r] = atomicAdd(&hist[min(key[r] >> hs, (uint32_t)BS_BINS - 1u)], 1u);
        }
      }
      __syncthreads();
      {
        constexpr int PER = BS_BINS / BK_THREADS;
        uint32_t c[PER], sum = 0, mx = 0;
#pragma unroll
        for (int j = 0; j < PER; j++) { c[j] = hist[threadIdx.x * PER + j]; sum += c[j]; mx = max(mx, c[j]); }
        uint32_t tot;
        uint32_t excl = block_exclusive_scan_256(sum, wsum, tot);
#pragma unroll
        for (int j = 0; j < PER; j++) { hist[threadIdx.x * PER + j] = excl; excl += c[j]; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d));
        if (lane == 0) atomicMax(&dstart[0], mx);
      }
      __syncthreads();
      if (DIRECT && dstart[0] > BS_BIN_MAX) {          // a pile of equal depths: the stable passes below need id order on entry
        if (threadIdx.x == 0) counters[GM_CNT_DIRECT_FAIL] = 1u;
        return;
      }
      if (dstart[0] <= BS_BIN_MAX) {                   // workgroup-uniform
#pragma unroll
        for (int r = 0; r < BS_ROUNDS; r++) {
          if ((uint32_t)r < rounds) {
            const uint32_t p = (wave * rounds + r) * 64u + lane;
            if (p < n) lkey[hist[min(key[r] >> hs, (uint32_t)BS_BINS - 1u)] + rank[r]] = key[r];
          }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < BS_ROUNDS; r++) {
          if ((uint32_t)r < rounds) {
            const uint32_t p = (wave * rounds + r) * 64u + lane;
            if (p < n) {
              const uint32_t d = min(key[r] >> hs, (uint32_t)BS_BINS - 1u);
              const uint32_t lo = hist[d], hi = d + 1u < (uint32_t)BS_BINS ? hist[d + 1u] : n;
              uint32_t below = 0;
              for (uint32_t j = lo; j < hi; j++) {
                const uint32_t wj = lkey[j];
                bool less = wj < key[r];
                if (DIRECT && ((wj ^ key[r]) >> 12) == 0u && wj != key[r])        // equal keys: ascending id (rare)
                  less = p1[wj & 0xFFFu].y < p1[key[r] & 0xFFFu].y;
                below += less ? 1u : 0u;
              }
              rank[r] = lo + below;
            }
          }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < BS_ROUNDS; r++) {
          if ((uint32_t)r < rounds) {
            const uint32_t p = (wave * rounds + r) * 64u + lane;
            if (p < n) lkey[rank[r]] = key[r];
          }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < BS_ROUNDS; r++) {
          if ((uint32_t)r < rounds) {
            const uint32_t p = (wave * rounds + r) * 64u + lane;
            if (p < n) key[r] = lkey[p];
          }
        }
        done = true;
      }
      __syncthreads();
    }
    for (uint32_t pass = 0; !done && pass < npass; pass++) {
      const uint32_t lo = pass * pb, pmask = (1u << min(pb, low_bits - lo)) - 1u;
      wcnt[0][threadIdx.x] = 0; wcnt[1][threadIdx.x] = 0; wcnt[2][threadIdx.x] = 0; wcnt[3][threadIdx.x] = 0;
      __syncthreads();
#pragma unroll
      for (int r = 0; r < BS_ROUNDS; r++) {
        if ((uint32_t)r < rounds) {                   // wave-uniform
          const uint32_t p = (wave * rounds + r) * 64u + lane;
          const bool valid = p < n;
          const uint32_t d = (key[r] >> (12u + lo)) & pmask;
          uint64_t peers = __ballot(valid);
#pragma unroll
          for (int bb = 0; bb < 8; bb++) {
            const uint64_t bal = __ballot((d >> bb) & 1u);
            peers &= ((d >> bb) & 1u) ? bal : ~bal;
          }
          const uint32_t before = lanes_below(peers);
          const int leader = __ffsll((unsigned long long)peers) - 1;
          uint32_t old = 0;
          if (valid && lane == leader) {
            old = wcnt[wave][d];
            wcnt[wave][d] = old + __popcll(peers);
          }
          old = __shfl(old, leader < 0 ? 0 : leader);
          rank[r] = old + before;
          __builtin_amdgcn_wave_barrier();
        }
      }
      __syncthreads();
      {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < BK_WAVES; w++) {
          const uint32_t c = wcnt[w][threadIdx.x];
          wcnt[w][threadIdx.x] = run;
          run += c;
        }
        uint32_t tot;
        dstart[threadIdx.x] = block_exclusive_scan_256(run, wsum, tot);
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < BS_ROUNDS; r++) {
        if ((uint32_t)r < rounds) {
          const uint32_t p = (wave * rounds + r) * 64u + lane;
          if (p < n) {
            const uint32_t d = (key[r] >> (12u + lo)) & pmask;
            const uint32_t lp = dstart[d] + wcnt[wave][d] + rank[r];
            lkey[lp] = key[r];
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < BS_ROUNDS; r++) {
        if ((uint32_t)r < rounds) {
          const uint32_t p = (wave * rounds + r) * 64u + lane;
          if (p < n) key[r] = lkey[p];
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < BS_ROUNDS; r++) {
      if ((uint32_t)r < rounds) {
        const uint32_t p = (wave * rounds + r) * 64u + lane;
        uint32_t inst = 0;
        if (p < n) {
          // the one random access per Gaussian of the whole ordering: its emission record travels to its sorted position
          const uint32_t id = p1[start + (key[r] & 0xFFFu)].y;
          const uint4 rec = bins[DIRECT ? (key[r] & 0xFFFu) : id];
          uint32_t c = bin_count(rec);
          if (c == GM_BIN_COUNT_SAT) c = tiles[id];
          order0[obase + p] = id;
          bin_sorted[obase + p] = rec;
          inst = c;
        }
        chunk_add(chunk_inst, obase + (wave * rounds + r) * 64u, lane, inst);
      }
    }
  } else {
    // Slow path (more equal-depth Gaussians than the LDS holds): the workgroup sorts its own range through global memory,
    // ping-ponging between p1 and p0 restricted to [start, end): per pass a digit histogram over the
    // range, then 1024-entry tiles in order, each ranked stably as above.  Exact, sequential, rare.
    __shared__ uint32_t base[256];
    __shared__ uint32_t tcount[256];
    uint2* sp = p1; uint2* dp = p0;
    for (uint32_t pass = 0; pass < npass; pass++) {
      const uint32_t lo = pass * pb, pmask = (1u << min(pb, low_bits - lo)) - 1u;
      base[threadIdx.x] = 0;
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < n; i += BK_THREADS) atomicAdd(&base[((sp[start + i].x - ds.sub) >> lo) & pmask], 1u);
      __syncthreads();
      {
        uint32_t tot;
        const uint32_t v = base[threadIdx.x];
        const uint32_t e = block_exclusive_scan_256(v, wsum, tot);
        __syncthreads();
        base[threadIdx.x] = e;
      }
      __syncthreads();
      for (uint32_t t0 = 0; t0 < n; t0 += 1024u) {
        wcnt[0][threadIdx.x] = 0; wcnt[1][threadIdx.x] = 0; wcnt[2][threadIdx.x] = 0; wcnt[3][threadIdx.x] = 0;
        __syncthreads();
        uint32_t kk[4], vv[4], rk[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const uint32_t p = t0 + (wave * 4u + r) * 64u + lane;
          const bool valid = p < n;
          const uint2 kv = valid ? sp[start + p] : make_uint2(0u, 0u);
          kk[r] = kv.x; vv[r] = kv.y;
          const uint32_t d = ((kk[r] - ds.sub) >> lo) & pmask;
          uint64_t peers = __ballot(valid);
#pragma unroll
          for (int bb = 0; bb < 8; bb++) {
            const uint64_t bal = __ballot((d >> bb) & 1u);
            peers &= ((d >> bb) & 1u) ? bal : ~bal;
          }
          const uint32_t before = lanes_below(peers);
          const int leader = __ffsll((unsigned long long)peers) - 1;
          uint32_t old = 0;
          if (valid && lane == leader) {
            old = wcnt[wave][d];
            wcnt[wave][d] = old + __popcll(peers);
          }
          old = __shfl(old, leader < 0 ? 0 : leader);
          rk[r] = old + before;
          __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        {
          uint32_t run = 0;
#pragma unroll
          for (int w = 0; w < BK_WAVES; w++) {
            const uint32_t c = wcnt[w][threadIdx.x];
            wcnt[w][threadIdx.x] = run;
            run += c;
          }
          tcount[threadIdx.x] = run;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const uint32_t p = t0 + (wave * 4u + r) * 64u + lane;
          if (p < n) {
            const uint32_t d = ((kk[r] - ds.sub) >> lo) & pmask;
            const uint32_t dst = start + base[d] + wcnt[wave][d] + rk[r];
            dp[dst] = make_uint2(kk[r], vv[r]);
          }
        }
        __syncthreads();
        base[threadIdx.x] += tcount[threadIdx.x];
        __syncthreads();
      }
      __threadfence();                               // this workgroup's stores reach L2, its L1 forgets the old lines
      __syncthreads();
      uint2* t = sp; sp = dp; dp = t;
    }
    for (uint32_t i0 = 0; i0 < n; i0 += BK_THREADS) {
      const uint32_t i = i0 + threadIdx.x;
      uint32_t c = 0;
      if (i < n) {
        const uint32_t id = sp[start + i].y;
        const uint4 rec = bins[id];
        c = bin_count(rec);
        if (c == GM_BIN_COUNT_SAT) c = tiles[id];
        order0[start + i] = id;
        bin_sorted[start + i] = rec;
      }
      chunk_add(chunk_inst, start + i0 + wave * 64u, lane, c);
    }
  }
  if (trace && threadIdx.x == 0) { trace[3 * b] = t_begin; trace[3 * b + 1] = wall_clock64(); trace[3 * b + 2] = n; }
}

static unsigned long long* g_bucket_trace = nullptr;      // debugging aid (tools/bucket_stats.py, tools/pipeline_trace.py), never set by the package
extern "C" void gm_debug_bucket_trace(void* buffer) { g_bucket_trace = reinterpret_cast<unsigned long long*>(buffer); }
// bucket_sort_kernel's records are followed by those of the depth partition's scatter (from word 3 * 2048) and of the tile pass's
// scatter (from word 3 * 4096); the buffer holds 3 * (2048 + 2048 + 4096) words
static unsigned long long* scatter_trace(bool msd) { return g_bucket_trace ? g_bucket_trace + 3 * (msd ? 2048 : 4096) : nullptr; }

// ---------------------------------------------------------------------------------------------
// host side
int launch_depth_order(GeomState& g, int P, int debug, hipStream_t s, int* num_rendered_host, hipEvent_t count_event, const BatchOfs* bt) {
  constexpr int DB = GM_BUCKET_BITS, WAVES = 4, ROUNDS = GM_DP_ROUNDS, TILE = WAVES * ROUNDS * 64;
  const uint32_t nblk = ((uint32_t)P + TILE - 1) / TILE;
  const uint32_t nchunks = (nblk + BK_CHUNK - 1) / BK_CHUNK;
  const DigitSpec ds{0u, 0u, 0xFFFFFFFFu};
  const BatchOfs one = single_frame();
  const BatchOfs& B = bt ? *bt : one;
  const uint32_t nf = (uint32_t)B.frames;
  {
    StageScope sc(ST_DEPTH_SORT, s);
    hipLaunchKernelGGL((bk_hist_kernel<true, DB, WAVES, ROUNDS>), dim3(nblk + 1, 1, nf), dim3(WAVES * 64), 0, s, g.depth_key, (uint32_t)P, nullptr, ds, g.slots, g.hist,
                       g.acc, g.counters, g.coarse, g.dmap, g.bmap, B.geom, B.binning);
    GM_LAUNCH_CHECK(debug, s);
  }
  if (num_rendered_host) {      // the instance total is known here; the rest of the ordering overlaps the host's wait for it
    GM_HIP(hipMemcpyAsync(num_rendered_host, g.counters + GM_CNT_RENDERED, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (count_event) GM_HIP(hipEventRecord(count_event, s));
  }
  StageScope sc(ST_DEPTH_SORT, s);
  hipLaunchKernelGGL((bk_scan_kernel<true, DB>), dim3(nchunks, (1 << DB) / 256, nf), dim3(BK_THREADS), 0, s, g.hist, (uint32_t)TILE, (uint32_t)P, nullptr,
                     g.acc, g.bucket_start, g.counters, nullptr, 0u, B.geom, B.binning, B.image);
  GM_LAUNCH_CHECK(debug, s);
  hipLaunchKernelGGL((bk_scatter_kernel<true, DB, WAVES, ROUNDS>), dim3(nblk, 1, nf), dim3(WAVES * 64), 0, s, g.depth_key, g.dpairs[1], (uint32_t)P, nullptr, ds,
                     g.hist, nullptr, 0u, nullptr, 0, nullptr, g.dmap, g.counters, nullptr, nullptr, nf == 1 ? scatter_trace(true) : nullptr, nullptr,
                     B.geom, B.binning, B.image);
  GM_LAUNCH_CHECK(debug, s);
  hipLaunchKernelGGL(bucket_sort_kernel<false>, dim3(1 << DB, 1, nf), dim3(BK_THREADS), 0, s, g.counters, g.bmap, g.bucket_start, g.dpairs[1], g.dpairs[0], g.order,
                     g.tiles_touched, g.bin, g.bin_sorted, g.chunk_inst, nf == 1 ? g_bucket_trace : nullptr, 0u, nullptr, nullptr, nullptr, nullptr, B.geom);
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// C. Direct depth placement (gm_common.h, DepthSlab / GM_PLAN_*): frames of one view stream share a DepthPlan, a ring of
//    GM_PLAN_SLOTS depth tables.  Per frame:
//      arm_direct_kernel     zeroes the frame's accumulators and bucket counters, draws the frame's sequence number and copies the
//                            newest complete table of the plan into the frame's own dmap (a table in the plan may be replaced while
//                            this frame runs: frames of a stream are in flight on several HIP streams);
//      the fused preprocess  appends every visible Gaussian to its bucket's slab (gm_deform.hip, direct_place);
//      direct_plan_kernel    one workgroup: totals -> counters, bucket counts -> bucket_start, overflow / no table -> the frame is
//                            refused, and this frame's coarse histogram -> the stream's next table (slot = sequence % slots);
//      bucket_sort_kernel<true>.
//    Three launches and the random record gather less than A.  The writer protocol of a table: stamp = 0, table words, stamp =
//    sequence, newest = max(newest, sequence), all agent-scope with AGENT-scope release / acquire fences between (a workgroup-scope fence orders
//    nothing another XCD can observe: ADVICE round 4); a reader checks the stamp before and
//    after its copy.  A slot is rewritten GM_PLAN_SLOTS frames later - the copy of a frame that far behind would be torn, and is
//    caught by the stamp.
__global__ __launch_bounds__(256) void arm_direct_kernel(uint4* __restrict__ arm, uint32_t arm_vec, uint4* __restrict__ cnt, uint32_t cnt_vec,
                                                          uint32_t* __restrict__ hdr, uint32_t* __restrict__ plan, uint32_t* __restrict__ dmap,
                                                          uint32_t cap) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < arm_vec; i += gridDim.x * 256) arm[i] = z;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < cnt_vec; i += gridDim.x * 256) cnt[i] = z;
  if (blockIdx.x != 0) return;
  __shared__ uint32_t s_v;
  __shared__ uint32_t s_first[257];
  if (threadIdx.x == 0) {
    hdr[0] = atomicAdd(plan + 1, 1u) + 1u;
    s_v = ld_agent(plan);
  }
  __syncthreads();
  const uint32_t v = s_v, slot = v % GM_PLAN_SLOTS;
  const uint32_t* stamp = plan + 2 + slot;
  const uint32_t* table = plan + 2 + GM_PLAN_SLOTS + (size_t)slot * GM_COARSE_BINS;
  constexpr int PER = GM_COARSE_BINS / 256;
  bool ok = v != 0u && ld_agent(stamp) == v;
  uint32_t w[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) w[j] = ok ? ld_agent(table + threadIdx.x * PER + j) : 0u;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");            // (the words are in before the stamp is looked at again)
  ok = ok && ld_agent(stamp) == v;
  // Whatever the memory system did to the copy, it is USED only if it is a table: first buckets non-decreasing and equal to the
  // running sum of the bucket counts (base[c + 1] == base[c] + nb[c], base[0] == 0, at most 2048 buckets).  Any word sequence with
  // that property maps keys to buckets monotonically, which is all the order of the frame depends on.
  s_first[threadIdx.x] = w[0] >> 16;
  if (threadIdx.x == 0) s_first[256] = 0xFFFFFFFFu;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const uint32_t next = j + 1 < PER ? (w[j + 1 < PER ? j + 1 : j] >> 16) : s_first[threadIdx.x + 1];
    const uint32_t sum = (w[j] >> 16) + (w[j] & 0xFFFFu);
    ok = ok && (next == 0xFFFFFFFFu ? sum <= (1u << GM_BUCKET_BITS) : next == sum);
  }
  if (threadIdx.x == 0) ok = ok && (w[0] >> 16) == 0u;
  ok = __syncthreads_and(ok ? 1 : 0) != 0;
#pragma unroll
  for (int j = 0; j < PER; j++) dmap[threadIdx.x * PER + j] = ok ? w[j] : 0u;       // (no table: everything to bucket 0, refused below)
  if (threadIdx.x == 0) { hdr[1] = ok ? 1u : 0u; hdr[2] = cap; }
}

// one workgroup on the frame's critical path: the preprocess kernel's totals -> counters, the bucket counters -> bucket_start, and the
// verdict (no table / a bucket above its slab: refused)
__global__ __launch_bounds__(BK_THREADS) void direct_plan_kernel(const uint32_t* __restrict__ slots, uint32_t* __restrict__ counters,
                                                                  uint32_t* __restrict__ bucket_start, const uint32_t* __restrict__ hdr,
                                                                  const uint32_t* __restrict__ cnt, uint32_t cap) {
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_part[8];
  constexpr int PER = (1 << GM_BUCKET_BITS) / BK_THREADS;
  static_assert(GM_SLOTS == BK_THREADS, "one slot per thread");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint2 sl = *reinterpret_cast<const uint2*>(slots + GM_SLOT_STRIDE * threadIdx.x);      // {instances, visible}
  const uint32_t valid = hdr[1];
  uint32_t c[PER], sum = 0;
  bool over = false;
#pragma unroll
  for (int j = 0; j < PER; j++) c[j] = cnt[(size_t)(threadIdx.x * PER + j) * GM_SLAB_CNT_STRIDE];
#pragma unroll
  for (int j = 0; j < PER; j++) { over = over || c[j] > cap; sum += c[j]; }
  const uint32_t inst = wave_sum_u32(sl.x), vis = wave_sum_u32(sl.y);
  if (lane == 0) { s_part[wave] = inst; s_part[4 + wave] = vis; }
  uint32_t total;
  uint32_t excl = block_exclusive_scan_256(sum, s_w, total);         // (its barrier publishes s_part too)
#pragma unroll
  for (int j = 0; j < PER; j++) { bucket_start[threadIdx.x * PER + j] = excl; excl += c[j]; }
  const uint32_t visible = (s_part[4] + s_part[5]) + (s_part[6] + s_part[7]);
  const bool fail = __syncthreads_or(over ? 1 : 0) != 0 || valid == 0u || total != visible;
  if (threadIdx.x == 0) {
    bucket_start[1 << GM_BUCKET_BITS] = total;
    counters[GM_CNT_RENDERED] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    counters[GM_CNT_VISIBLE] = visible;
    counters[GM_CNT_NBUCKETS] = 1u << GM_BUCKET_BITS;
    if (fail) counters[GM_CNT_DIRECT_FAIL] = 1u;
  }
}

// partition path with a plan: the table bk_hist_kernel has just built goes to the stream's next frames
__global__ __launch_bounds__(256) void publish_plan_kernel(const uint32_t* __restrict__ dmap, const uint32_t* __restrict__ counters,
                                                            uint32_t* __restrict__ plan) {
  __shared__ uint32_t s_seq;
  if (counters[GM_CNT_VISIBLE] == 0u) return;                       // (nothing visible: the histogram launch wrote no table)
  if (threadIdx.x == 0) {
    s_seq = atomicAdd(plan + 1, 1u) + 1u;
    st_agent(plan + 2 + s_seq % GM_PLAN_SLOTS, 0u);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  const uint32_t seq = s_seq, slot = seq % GM_PLAN_SLOTS;
  uint32_t* table = plan + 2 + GM_PLAN_SLOTS + (size_t)slot * GM_COARSE_BINS;
  for (int c = threadIdx.x; c < GM_COARSE_BINS; c += 256) st_agent(table + c, dmap[c]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) {
    st_agent(plan + 2 + slot, seq);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    atomicMax(plan, seq);
  }
}

int launch_arm_direct(GeomState& g, DepthSlab& d, uint32_t* plan, hipStream_t s) {
  const uint32_t arm_vec = (uint32_t)(g.arm_words / 4), cnt_vec = (uint32_t)(((size_t)GM_SLAB_CNT_STRIDE << GM_BUCKET_BITS) / 4);
  hipLaunchKernelGGL(arm_direct_kernel, dim3(64), dim3(256), 0, s, reinterpret_cast<uint4*>(g.slots), arm_vec, reinterpret_cast<uint4*>(d.cnt), cnt_vec,
                     d.hdr, plan, g.dmap, d.cap);
  GM_HIP(hipGetLastError());
  return 0;
}

int launch_depth_order_direct(GeomState& g, DepthSlab& d, uint32_t* plan, int P, int debug, hipStream_t s, int* num_rendered_host, hipEvent_t count_event) {
  (void)P;
  StageScope sc(ST_DEPTH_SORT, s);
  hipLaunchKernelGGL(direct_plan_kernel, dim3(1), dim3(BK_THREADS), 0, s, g.slots, g.counters, g.bucket_start, d.hdr, d.cnt, d.cap);
  GM_LAUNCH_CHECK(debug, s);
  if (num_rendered_host) {
    GM_HIP(hipMemcpyAsync(num_rendered_host, g.counters + GM_CNT_RENDERED, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (count_event) GM_HIP(hipEventRecord(count_event, s));
  }
  hipLaunchKernelGGL(bucket_sort_kernel<true>, dim3((1 << GM_BUCKET_BITS) + 1), dim3(BK_THREADS), 0, s, g.counters, g.bmap, g.bucket_start, d.pairs, g.dpairs[0],
                     g.order, g.tiles_touched, d.recs, g.bin_sorted, g.chunk_inst, g_bucket_trace, d.cap, g.slots, g.coarse, d.hdr, plan, FrameOfs{});
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

int launch_publish_depth_plan(GeomState& g, uint32_t* plan, hipStream_t s) {
  hipLaunchKernelGGL(publish_plan_kernel, dim3(1), dim3(256), 0, s, g.dmap, g.counters, plan);
  GM_HIP(hipGetLastError());
  return 0;
}

// One stable pass over the pair stream b.pairs[from] -> b.pairs[from ^ 1].  b.acc must be zero on entry.
template <int DB, int WAVES>
static int tile_pass(BinningState& b, GeomState& g, int from, uint32_t n, const uint32_t* n_dev, DigitSpec ds, uint2* ranges, uint32_t nranges,
                     bool zero_acc_after, uint32_t* order_out, uint32_t* hint, uint32_t* epoch, uint32_t* order_scratch, int debug, hipStream_t s,
                     const BatchOfs& B) {
  constexpr int ROUNDS = GM_TP_ROUNDS;
  constexpr uint32_t TILE = WAVES * ROUNDS * 64;
  const uint32_t nblk = (n + TILE - 1) / TILE;
  const uint32_t nchunks = (nblk + BK_CHUNK - 1) / BK_CHUNK;
  const uint32_t nf = (uint32_t)B.frames;
  hipLaunchKernelGGL((bk_hist_kernel<false, DB, WAVES, ROUNDS>), dim3(nblk, 1, nf), dim3(WAVES * 64), 0, s, b.pairs[from], n, n_dev, ds, nullptr, b.hist, b.acc,
                     g.counters, nullptr, nullptr, nullptr, B.geom, B.binning);
  GM_LAUNCH_CHECK(debug, s);
  hipLaunchKernelGGL((bk_scan_kernel<false, DB>), dim3(nchunks ? nchunks : 1, (1 << DB) / 256, nf), dim3(BK_THREADS), 0, s, b.hist, TILE, n, n_dev,
                     b.acc, nullptr, g.counters, ranges, nranges, B.geom, B.binning, B.image);
  GM_LAUNCH_CHECK(debug, s);
  hipLaunchKernelGGL((bk_scatter_kernel<false, DB, WAVES, ROUNDS>), dim3(nblk + (order_out ? 1u : 0u), 1, nf), dim3(WAVES * 64), 0, s, b.pairs[from],
                     b.pairs[from ^ 1], n, n_dev, ds, b.hist, zero_acc_after ? b.acc : nullptr, (uint32_t)bk_acc_words(n), ranges,
                     (int)nranges, order_out, nullptr, nullptr, hint, epoch, (nf == 1 && nblk <= 4096u) ? scatter_trace(false) : nullptr, order_scratch,
                     B.geom, B.binning, B.image);
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

// Tile sort of the instance stream b.pairs[0] (n instances; n_dev != nullptr: the count is read on the device and n is the
// capacity).  tiles <= 2048: one 11-bit pass, result in pairs[1], ranges written by the scan.  Otherwise two 8-bit
// passes, result in pairs[0], ranges by tile_ranges_kernel (caller).  duplicate_kernel has zeroed b.acc.
int launch_tile_sort(GeomState& g, BinningState& b, ImageState& img, size_t n, const uint32_t* n_dev, int tiles, bool* order_done,
                     uint32_t* work_hint, int debug, hipStream_t s, const BatchOfs* bt) {
  StageScope sc(ST_TILE_SORT, s);
  *order_done = false;
  const BatchOfs one = single_frame();
  const BatchOfs& B = bt ? *bt : one;
  if (B.frames > 1 && (n == 0 || tiles > (1 << GM_BUCKET_BITS))) { set_error("batched tile sort: needs instances and at most 2048 list tiles"); return 1; }
  if (n == 0) {
    GM_HIP(hipMemsetAsync(img.ranges, 0, sizeof(uint2) * (size_t)tiles, s));
    return 0;
  }
  if (n > 0xFFFFF000ull) { set_error("tile sort: too many instances"); return 1; }
  // below ~0.5 M instances 16-wave workgroups would leave most of the chip idle: 4-wave ones
  if (tiles <= (1 << GM_BUCKET_BITS)) {
    *order_done = true;                 // the scatter launch carries the dispatch-order workgroup
    if (n <= (size_t(1) << 19))
      return tile_pass<GM_BUCKET_BITS, 4>(b, g, 0, (uint32_t)n, n_dev, DigitSpec{0u, 0u, (1u << GM_BUCKET_BITS) - 1u}, img.ranges, (uint32_t)tiles,
                                          false, img.tile_order, work_hint, img.epoch, img.tile_work, debug, s, B);
    return tile_pass<GM_BUCKET_BITS, GM_TILE_PASS_WAVES>(b, g, 0, (uint32_t)n, n_dev, DigitSpec{0u, 0u, (1u << GM_BUCKET_BITS) - 1u}, img.ranges, (uint32_t)tiles,
                                         false, img.tile_order, work_hint, img.epoch, img.tile_work, debug, s, B);
  }
  if (int rc = tile_pass<8, 4>(b, g, 0, (uint32_t)n, n_dev, DigitSpec{0u, 0u, 0xFFu}, nullptr, 0u, true, nullptr, nullptr, nullptr, nullptr, debug, s, B)) return rc;
  return tile_pass<8, 4>(b, g, 1, (uint32_t)n, n_dev, DigitSpec{0u, 8u, 0xFFu}, nullptr, 0u, false, nullptr, nullptr, nullptr, nullptr, debug, s, B);
}

}  // namespace gm
