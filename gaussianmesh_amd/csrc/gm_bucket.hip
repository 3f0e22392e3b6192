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
//    the preprocess kernel leaves min / max of the visible keys and the instance total in 64 atomic slots
//    (GeomState::slots; one slot per 64th of the workgroups, so no address sees more than a few hundred atomics);
//    1. MSD partition by bucket = (key - kmin) >> shift, shift chosen on the device so that the occupied range maps onto
//       <= 2048 buckets: bk_hist -> bk_scan -> bk_scatter (stable, visible keys only);
//    2. bucket_sort_kernel: one workgroup per bucket sorts its few hundred to few thousand (key, id) pairs by the
//       remaining low bits entirely in LDS (stable LSD passes of <= 8 bits on registers + one LDS copy), writes the final
//       order, the instance count of every Gaussian in that order and the bucket's instance total.  A bucket that does not
//       fit (a pile-up of equal depths) takes a slow, still exact, path through global memory.
//    Four launches and two passes over 8 B/Gaussian instead of twelve launches and four passes.
// B. Instances (emitted by duplicate_kernel in that order, keyed by list tile id | child mask << 16):
//    list tiles <= 2048 (1080p with 32-px parents: 2040; 4K with 64-px parents: 2040): ONE stable pass on an 11-bit digit,
//    bk_hist -> bk_scan -> bk_scatter; the scan's exclusive digit bases ARE the tile ranges, so identifyTileRanges and
//    its memset disappear.  More list tiles: two 8-bit passes + tile_ranges_kernel.
//
// One pass = three kernels; none depends on another workgroup of the same launch except bk_scan's epilogue, where the
// last workgroup to finish (agent-scope release -> relaxed counter -> agent-scope acquire, MI355X_MICROARCH.md
// "valid forms") turns the per-chunk totals into bases:
//   bk_hist     workgroup b counts the digits of its 4096 keys -> hist[b][digit] (one contiguous row)
//   bk_scan     workgroup (c, g): rows [32 c, 32 c + 32) x digits [256 g, 256 g + 256) -> exclusive prefixes in place +
//               chunk totals; last workgroup of digit group g: prefix over the chunks (in place), digit totals, prefix
//               over its 256 digits; last group: adds the group bases -> digit_base[0..ND], writes the tile ranges
//   bk_scatter  workgroup b: base[d] = digit_base[d] + chunk_base[b / 32][d] + hist[b][d]; ranks its keys stably
//               (match-any from DB ballots + per-wave digit counters in LDS), sorts the tile by digit in LDS, streams
//               it out as contiguous runs.
#include "gm_common.h"

namespace gm {

#define BK_THREADS 256
#define BK_WAVES 4
#define BK_ROUNDS 16
#define BK_CHUNK GM_BK_CHUNK
#define BS_CAP 4096                 // entries a bucket may have for the in-LDS sort (256 threads x 16)

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

// Every thread of the workgroup calls this.  Reduces the 64 slots {instance sum, max ~key, max key} the preprocess
// kernel filled: returns the bucket mapping (sub = kmin, shift), the number of buckets in use (0: nothing visible) and
// the instance total.  s_tmp: 4 shared words.
__device__ __forceinline__ uint32_t block_msd_params(const uint32_t* __restrict__ slots, uint32_t* s_tmp, DigitSpec& ds, uint32_t& total) {
  if (threadIdx.x < 64) {
    const uint4 s = reinterpret_cast<const uint4*>(slots)[threadIdx.x];
    const uint32_t sum = wave_sum_u32(s.x), nkmin = wave_max_u32(s.y), kmax = wave_max_u32(s.z);
    if (threadIdx.x == 0) { s_tmp[0] = sum; s_tmp[1] = ~nkmin; s_tmp[2] = kmax; }
  }
  __syncthreads();
  total = s_tmp[0];
  const uint32_t kmin = s_tmp[1], kmax = s_tmp[2];
  ds.sub = kmin; ds.mask = 0xFFFFFFFFu; ds.shift = 0;
  if (kmin > kmax) return 0u;                         // no visible key was recorded (slots still hold max(~key) = max(key) = 0)
  const uint32_t range = kmax - kmin;
  const int bits = range ? 32 - __clz((int)range) : 0;
  ds.shift = (uint32_t)max(0, bits - GM_BUCKET_BITS);
  return (range >> ds.shift) + 1u;                    // <= 2^GM_BUCKET_BITS
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

// ---------------------------------------------------------------------------------------------
template <bool MSD, int DB>
__global__ __launch_bounds__(BK_THREADS) void bk_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n_host,
                                                              const uint32_t* __restrict__ n_dev, DigitSpec ds,
                                                              const uint32_t* __restrict__ slots, uint32_t* __restrict__ hist,
                                                              uint32_t* __restrict__ counters) {
  constexpr int ND = 1 << DB;
  __shared__ uint32_t h[ND];
  __shared__ uint32_t s_tmp[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t n = n_host;
  if (n_dev) n = min(n, *n_dev);
  if (MSD) {
    uint32_t total;
    const uint32_t nb = block_msd_params(slots, s_tmp, ds, total);
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[GM_CNT_RENDERED] = total;   // num_rendered, for the host read-back
    if (nb == 0u) return;
  }
  const uint32_t nblk = (n + GM_BK_TILE - 1) / GM_BK_TILE;
  if (blockIdx.x >= nblk) return;
  for (int d = threadIdx.x; d < ND; d += BK_THREADS) h[d] = 0;
  __syncthreads();
  const uint32_t wbase = blockIdx.x * GM_BK_TILE + wave * (BK_ROUNDS * 64);
  uint32_t k[BK_ROUNDS];
#pragma unroll
  for (int r = 0; r < BK_ROUNDS; r++) {
    const uint32_t idx = wbase + r * 64 + lane;
    k[r] = idx < n ? keys[idx] : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int r = 0; r < BK_ROUNDS; r++) {
    const uint32_t idx = wbase + r * 64 + lane;
    const bool valid = idx < n && (!MSD || k[r] != 0xFFFFFFFFu);
    if (valid) atomicAdd(&h[((k[r] - ds.sub) >> ds.shift) & ds.mask], 1u);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < ND; d += BK_THREADS) hist[(size_t)blockIdx.x * ND + d] = h[d];
}

// ---------------------------------------------------------------------------------------------
// grid = (chunks, ND / 256).  ranges_out (optional): [nranges] uint2 {first, one past last} per digit value.
template <bool MSD, int DB>
__global__ __launch_bounds__(BK_THREADS) void bk_scan_kernel(uint32_t* __restrict__ hist, uint32_t n_host, const uint32_t* __restrict__ n_dev,
                                                              const uint32_t* __restrict__ slots, uint32_t* __restrict__ chunk_total,
                                                              uint32_t* __restrict__ digit_base, uint32_t* __restrict__ digit_total,
                                                              uint32_t* __restrict__ counters, uint2* __restrict__ ranges_out, uint32_t nranges) {
  constexpr int ND = 1 << DB, NG = ND / 256;
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t s_tmp[4];
  __shared__ uint32_t s_last;
  uint32_t n = n_host;
  if (n_dev) n = min(n, *n_dev);
  if (MSD) {
    DigitSpec ds; uint32_t total;
    if (block_msd_params(slots, s_tmp, ds, total) == 0u) n = 0;
  }
  const uint32_t nblk = (n + GM_BK_TILE - 1) / GM_BK_TILE;
  const uint32_t nchunks = (nblk + BK_CHUNK - 1) / BK_CHUNK;
  const uint32_t c = blockIdx.x, g = blockIdx.y;
  const uint32_t d = g * 256 + threadIdx.x;
  if (nblk == 0) {                                  // nothing to sort: every list is empty
    if (c == 0) {
      digit_base[d] = 0; digit_total[d] = 0;
      if (ranges_out && d < nranges) ranges_out[d] = make_uint2(0u, 0u);
      if (d == 0) digit_base[ND] = 0;
    }
    return;
  }
  if (c >= nchunks) return;
  {
    const uint32_t r0 = c * BK_CHUNK;
    const uint32_t nr = min((uint32_t)BK_CHUNK, nblk - r0);
    uint32_t v[BK_CHUNK];
#pragma unroll
    for (int r = 0; r < BK_CHUNK; r++) v[r] = (uint32_t)r < nr ? hist[(size_t)(r0 + r) * ND + d] : 0u;
    uint32_t run = 0;
#pragma unroll
    for (int r = 0; r < BK_CHUNK; r++) {
      if ((uint32_t)r < nr) hist[(size_t)(r0 + r) * ND + d] = run;
      run += v[r];
    }
    chunk_total[(size_t)c * ND + d] = run;
  }
  // ---- last workgroup of this digit group: chunk totals -> chunk bases, digit totals, prefix over the group's digits
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t old = __hip_atomic_fetch_add(&counters[GM_CNT_DONE + g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (old == nchunks - 1u) ? 1u : 0u;
    if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!s_last) return;
  uint32_t run = 0;
  for (uint32_t c2 = 0; c2 < nchunks; c2++) {
    const uint32_t t = __hip_atomic_load(&chunk_total[(size_t)c2 * ND + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    chunk_total[(size_t)c2 * ND + d] = run;
    run += t;
  }
  digit_total[d] = run;
  uint32_t gtotal;
  const uint32_t excl = block_exclusive_scan_256(run, wsum, gtotal);
  digit_base[d] = excl;                             // relative to the group until the epilogue below
  if (threadIdx.x == 0) counters[GM_CNT_GROUP + g] = gtotal;
  // ---- last digit group: add the group bases, publish the ranges, re-arm the counters
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t old = __hip_atomic_fetch_add(&counters[GM_CNT_DONE + 8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (old == (uint32_t)NG - 1u) ? 1u : 0u;
    if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  if (!s_last) return;
  const bool refused = !MSD && counters[GM_CNT_REFUSED] != 0u;     // emission refused: every list stays empty
  uint32_t gb = 0;
#pragma unroll
  for (int k = 0; k < NG; k++) {
    const uint32_t d2 = k * 256 + threadIdx.x;
    const uint32_t base = __hip_atomic_load(&digit_base[d2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + gb;
    const uint32_t tot = __hip_atomic_load(&digit_total[d2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    digit_base[d2] = base;
    if (ranges_out && d2 < nranges) ranges_out[d2] = (refused || tot == 0u) ? make_uint2(0u, 0u) : make_uint2(base, base + tot);   // empty lists stay {0, 0} as in the reference (zeroed ranges)
    gb += __hip_atomic_load(&counters[GM_CNT_GROUP + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) digit_base[ND] = gb;
  if (threadIdx.x <= 8) counters[GM_CNT_DONE + threadIdx.x] = 0u;
}

// ---------------------------------------------------------------------------------------------
template <bool MSD, bool IOTA, int DB>
__global__ __launch_bounds__(BK_THREADS) void bk_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                 uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                 uint32_t n_host, const uint32_t* __restrict__ n_dev, DigitSpec ds,
                                                                 const uint32_t* __restrict__ slots, const uint32_t* __restrict__ hist,
                                                                 const uint32_t* __restrict__ chunk_total,
                                                                 const uint32_t* __restrict__ digit_base) {
  constexpr int ND = 1 << DB, DPT = ND / 256;
  __shared__ uint16_t wcnt[BK_WAVES][ND];    // per-wave running digit counts (<= 1024) -> per-wave exclusive offsets (< 4096)
  __shared__ uint32_t gbase[ND];             // global base of each digit for this workgroup
  __shared__ uint32_t dstart[ND];            // start of each digit's run inside the locally sorted tile
  __shared__ uint32_t lkey[GM_BK_TILE];
  __shared__ uint32_t lval[GM_BK_TILE];
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t s_tmp[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t n = n_host;
  if (n_dev) n = min(n, *n_dev);
  if (MSD) {
    uint32_t total;
    if (block_msd_params(slots, s_tmp, ds, total) == 0u) return;
  }
  const uint32_t nblk = (n + GM_BK_TILE - 1) / GM_BK_TILE;
  if (blockIdx.x >= nblk) return;
  const uint32_t chunk = blockIdx.x / BK_CHUNK;
  for (int d = threadIdx.x; d < ND; d += BK_THREADS) {
#pragma unroll
    for (int w = 0; w < BK_WAVES; w++) wcnt[w][d] = 0;
    gbase[d] = digit_base[d] + chunk_total[(size_t)chunk * ND + d] + hist[(size_t)blockIdx.x * ND + d];
  }
  __syncthreads();

  const uint32_t wbase = blockIdx.x * GM_BK_TILE + wave * (BK_ROUNDS * 64);
  uint32_t key[BK_ROUNDS], rank[BK_ROUNDS];
#pragma unroll
  for (int r = 0; r < BK_ROUNDS; r++) {
    const uint32_t idx = wbase + r * 64 + lane;
    key[r] = idx < n ? keys_in[idx] : 0xFFFFFFFFu;
  }
  uint32_t vmask = 0;                          // bit r: this lane's key of round r takes part
#pragma unroll
  for (int r = 0; r < BK_ROUNDS; r++) {
    const uint32_t idx = wbase + r * 64 + lane;
    const bool valid = idx < n && (!MSD || key[r] != 0xFFFFFFFFu);
    vmask |= valid ? (1u << r) : 0u;
    const uint32_t d = ((key[r] - ds.sub) >> ds.shift) & ds.mask;
    uint64_t peers = __ballot(valid);          // match-any over the digit bits among the valid lanes
#pragma unroll
    for (int b = 0; b < DB; b++) {
      const uint64_t bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t before = __popcll(peers & lt_mask);
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
  {  // per digit: per-wave counts -> per-wave exclusive offsets; digit counts -> start of each digit's run in the tile
    uint32_t dc[DPT], sum = 0;
#pragma unroll
    for (int j = 0; j < DPT; j++) {
      const int d = threadIdx.x * DPT + j;
      uint32_t run = 0;
#pragma unroll
      for (int w = 0; w < BK_WAVES; w++) {
        const uint32_t c = wcnt[w][d];
        wcnt[w][d] = (uint16_t)run;
        run += c;
      }
      dc[j] = run; sum += run;
    }
    uint32_t excl = block_exclusive_scan_256(sum, wsum, tile_n);
#pragma unroll
    for (int j = 0; j < DPT; j++) { dstart[threadIdx.x * DPT + j] = excl; excl += dc[j]; }
  }
  __syncthreads();
  // local scatter into LDS: the tile becomes sorted by digit (stable), so the global stores below are contiguous runs
#pragma unroll
  for (int r = 0; r < BK_ROUNDS; r++) {
    if ((vmask >> r) & 1u) {
      const uint32_t idx = wbase + r * 64 + lane;
      const uint32_t d = ((key[r] - ds.sub) >> ds.shift) & ds.mask;
      const uint32_t lp = dstart[d] + wcnt[wave][d] + rank[r];
      lkey[lp] = key[r];
      lval[lp] = IOTA ? idx : vals_in[idx];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < GM_BK_TILE / BK_THREADS; i++) {
    const uint32_t lp = i * BK_THREADS + threadIdx.x;
    if (lp < tile_n) {
      const uint32_t k = lkey[lp];
      const uint32_t d = ((k - ds.sub) >> ds.shift) & ds.mask;
      const uint32_t dst = gbase[d] + (lp - dstart[d]);
      keys_out[dst] = k;
      vals_out[dst] = lval[lp];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// One workgroup per bucket of the MSD partition: (key, id) pairs [start, end) of k1 / v1 -> final order.
// Outputs: order0[start..end) = ids in (key, id) order, cnt_sorted[start..end) = tiles_touched of those ids,
// bucket_inst[b] = their sum.  k0 / order0 ranges [start, end) double as scratch on the slow path.
__global__ __launch_bounds__(BK_THREADS) void bucket_sort_kernel(const uint32_t* __restrict__ slots, const uint32_t* __restrict__ bucket_start,
                                                                  uint32_t* __restrict__ k1, uint32_t* __restrict__ v1,
                                                                  uint32_t* __restrict__ k0, uint32_t* __restrict__ order0,
                                                                  const uint32_t* __restrict__ tiles, uint32_t* __restrict__ cnt_sorted,
                                                                  uint32_t* __restrict__ bucket_inst) {
  __shared__ uint32_t wcnt[BK_WAVES][256];
  __shared__ uint32_t dstart[256];
  __shared__ uint32_t lkey[BS_CAP];
  __shared__ uint32_t lval[BS_CAP];
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t s_tmp[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  DigitSpec ds; uint32_t total;
  const uint32_t nb = block_msd_params(slots, s_tmp, ds, total);
  const uint32_t b = blockIdx.x;
  uint32_t start = 0, end = 0;
  if (b < nb) { start = bucket_start[b]; end = bucket_start[b + 1]; }
  const uint32_t n = end - start;
  if (n == 0u) { if (threadIdx.x == 0) bucket_inst[b] = 0u; return; }
  const uint32_t low_bits = ds.shift;                 // bits of (key - kmin) below the bucket index
  const uint32_t npass = (low_bits + 7u) / 8u;
  const uint32_t pb = npass ? (low_bits + npass - 1u) / npass : 0u;
  uint32_t inst = 0;
  if (n <= BS_CAP) {
    // wave w owns positions [w * rounds * 64, (w + 1) * rounds * 64) in rounds of 64: rank order == position order
    const uint32_t rounds = (n + 255u) / 256u;
    uint32_t key[BK_ROUNDS], val[BK_ROUNDS], rank[BK_ROUNDS];
#pragma unroll
    for (int r = 0; r < BK_ROUNDS; r++) {
      const uint32_t p = (wave * rounds + r) * 64u + lane;
      const bool valid = (uint32_t)r < rounds && p < n;
      key[r] = valid ? k1[start + p] : 0u;
      val[r] = valid ? v1[start + p] : 0u;
    }
    for (uint32_t pass = 0; pass < npass; pass++) {
      const uint32_t lo = pass * pb, pmask = (1u << min(pb, low_bits - lo)) - 1u;
      wcnt[0][threadIdx.x] = 0; wcnt[1][threadIdx.x] = 0; wcnt[2][threadIdx.x] = 0; wcnt[3][threadIdx.x] = 0;
      __syncthreads();
#pragma unroll
      for (int r = 0; r < BK_ROUNDS; r++) {
        if ((uint32_t)r < rounds) {                   // wave-uniform
          const uint32_t p = (wave * rounds + r) * 64u + lane;
          const bool valid = p < n;
          const uint32_t d = ((key[r] - ds.sub) >> lo) & pmask;
          uint64_t peers = __ballot(valid);
#pragma unroll
          for (int bb = 0; bb < 8; bb++) {
            const uint64_t bal = __ballot((d >> bb) & 1u);
            peers &= ((d >> bb) & 1u) ? bal : ~bal;
          }
          const uint32_t before = __popcll(peers & lt_mask);
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
      for (int r = 0; r < BK_ROUNDS; r++) {
        if ((uint32_t)r < rounds) {
          const uint32_t p = (wave * rounds + r) * 64u + lane;
          if (p < n) {
            const uint32_t d = ((key[r] - ds.sub) >> lo) & pmask;
            const uint32_t lp = dstart[d] + wcnt[wave][d] + rank[r];
            lkey[lp] = key[r];
            lval[lp] = val[r];
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < BK_ROUNDS; r++) {
        if ((uint32_t)r < rounds) {
          const uint32_t p = (wave * rounds + r) * 64u + lane;
          if (p < n) { key[r] = lkey[p]; val[r] = lval[p]; }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < BK_ROUNDS; r++) {
      if ((uint32_t)r < rounds) {
        const uint32_t p = (wave * rounds + r) * 64u + lane;
        if (p < n) {
          const uint32_t c = tiles[val[r]];
          order0[start + p] = val[r];
          cnt_sorted[start + p] = c;
          inst += c;
        }
      }
    }
  } else {
    // Slow path (more equal-depth Gaussians than the LDS holds): the workgroup sorts its own range through global memory,
    // ping-ponging between (k1, v1) and (k0, order0) restricted to [start, end): per pass a digit histogram over the
    // range, then 1024-entry tiles in order, each ranked stably as above.  Exact, sequential, rare.
    __shared__ uint32_t base[256];
    __shared__ uint32_t tcount[256];
    uint32_t* sk = k1; uint32_t* sv = v1; uint32_t* dk = k0; uint32_t* dv = order0;
    for (uint32_t pass = 0; pass < npass; pass++) {
      const uint32_t lo = pass * pb, pmask = (1u << min(pb, low_bits - lo)) - 1u;
      base[threadIdx.x] = 0;
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < n; i += BK_THREADS) atomicAdd(&base[((sk[start + i] - ds.sub) >> lo) & pmask], 1u);
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
          kk[r] = valid ? sk[start + p] : 0u;
          vv[r] = valid ? sv[start + p] : 0u;
          const uint32_t d = ((kk[r] - ds.sub) >> lo) & pmask;
          uint64_t peers = __ballot(valid);
#pragma unroll
          for (int bb = 0; bb < 8; bb++) {
            const uint64_t bal = __ballot((d >> bb) & 1u);
            peers &= ((d >> bb) & 1u) ? bal : ~bal;
          }
          const uint32_t before = __popcll(peers & lt_mask);
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
            dk[dst] = kk[r];
            dv[dst] = vv[r];
          }
        }
        __syncthreads();
        base[threadIdx.x] += tcount[threadIdx.x];
        __syncthreads();
      }
      __threadfence();                               // this workgroup's stores reach L2, its L1 forgets the old lines
      __syncthreads();
      uint32_t* t;
      t = sk; sk = dk; dk = t;
      t = sv; sv = dv; dv = t;
    }
    for (uint32_t i = threadIdx.x; i < n; i += BK_THREADS) {
      const uint32_t id = sv[start + i];
      const uint32_t c = tiles[id];
      if (sv != order0) order0[start + i] = id;
      cnt_sorted[start + i] = c;
      inst += c;
    }
  }
  inst = wave_sum_u32(inst);
  __syncthreads();
  if (lane == 0) wsum[wave] = inst;
  __syncthreads();
  if (threadIdx.x == 0) bucket_inst[b] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ---------------------------------------------------------------------------------------------
// host side
int launch_depth_order(GeomState& g, int P, int debug, hipStream_t s, int* num_rendered_host, hipEvent_t count_event) {
  constexpr int DB = GM_BUCKET_BITS;
  const uint32_t nblk = ((uint32_t)P + GM_BK_TILE - 1) / GM_BK_TILE;
  const uint32_t nchunks = (nblk + BK_CHUNK - 1) / BK_CHUNK;
  const DigitSpec ds{0u, 0u, 0xFFFFFFFFu};
  {
    StageScope sc(ST_DEPTH_SORT, s);
    hipLaunchKernelGGL((bk_hist_kernel<true, DB>), dim3(nblk), dim3(BK_THREADS), 0, s, g.depth_key[0], (uint32_t)P, nullptr, ds, g.slots, g.hist,
                       g.counters);
    GM_LAUNCH_CHECK(debug, s);
  }
  if (num_rendered_host) {      // the instance total is known here; the rest of the ordering overlaps the host's wait for it
    GM_HIP(hipMemcpyAsync(num_rendered_host, g.counters + GM_CNT_RENDERED, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (count_event) GM_HIP(hipEventRecord(count_event, s));
  }
  StageScope sc(ST_DEPTH_SORT, s);
  hipLaunchKernelGGL((bk_scan_kernel<true, DB>), dim3(nchunks, (1 << DB) / 256), dim3(BK_THREADS), 0, s, g.hist, (uint32_t)P, nullptr, g.slots,
                     g.chunk_total, g.bucket_start, g.digit_total, g.counters, nullptr, 0u);
  GM_LAUNCH_CHECK(debug, s);
  hipLaunchKernelGGL((bk_scatter_kernel<true, true, DB>), dim3(nblk), dim3(BK_THREADS), 0, s, g.depth_key[0], nullptr, g.depth_key[1],
                     g.order[1], (uint32_t)P, nullptr, ds, g.slots, g.hist, g.chunk_total, g.bucket_start);
  GM_LAUNCH_CHECK(debug, s);
  hipLaunchKernelGGL(bucket_sort_kernel, dim3(1 << DB), dim3(BK_THREADS), 0, s, g.slots, g.bucket_start, g.depth_key[1], g.order[1],
                     g.depth_key[0], g.order[0], g.tiles_touched, g.cnt_sorted, g.bucket_inst);
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

// tile sort of the instance stream keys[0] / vals[0] (n instances; n_dev != nullptr: the count is read on the device and
// n is the capacity).  tiles <= 2048: one 11-bit pass, result in slot 1, ranges written by the scan.  Otherwise two
// 8-bit passes, result in slot 0, ranges by tile_ranges_kernel (caller).
template <int DB>
static int tile_pass(BinningState& b, GeomState& g, int from, uint32_t n, const uint32_t* n_dev, DigitSpec ds, uint2* ranges, uint32_t nranges,
                     int debug, hipStream_t s) {
  const uint32_t nblk = (n + GM_BK_TILE - 1) / GM_BK_TILE;
  const uint32_t nchunks = (nblk + BK_CHUNK - 1) / BK_CHUNK;
  hipLaunchKernelGGL((bk_hist_kernel<false, DB>), dim3(nblk), dim3(BK_THREADS), 0, s, b.keys[from], n, n_dev, ds, nullptr, b.hist, g.counters);
  GM_LAUNCH_CHECK(debug, s);
  hipLaunchKernelGGL((bk_scan_kernel<false, DB>), dim3(nchunks ? nchunks : 1, (1 << DB) / 256), dim3(BK_THREADS), 0, s, b.hist, n, n_dev, nullptr,
                     b.chunk_total, b.digit_base, b.digit_total, g.counters, ranges, nranges);
  GM_LAUNCH_CHECK(debug, s);
  hipLaunchKernelGGL((bk_scatter_kernel<false, false, DB>), dim3(nblk), dim3(BK_THREADS), 0, s, b.keys[from], b.vals[from], b.keys[from ^ 1],
                     b.vals[from ^ 1], n, n_dev, ds, nullptr, b.hist, b.chunk_total, b.digit_base);
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

int launch_tile_sort(GeomState& g, BinningState& b, ImageState& img, size_t n, const uint32_t* n_dev, int tiles, int debug, hipStream_t s) {
  StageScope sc(ST_TILE_SORT, s);
  if (n == 0) {
    GM_HIP(hipMemsetAsync(img.ranges, 0, sizeof(uint2) * (size_t)tiles, s));
    return 0;
  }
  if (n > 0xFFFFF000ull) { set_error("tile sort: too many instances"); return 1; }
  if (tiles <= (1 << GM_BUCKET_BITS))
    return tile_pass<GM_BUCKET_BITS>(b, g, 0, (uint32_t)n, n_dev, DigitSpec{0u, 0u, (1u << GM_BUCKET_BITS) - 1u}, img.ranges, (uint32_t)tiles, debug, s);
  if (int rc = tile_pass<8>(b, g, 0, (uint32_t)n, n_dev, DigitSpec{0u, 0u, 0xFFu}, nullptr, 0u, debug, s)) return rc;
  return tile_pass<8>(b, g, 1, (uint32_t)n, n_dev, DigitSpec{0u, 8u, 0xFFu}, nullptr, 0u, debug, s);
}

}  // namespace gm
