// gm_sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs, hand-written for wave64 / gfx950.
//
// Replaces the cub::DeviceRadixSort::SortPairs calls of the reference
// (cuda_rasterizer/rasterizer_impl.cu:478-483 for the (tile|depth) instance sort,
//  scene/simple_knn/cuda_headers/simple_knn.cu:210-213 for the Morton sort).
//
// Design (DESIGN.md "ordering"): the reference sorts R instances by a 64-bit (tile<<32 | depth) key in
// ceil((32+bit)/8) passes over R x 12 B.  Here the same total order is produced by (1) sorting the P
// Gaussians by the 32 depth bits (4 passes over P x 8 B) and (2) a stable sort of the instances, emitted
// in that depth order, by tile id only (2 passes over R x 8 B): ~4x less HBM traffic for the same list.
// Both steps use this file.
//
// One pass = three kernels (no inter-workgroup dependency inside a launch, so no reliance on
// cross-XCD L2 coherence):
//   K1 radix_hist     : workgroup b counts the 256 digit values of its keys -> hist[b][digit] (one contiguous
//                       1 KiB row per workgroup; the transposed [digit][b] layout of the first version turned every
//                       4-byte counter into its own partial cache line: WRITE_SIZE 8 MB for a 1 MB table, and the
//                       scatter's strided reads of it doubled that kernel's traffic)
//   K2 radix_scan     : workgroup c turns rows [32 c, 32 c + 32) into exclusive prefixes per digit (thread =
//                       digit, the 32 row loads are independent and coalesced) and writes the chunk totals
//                       chunk_total[c][digit]
//   K3 radix_scatter  : workgroup b sums the chunk totals below / over all chunks (digit totals), re-reads its keys,
//                       ranks them stably (per-wave match-any with 64-bit ballots + per-wave digit counters in LDS),
//                       sorts the tile by digit inside LDS, and streams it out: digit d's run goes to
//                       exclusive_scan(digit totals)[d] + chunk offset[d] + hist[b][d] + (position in run), so
//                       global stores are contiguous runs rather than per-lane scatters.
// A workgroup is 256 threads = 4 waves; wave w owns the contiguous 1024-key slice w of the tile and
// walks it in 16 rounds of 64 consecutive keys (lane l <-> key round*64+l), so loads are fully
// coalesced and rank order == index order (stability).
#include "gm_common.h"

namespace gm {

#define RS_THREADS 256
#define RS_WAVES 4
// ROUNDS = rounds of 64 keys per wave; a workgroup sorts 4 * ROUNDS * 64 keys.  ROUNDS = 16 (4096 keys) for the
// instance sort; ROUNDS = 4 (1024 keys) for the 1 M-key depth / Morton sorts, which would otherwise run on fewer
// workgroups than the chip has CUs.

// Digit histogram of one tile.  Keys with equal digits inside a wave are counted once (match-any with eight 64-bit
// ballots, the peer-group leader adds popcount) instead of one LDS atomic per key: the tile-id digits of the instance
// sort are so coherent that per-key atomics serialise on a handful of LDS addresses.
template <int ROUNDS>
__global__ __launch_bounds__(RS_THREADS) void radix_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                                 int shift, uint32_t mask, uint32_t* __restrict__ hist,
                                                                 uint32_t nblk) {
  __shared__ uint32_t h[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t wbase = blockIdx.x * (RS_WAVES * ROUNDS * 64) + wave * (ROUNDS * 64);
#pragma unroll
  for (int r = 0; r < ROUNDS; r++) {
    const uint32_t idx = wbase + r * 64 + lane;
    const bool valid = idx < n;
    const uint32_t d = valid ? (keys[idx] >> shift) & mask : 0u;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const uint64_t bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    if (valid && lane == __ffsll((unsigned long long)peers) - 1) atomicAdd(&h[d], (uint32_t)__popcll(peers));
  }
  __syncthreads();
  hist[(size_t)blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}

// workgroup c: exclusive prefix per digit over the histogram rows of chunk c (in place), chunk totals -> chunk_total[c][.]
__global__ __launch_bounds__(RS_THREADS) void radix_scan_kernel(uint32_t* __restrict__ hist, uint32_t nblk,
                                                                 uint32_t* __restrict__ chunk_total) {
  const uint32_t r0 = blockIdx.x * GM_SORT_CHUNK;
  const uint32_t nr = min((uint32_t)GM_SORT_CHUNK, nblk - r0);
  uint32_t v[GM_SORT_CHUNK];
#pragma unroll
  for (int r = 0; r < GM_SORT_CHUNK; r++) v[r] = (uint32_t)r < nr ? hist[(size_t)(r0 + r) * 256 + threadIdx.x] : 0u;
  uint32_t run = 0;
#pragma unroll
  for (int r = 0; r < GM_SORT_CHUNK; r++) {
    if ((uint32_t)r < nr) hist[(size_t)(r0 + r) * 256 + threadIdx.x] = run;
    run += v[r];
  }
  chunk_total[(size_t)blockIdx.x * 256 + threadIdx.x] = run;
}

template <bool IOTA, int RS_ROUNDS>
__global__ __launch_bounds__(RS_THREADS) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                    const uint32_t* __restrict__ vals_in,
                                                                    uint32_t* __restrict__ keys_out,
                                                                    uint32_t* __restrict__ vals_out, uint32_t n, int shift,
                                                                    uint32_t mask, const uint32_t* __restrict__ hist,
                                                                    uint32_t nblk, const uint32_t* __restrict__ digit_total) {
  __shared__ uint32_t wcnt[RS_WAVES][256];   // per-wave running digit counts -> per-wave exclusive offsets
  __shared__ uint32_t gbase[256];            // global base of each digit for this workgroup
  __shared__ uint32_t dstart[256];           // start of each digit's run inside the locally sorted tile
  __shared__ uint32_t wsum2[RS_WAVES];
  constexpr int TILE_KEYS = RS_WAVES * RS_ROUNDS * 64;
  __shared__ uint32_t lkey[TILE_KEYS];       // locally sorted tile (32 KiB with lval at 4096 keys)
  __shared__ uint32_t lval[TILE_KEYS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < RS_WAVES; w++) wcnt[w][threadIdx.x] = 0;
  {  // digit totals and the offset of this workgroup's chunk from the chunk totals, then the exclusive scan over digits
    const uint32_t nchunks = (nblk + GM_SORT_CHUNK - 1) / GM_SORT_CHUNK, mychunk = blockIdx.x / GM_SORT_CHUNK;
    uint32_t v = 0, below = 0;
    for (uint32_t c = 0; c < nchunks; c++) {
      const uint32_t t = digit_total[(size_t)c * 256 + threadIdx.x];
      v += t;
      below += c < mychunk ? t : 0u;
    }
    __shared__ uint32_t wsum[RS_WAVES];
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) woff += (w < wave) ? wsum[w] : 0;
    gbase[threadIdx.x] = woff + incl - v + below + hist[(size_t)blockIdx.x * 256 + threadIdx.x];
  }
  __syncthreads();

  const uint32_t wbase = blockIdx.x * TILE_KEYS + wave * (RS_ROUNDS * 64);
  uint32_t key[RS_ROUNDS], rank[RS_ROUNDS];
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; r++) {
    const uint32_t idx = wbase + r * 64 + lane;
    key[r] = idx < n ? keys_in[idx] : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; r++) {
    const uint32_t idx = wbase + r * 64 + lane;
    const bool valid = idx < n;
    const uint32_t d = (key[r] >> shift) & mask;
    // match-any over the 8 digit bits among valid lanes
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const uint64_t bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
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
    __builtin_amdgcn_wave_barrier();         // keep the per-wave LDS counter updates of successive rounds in order
  }
  __syncthreads();
  uint32_t dcount;
  {  // turn per-wave counts into per-wave exclusive offsets (thread d handles digit d)
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) {
      const uint32_t c = wcnt[w][threadIdx.x];
      wcnt[w][threadIdx.x] = run;
      run += c;
    }
    dcount = run;                               // keys of this workgroup with digit == threadIdx.x
  }
  {  // dstart[d] = exclusive scan of dcount over digits: position of digit d's run inside the sorted tile
    uint32_t incl = dcount;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wsum2[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) woff += (w < wave) ? wsum2[w] : 0;
    dstart[threadIdx.x] = woff + incl - dcount;
  }
  __syncthreads();
  // local scatter into LDS: the tile becomes sorted by digit (stable), so the global stores below are
  // contiguous runs (one run per digit) instead of 64 unrelated 4-byte writes per wave instruction
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; r++) {
    const uint32_t idx = wbase + r * 64 + lane;
    if (idx < n) {
      const uint32_t d = (key[r] >> shift) & mask;
      const uint32_t lp = dstart[d] + wcnt[wave][d] + rank[r];
      lkey[lp] = key[r];
      lval[lp] = IOTA ? idx : vals_in[idx];
    }
  }
  __syncthreads();
  const uint32_t tile_n = min((uint32_t)TILE_KEYS, n - blockIdx.x * TILE_KEYS);
#pragma unroll
  for (int i = 0; i < TILE_KEYS / RS_THREADS; i++) {
    const uint32_t lp = i * RS_THREADS + threadIdx.x;
    if (lp < tile_n) {
      const uint32_t k = lkey[lp];
      const uint32_t d = (k >> shift) & mask;
      const uint32_t dst = gbase[d] + (lp - dstart[d]);
      keys_out[dst] = k;
      vals_out[dst] = lval[lp];
    }
  }
}

template <int ROUNDS>
static int radix_sort_pairs_t(uint32_t* keys[2], uint32_t* vals[2], uint32_t* hist, uint32_t* digit_total, size_t n,
                              int bits, bool iota_values, int debug, hipStream_t s) {
  const uint32_t tile_keys = RS_WAVES * ROUNDS * 64;
  const uint32_t nblk = (uint32_t)((n + tile_keys - 1) / tile_keys);
  int cur = 0;
  for (int shift = 0; shift < bits; shift += 8) {
    const int nb = (bits - shift) < 8 ? (bits - shift) : 8;
    const uint32_t mask = (1u << nb) - 1u;
    hipLaunchKernelGGL(radix_hist_kernel<ROUNDS>, dim3(nblk), dim3(RS_THREADS), 0, s, keys[cur], (uint32_t)n, shift, mask, hist, nblk);
    GM_LAUNCH_CHECK(debug, s);
    hipLaunchKernelGGL(radix_scan_kernel, dim3((nblk + GM_SORT_CHUNK - 1) / GM_SORT_CHUNK), dim3(RS_THREADS), 0, s, hist, nblk, digit_total);
    GM_LAUNCH_CHECK(debug, s);
    if (iota_values && shift == 0)
      hipLaunchKernelGGL((radix_scatter_kernel<true, ROUNDS>), dim3(nblk), dim3(RS_THREADS), 0, s, keys[cur], vals[cur], keys[cur ^ 1],
                         vals[cur ^ 1], (uint32_t)n, shift, mask, hist, nblk, digit_total);
    else
      hipLaunchKernelGGL((radix_scatter_kernel<false, ROUNDS>), dim3(nblk), dim3(RS_THREADS), 0, s, keys[cur], vals[cur], keys[cur ^ 1],
                         vals[cur ^ 1], (uint32_t)n, shift, mask, hist, nblk, digit_total);
    GM_LAUNCH_CHECK(debug, s);
    cur ^= 1;
  }
  return 0;
}

// hist must hold 256 * sort_blocks(n) counters, digit_total sort_chunk_counters(n)
int radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], uint32_t* hist, uint32_t* digit_total, size_t n,
                     int bits, bool iota_values, int debug, hipStream_t s) {
  if (n == 0) return 0;
  if (n > 0xFFFFFFF0ull) { set_error("radix_sort_pairs: n too large"); return 1; }
  if (n <= GM_SORT_SMALL_N) return radix_sort_pairs_t<4>(keys, vals, hist, digit_total, n, bits, iota_values, debug, s);
  if (n <= GM_SORT_MID_N) return radix_sort_pairs_t<8>(keys, vals, hist, digit_total, n, bits, iota_values, debug, s);
  return radix_sort_pairs_t<16>(keys, vals, hist, digit_total, n, bits, iota_values, debug, s);
}

}  // namespace gm
