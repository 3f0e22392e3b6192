// gm_binning.hip -- instance counting, instance emission and tile ranges.
//
// Replaces (reference, RAST = gaussian_renderer/diff_gaussian_rasterizater/cuda_rasterizer):
//   RAST/rasterizer_impl.cu:407   cub::DeviceScan::InclusiveSum over tiles_touched
//   RAST/rasterizer_impl.cu:70-111 duplicateWithKeys
//   RAST/rasterizer_impl.cu:116-138 identifyTileRanges (+ the cudaMemset of ranges, :485)
//
// Gaussians are visited in (depth, id) order (GeomState::order[0], produced by the depth sort), so the
// emitted instance stream is already depth-ordered and only needs a stable sort by tile id afterwards.
// Workgroup b owns sorted positions [b*2048, (b+1)*2048); thread t owns 8 consecutive positions.
#include "gm_common.h"
#pragma clang fp contract(off)

namespace gm {

#define BN_THREADS 256
#define BN_PER_THREAD (GM_SCAN_ITEMS / BN_THREADS)   // 8

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wsum /*[4] shared*/, uint32_t& total) {
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
  for (int w = 0; w < BN_THREADS / 64; w++) {
    const uint32_t s = wsum[w];
    woff += (w < wave) ? s : 0;
    tot += s;
  }
  total = tot;
  return woff + incl - v;
}

__global__ __launch_bounds__(BN_THREADS) void tile_block_sums_kernel(const uint32_t* __restrict__ order,
                                                                      const uint32_t* __restrict__ tiles, int P,
                                                                      uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t wsum[BN_THREADS / 64];
  const int base = blockIdx.x * GM_SCAN_ITEMS + threadIdx.x * BN_PER_THREAD;
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < BN_PER_THREAD; i++) {
    const int s = base + i;
    if (s < P) sum += tiles[order[s]];
  }
  uint32_t total;
  block_exclusive_scan(sum, wsum, total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single workgroup: exclusive scan of block_sums[nb] in place, grand total -> counters[0]
__global__ __launch_bounds__(BN_THREADS) void scan_block_sums_kernel(uint32_t* __restrict__ block_sums, int nb,
                                                                      uint32_t* __restrict__ counters) {
  __shared__ uint32_t wsum[BN_THREADS / 64];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += BN_THREADS) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < nb ? block_sums[i] : 0;
    uint32_t total;
    const uint32_t excl = block_exclusive_scan(v, wsum, total);
    const uint32_t carry = carry_s;
    if (i < nb) block_sums[i] = carry + excl;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) counters[0] = carry_s;
}

int launch_tile_count_scan(GeomState& g, int P, int debug, hipStream_t s) {
  StageScope sc(ST_SCAN, s);
  const int nb = (P + GM_SCAN_ITEMS - 1) / GM_SCAN_ITEMS;
  if (nb > 0) {
    hipLaunchKernelGGL(tile_block_sums_kernel, dim3(nb), dim3(BN_THREADS), 0, s, g.order[0], g.tiles_touched, P, g.block_sums);
    GM_LAUNCH_CHECK(debug, s);
  }
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(BN_THREADS), 0, s, g.block_sums, nb, g.counters);
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

__device__ __forceinline__ void get_rect(float px, float py, int r, int gx, int gy, int& x0, int& y0, int& x1, int& y1) {
  x0 = min(gx, max(0, (int)((px - r) / GM_TILE)));
  y0 = min(gy, max(0, (int)((py - r) / GM_TILE)));
  x1 = min(gx, max(0, (int)((px + r + GM_TILE - 1) / GM_TILE)));
  y1 = min(gy, max(0, (int)((py + r + GM_TILE - 1) / GM_TILE)));
}

__global__ __launch_bounds__(BN_THREADS) void duplicate_kernel(const uint32_t* __restrict__ order,
                                                                const uint32_t* __restrict__ tiles,
                                                                const float4* __restrict__ splat,
                                                                const int* __restrict__ radii, int P, int gx, int gy,
                                                                const uint32_t* __restrict__ block_sums,
                                                                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
  __shared__ uint32_t wsum[BN_THREADS / 64];
  const int base = blockIdx.x * GM_SCAN_ITEMS + threadIdx.x * BN_PER_THREAD;
  uint32_t gid[BN_PER_THREAD], cnt[BN_PER_THREAD], sum = 0;
#pragma unroll
  for (int i = 0; i < BN_PER_THREAD; i++) {
    const int s = base + i;
    gid[i] = s < P ? order[s] : 0u;
    cnt[i] = s < P ? tiles[gid[i]] : 0u;
    sum += cnt[i];
  }
  uint32_t total;
  uint32_t off = block_sums[blockIdx.x] + block_exclusive_scan(sum, wsum, total);
#pragma unroll
  for (int i = 0; i < BN_PER_THREAD; i++) {
    if (cnt[i] == 0) continue;
    const uint32_t g = gid[i];
    const float4 s0 = splat[3 * (size_t)g];
    int x0, y0, x1, y1;
    get_rect(s0.x, s0.y, radii[g], gx, gy, x0, y0, x1, y1);
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) {
        keys_out[off] = (uint32_t)(y * gx + x);
        vals_out[off] = g;
        off++;
      }
  }
}

int launch_duplicate(GeomState& g, BinningState& b, int P, int W, int H, const int* radii, int debug, hipStream_t s) {
  StageScope sc(ST_DUPLICATE, s);
  const int nb = (P + GM_SCAN_ITEMS - 1) / GM_SCAN_ITEMS;
  const int gx = (W + GM_TILE - 1) / GM_TILE, gy = (H + GM_TILE - 1) / GM_TILE;
  (void)radii;
  if (nb > 0)
    hipLaunchKernelGGL(duplicate_kernel, dim3(nb), dim3(BN_THREADS), 0, s, g.order[0], g.tiles_touched, g.splat, g.radii, P,
                       gx, gy, g.block_sums, b.keys[0], b.vals[0]);
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint32_t* __restrict__ keys, int R, uint2* __restrict__ ranges) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R) return;
  const uint32_t cur = keys[i];
  if (i == 0) ranges[cur].x = 0;
  else {
    const uint32_t prev = keys[i - 1];
    if (cur != prev) { ranges[prev].y = (uint32_t)i; ranges[cur].x = (uint32_t)i; }
  }
  if (i == R - 1) ranges[cur].y = (uint32_t)R;
}

// Workgroup schedule for the blend kernels: tiles ordered by list length, longest first (LPT), so the
// multi-thousand-entry tiles start at once and the short ones fill in behind them instead of the reverse.
// Counting sort on a quarter-octave length class (64 classes); single workgroup, tiles <= a few 10k.
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint2* __restrict__ ranges, int tiles, uint32_t* __restrict__ order) {
  __shared__ uint32_t cnt[64];
  if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
  __syncthreads();
  auto cls = [](uint32_t n) -> uint32_t {
    if (n == 0) return 63u;
    const uint32_t lg = 31u - (uint32_t)__clz((int)n);           // floor(log2 n)
    const uint32_t frac = lg >= 2 ? (n >> (lg - 2)) & 3u : 0u;     // next two bits
    const uint32_t c = 4u * lg + frac;                             // 0..127 in principle, n < 2^15.75 in practice
    return c >= 62u ? 0u : 62u - c;                                // long lists -> small class
  };
  for (int t = threadIdx.x; t < tiles; t += 1024) atomicAdd(&cnt[cls(ranges[t].y - ranges[t].x)], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int c = 0; c < 64; c++) { const uint32_t v = cnt[c]; cnt[c] = run; run += v; }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < tiles; t += 1024) {
    const uint32_t pos = atomicAdd(&cnt[cls(ranges[t].y - ranges[t].x)], 1u);
    order[pos] = (uint32_t)t;
  }
}

int launch_tile_ranges(BinningState& b, int slot, ImageState& img, int R, int tiles, int debug, hipStream_t s) {
  StageScope sc(ST_RANGES, s);
  GM_HIP(hipMemsetAsync(img.ranges, 0, sizeof(uint2) * (size_t)tiles, s));
  if (R > 0) hipLaunchKernelGGL(tile_ranges_kernel, dim3((R + 255) / 256), dim3(256), 0, s, b.keys[slot], R, img.ranges);
  GM_LAUNCH_CHECK(debug, s);
  hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, img.ranges, tiles, img.tile_order);
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

}  // namespace gm
