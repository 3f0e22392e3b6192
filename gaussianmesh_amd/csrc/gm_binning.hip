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
#include "gm_cull.h"
#pragma clang fp contract(off)

namespace gm {

#define BN_THREADS 256
#define BN_PER_THREAD (GM_SCAN_ITEMS / BN_THREADS)   // 2
#define DUP_STAGE 4096                               // instances a block can assemble in LDS (2 x 16 KB)

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

// Instance emission.  Workgroup b owns sorted positions [b*512, (b+1)*512); thread t owns 2 consecutive ones and
// gathers their bin records (candidate rectangle + emit mask, written by preprocess) and instance counts; the block
// scans the counts into output offsets; then instances are written (see (1) and (2) in the body).
// Emitted order = Gaussian order (depth, id), then rectangle row-major - the reference's order
// (RAST/rasterizer_impl.cu:98-109) restricted to the emitted tiles.
__global__ __launch_bounds__(BN_THREADS) void duplicate_kernel(const uint32_t* __restrict__ order,
                                                                const uint4* __restrict__ bins, const uint32_t* __restrict__ tiles,
                                                                const float4* __restrict__ splat, const uint32_t* __restrict__ counters, int P, int gx,
                                                                const uint32_t* __restrict__ block_sums,
                                                                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
  __shared__ uint32_t wsum[BN_THREADS / 64];
  __shared__ uint32_t stage_k[DUP_STAGE], stage_v[DUP_STAGE];
  const int tile_cull = (int)counters[2];        // policy recorded by preprocess_fwd_kernel (the counts were made with it)
  const int lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int base = blockIdx.x * GM_SCAN_ITEMS + threadIdx.x * BN_PER_THREAD;
  uint32_t gid[BN_PER_THREAD], cnt[BN_PER_THREAD], offs[BN_PER_THREAD], sum = 0;
  uint4 rc[BN_PER_THREAD];
#pragma unroll
  for (int i = 0; i < BN_PER_THREAD; i++) {
    const int s = base + i;
    gid[i] = s < P ? order[s] : 0u;
  }
#pragma unroll
  for (int i = 0; i < BN_PER_THREAD; i++) {
    const int s = base + i;
    cnt[i] = s < P ? tiles[gid[i]] : 0u;
    rc[i] = (s < P && cnt[i]) ? bins[gid[i]] : make_uint4(0u, 0u, 0u, 0u);
    sum += cnt[i];
  }
  uint32_t total;
  const uint32_t block_base = block_sums[blockIdx.x];
  uint32_t off = block_exclusive_scan(sum, wsum, total);
  // The block's instances form one contiguous output range.  When it fits the LDS stage (nearly always) they are
  // assembled there and copied out with full-line stores: per-lane runs written straight to HBM cost 2.3x the bytes
  // (partial lines, WRITE_SIZE 114 MB for 48.7 MB of pairs).  Otherwise the runs are stored directly.
  const bool staged = total <= DUP_STAGE;
  uint32_t* __restrict__ kdst = staged ? stage_k : keys_out + block_base;
  uint32_t* __restrict__ vdst = staged ? stage_v : vals_out + block_base;
#pragma unroll
  for (int i = 0; i < BN_PER_THREAD; i++) { offs[i] = off; off += cnt[i]; }
  // (1) rectangles of <= 64 tiles (virtually all): each lane expands the emit mask of its own Gaussian.  A wave
  //     spends max-over-lanes(popcount) iterations of ~8 instructions for 64 Gaussians; the 4-byte stores of a lane go
  //     to its own run, neighbouring lanes' runs are adjacent, so a wave writes one compact region.
#pragma unroll
  for (int i = 0; i < BN_PER_THREAD; i++) {
    const uint32_t w = rc[i].y & 0xFFFFu, ncand = w * (rc[i].y >> 16);
    if (cnt[i] != 0 && ncand <= 64) {
      const uint32_t x0 = rc[i].x & 0xFFFFu, y0 = rc[i].x >> 16;
      const float inv_w = 1.0f / (float)w;
      unsigned long long m = ((unsigned long long)rc[i].w << 32) | rc[i].z;
      uint32_t pos = offs[i];
      const uint32_t tile0 = y0 * (uint32_t)gx + x0;
      while (m) {
        const uint32_t k = (uint32_t)__ffsll(m) - 1u;
        m &= m - 1;
        const uint32_t row = (uint32_t)(((float)k + 0.5f) * inv_w);
        kdst[pos] = tile0 + row * (uint32_t)gx + (k - row * w);
        vdst[pos] = gid[i];
        pos++;
      }
    }
  }
  // (2) rectangles of more than 64 tiles: the wave walks them one at a time, 64 candidate tiles per step
#pragma unroll
  for (int i = 0; i < BN_PER_THREAD; i++) {
    const uint32_t wi = rc[i].y & 0xFFFFu;
    unsigned long long todo = __ballot(cnt[i] != 0 && wi * (rc[i].y >> 16) > 64);
    while (todo) {
      const int j = __ffsll(todo) - 1;
      todo &= todo - 1;
      const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)offs[i], j);
      const uint32_t rx = (uint32_t)__builtin_amdgcn_readlane((int)rc[i].x, j);
      const uint32_t ry = (uint32_t)__builtin_amdgcn_readlane((int)rc[i].y, j);
      const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)gid[i], j);
      const uint32_t x0 = rx & 0xFFFFu, y0 = rx >> 16, w = ry & 0xFFFFu, h = ry >> 16, ncand = w * h;
      const float inv_w = 1.0f / (float)w;
      const float4 s0 = splat[3 * (size_t)g], s1 = splat[3 * (size_t)g + 1];
      const TileCull tc = tile_cull_setup(s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, (float)(x0 * GM_TILE), (float)((x0 + w) * GM_TILE - 1),
                                          (float)(y0 * GM_TILE), (float)((y0 + h) * GM_TILE - 1));
      uint32_t run = o;
      for (uint32_t c0 = 0; c0 < ncand; c0 += 64) {
        const uint32_t k = c0 + lane;
        const uint32_t row = (uint32_t)(((float)k + 0.5f) * inv_w), col = k - row * w;
        bool pass = k < ncand;
        if (pass && tile_cull) {                   // same per-row span as preprocess used for the count
          int ta, tb;
          pass = row_tiles(tc, s0.x, s0.y, (int)(y0 + row), (int)x0, (int)(x0 + w), ta, tb) && (int)(x0 + col) >= ta && (int)(x0 + col) <= tb;
        }
        const unsigned long long bal = __ballot(pass);
        if (pass) {
          const uint32_t pos = run + (uint32_t)__popcll(bal & lt_mask);
          kdst[pos] = (y0 + row) * (uint32_t)gx + (x0 + col);
          vdst[pos] = g;
        }
        run += (uint32_t)__popcll(bal);
      }
    }
  }
  if (staged) {
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < total; j += BN_THREADS) {
      keys_out[block_base + j] = stage_k[j];
      vals_out[block_base + j] = stage_v[j];
    }
  }
}

int launch_duplicate(GeomState& g, BinningState& b, int P, int W, int H, int tile_cull, int debug, hipStream_t s) {
  StageScope sc(ST_DUPLICATE, s);
  const int nb = (P + GM_SCAN_ITEMS - 1) / GM_SCAN_ITEMS;
  const int gx = (W + GM_TILE - 1) / GM_TILE;
  (void)H; (void)tile_cull;
  if (nb > 0)
    hipLaunchKernelGGL(duplicate_kernel, dim3(nb), dim3(BN_THREADS), 0, s, g.order[0], g.bin, g.tiles_touched, g.splat, g.counters,
                       P, gx, g.block_sums, b.keys[0], b.vals[0]);
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

int launch_tile_ranges(BinningState& b, int slot, ImageState& img, int R, int tiles, int debug, hipStream_t s) {
  StageScope sc(ST_RANGES, s);
  GM_HIP(hipMemsetAsync(img.ranges, 0, sizeof(uint2) * (size_t)tiles, s));
  if (R > 0) hipLaunchKernelGGL(tile_ranges_kernel, dim3((R + 255) / 256), dim3(256), 0, s, b.keys[slot], R, img.ranges);
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

}  // namespace gm
