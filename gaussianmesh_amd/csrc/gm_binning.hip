// gm_binning.hip -- instance emission and (for more than 2048 list tiles) tile ranges.
//
// Replaces (reference, RAST = gaussian_renderer/diff_gaussian_rasterizater/cuda_rasterizer):
//   RAST/rasterizer_impl.cu:70-111 duplicateWithKeys
//   RAST/rasterizer_impl.cu:116-138 identifyTileRanges (+ the cudaMemset of ranges, :485)
// (the inclusive scan of tiles_touched, :407, is folded into the ordering: bucket_sort_kernel leaves per-bucket instance
//  totals, gm_bucket.hip)
//
// Gaussians are visited in (depth, id) order (GeomState::order[0]), so the emitted instance stream is already
// depth-ordered and only needs a stable sort by list tile afterwards.
#include "gm_common.h"
#include "gm_cull.h"
#pragma clang fp contract(off)

namespace gm {

#define BN_THREADS 256
#define DUP_STAGE 2048                               // instances a block can assemble in LDS (2 x 8 KB; 4096 is no faster alone and makes the
                                                     // workgroup harder to place next to other frames' blend kernels: 4350 -> 4400 frames/s)

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

// the armed blocks of the frames of a batch in one launch (a single frame: one memset)
__global__ __launch_bounds__(256) void arm_batch_kernel(uint4* __restrict__ arm, uint32_t arm_vec, const FrameOfs go) {
  arm = frame_ptr(arm, go);
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < arm_vec; i += gridDim.x * 256) arm[i] = z;
}

int launch_arm_counters(GeomState& g, hipStream_t s, const BatchOfs* bt) {
  if (bt && bt->frames > 1) {
    hipLaunchKernelGGL(arm_batch_kernel, dim3(32, 1, (uint32_t)bt->frames), dim3(256), 0, s, reinterpret_cast<uint4*>(g.slots), (uint32_t)(g.arm_words / 4), bt->geom);
    GM_HIP(hipGetLastError());
    return 0;
  }
  GM_HIP(hipMemsetAsync(g.slots, 0, sizeof(uint32_t) * g.arm_words, s));
  return 0;
}

__device__ __forceinline__ void get_rect(float px, float py, int r, int gx, int gy, int& x0, int& y0, int& x1, int& y1) {
  x0 = min(gx, max(0, (int)((px - r) / GM_TILE)));
  y0 = min(gy, max(0, (int)((py - r) / GM_TILE)));
  x1 = min(gx, max(0, (int)((px + r + GM_TILE - 1) / GM_TILE)));
  y1 = min(gy, max(0, (int)((py + r + GM_TILE - 1) / GM_TILE)));
}

// Instance emission.  Workgroup r owns the run of sorted positions [512 r, 512 r + 512) (gm_bucket.hip leaves the instance
// total of every run in chunk_inst; the work does not depend on how the depth buckets came out): thread t owns 2 consecutive
// positions and reads their
// emission records (candidate rectangle + instance count + emit mask, written by preprocess, brought into this order by
// bucket_sort_kernel: coalesced); the chunk scans the counts into output offsets; then instances are
// written (see (1) and (2) in the body).  The run's first output offset is the sum of the instance totals of the runs before
// it (chunk_inst, summed by every workgroup for itself).
// Emitted order = Gaussian order (depth, id), then rectangle row-major - the reference's order
// (RAST/rasterizer_impl.cu:98-109) restricted to the emitted tiles.
// S > 0: one instance per PARENT tile (2^S x 2^S tiles) that has a reached child; key = parent id | child mask << 16
// (child bit = (row in parent) << S | column in parent).  S == 0: key = tile id | 1 << 16.
template <int S, int BN_PER_THREAD>
__global__ __launch_bounds__(BN_THREADS) void duplicate_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ tiles,
                                                                const uint4* __restrict__ bin_sorted, const float4* __restrict__ splat,
                                                                uint32_t* __restrict__ counters, const uint32_t* __restrict__ chunk_inst,
                                                                int gx, int pgx, int mode,
                                                                uint32_t capacity, uint2* __restrict__ pairs_out,
                                                                uint32_t* __restrict__ acc, uint32_t acc_words, const FrameOfs go, const FrameOfs bo) {
  order = frame_ptr(order, go); tiles = frame_ptr(tiles, go); bin_sorted = frame_ptr(bin_sorted, go); splat = frame_ptr(splat, go);
  counters = frame_ptr(counters, go); chunk_inst = frame_ptr(chunk_inst, go);            // frame blockIdx.z of a batch (gm_common.h FrameOfs)
  pairs_out = frame_ptr(pairs_out, bo); acc = frame_ptr(acc, bo);
  __shared__ uint32_t wsum[BN_THREADS / 64];
  __shared__ uint32_t stage_k[DUP_STAGE], stage_v[DUP_STAGE];
  // accumulators of the tile pass that follows (bk_hist_kernel adds into them)
  for (uint32_t i = blockIdx.x * BN_THREADS + threadIdx.x; i < acc_words; i += gridDim.x * BN_THREADS) acc[i] = 0u;
  // the counts were made under another policy (it changed between the forward halves), or the instance total exceeds the
  // caller's binning capacity: emit nothing rather than overrun; every list stays empty (tile_ranges / bk_scan see the flag)
  // ... or the direct depth placement could not order the frame (status 2: the caller begins it again on the partition path)
  const uint32_t direct_fail = counters[GM_CNT_DIRECT_FAIL];
  if ((int)counters[GM_CNT_POLICY] != mode || counters[GM_CNT_RENDERED] > capacity || direct_fail) {
    if (threadIdx.x == 0) counters[GM_CNT_REFUSED] = direct_fail ? 2u : 1u;
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) counters[GM_CNT_REFUSED] = 0u;      // (a refused attempt on these buffers may have left it set)
  const int tile_cull = mode != 0;
  constexpr int M = (1 << S) - 1;
  const int lane = threadIdx.x & 63;
  const uint32_t run = blockIdx.x;
  const uint32_t V = counters[GM_CNT_VISIBLE];
  constexpr uint32_t RUN = BN_PER_THREAD * BN_THREADS;                  // sorted positions of this workgroup: 1 or 2 entries of chunk_inst
  static_assert(RUN % GM_SCAN_ITEMS == 0, "a workgroup takes whole runs");
  const uint32_t P0 = min(run * RUN, V), P1 = min(P0 + RUN, V);
  if (P0 == P1) return;
  uint32_t ibase;
  {
    uint32_t part = 0;
    for (uint32_t k = threadIdx.x; k < run * (RUN / GM_SCAN_ITEMS); k += BN_THREADS) part += chunk_inst[k];
    uint32_t tot;
    block_exclusive_scan(part, wsum, tot);
    ibase = tot;
    __syncthreads();
  }
  for (uint32_t c0 = P0; c0 < P1; c0 += RUN) {
  const uint32_t base = c0 + threadIdx.x * BN_PER_THREAD;
  uint32_t gid[BN_PER_THREAD], cnt[BN_PER_THREAD], offs[BN_PER_THREAD], sum = 0;
  uint4 rc[BN_PER_THREAD];
#pragma unroll
  for (int i = 0; i < BN_PER_THREAD; i++) {
    const uint32_t s = base + i;
    rc[i] = s < P1 ? bin_sorted[s] : make_uint4(0u, 0u, 0u, 0u);
    cnt[i] = bin_count(rc[i]);
    gid[i] = (s < P1 && cnt[i]) ? order[s] : 0u;
    if (cnt[i] == GM_BIN_COUNT_SAT) cnt[i] = tiles[gid[i]];
    sum += cnt[i];
  }
  uint32_t total;
  const uint32_t block_base = ibase;
  uint32_t off = block_exclusive_scan(sum, wsum, total);
  // The chunk's instances form one contiguous output range.  When it fits the LDS stage (nearly always) they are
  // assembled there and copied out with full-line stores: per-lane runs written straight to HBM cost 2.3x the bytes
  // (partial lines, WRITE_SIZE 114 MB for 48.7 MB of pairs).  Otherwise the runs are stored directly.
  const bool staged = total <= DUP_STAGE;
  uint32_t* __restrict__ kdst = stage_k;
  uint32_t* __restrict__ vdst = stage_v;
  // not staged: pairs go straight to their slots (kdst / vdst unused); EMIT covers both
#define EMIT(POS, K, V) do { if (staged) { kdst[POS] = (K); vdst[POS] = (V); } else pairs_out[block_base + (POS)] = make_uint2((K), (V)); } while (0)
#pragma unroll
  for (int i = 0; i < BN_PER_THREAD; i++) { offs[i] = off; off += cnt[i]; }
  // (1) small rectangles (<= 64 tiles; S > 0: also at most 60 wide): each lane expands the emit mask of its own
  //     Gaussian.  The 4-byte stores of a lane go to its own run, neighbouring lanes' runs are adjacent.
#pragma unroll
  for (int i = 0; i < BN_PER_THREAD; i++) {
    const uint32_t w = bin_w(rc[i]), h = bin_h(rc[i]), ncand = w * h;
    const bool small = ncand <= 64 && (S == 0 || w <= 60);
    if (cnt[i] != 0 && small) {
      const uint32_t x0 = bin_x0(rc[i]), y0 = bin_y0(rc[i]);
      unsigned long long m = ((unsigned long long)rc[i].w << 32) | rc[i].z;
      uint32_t pos = offs[i];
      if (S == 0) {
        const float inv_w = 1.0f / (float)w;
        const uint32_t tile0 = (y0 * (uint32_t)gx + x0) | (1u << GM_KEY_MASK_SHIFT);
        while (m) {
          const uint32_t k = (uint32_t)__ffsll(m) - 1u;
          m &= m - 1;
          const uint32_t row = (uint32_t)(((float)k + 0.5f) * inv_w);
          EMIT(pos, tile0 + row * (uint32_t)gx + (k - row * w), gid[i]);
          pos++;
        }
      } else if (S == 1 && quad_rect(x0, y0, w, h)) {
        // quadrant-level record (gm_cull.h): the mask IS the four parents' 16-bit quadrant masks, in parent-row-major order
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
          const uint32_t cm = (uint32_t)(m >> (16 * k)) & 0xFFFFu;
          if (cm) {
            EMIT(pos, (((y0 >> 1) + (k >> 1)) * (uint32_t)pgx + (x0 >> 1) + (k & 1u)) | (cm << GM_KEY_MASK_SHIFT), gid[i]);
            pos++;
          }
        }
      } else {
        const unsigned long long rowmask = (1ull << w) - 1ull;
        const int a = (int)(x0 & M);
        for (uint32_t pr = y0 >> S; pr <= (y0 + h - 1) >> S; pr++) {
          unsigned long long rows[1 << S], u = 0ull;
#pragma unroll
          for (int j = 0; j <= M; j++) {                 // child rows of this parent row that lie inside the rectangle
            const int rr = (int)((pr << S) + j) - (int)y0;
            rows[j] = (rr >= 0 && rr < (int)h) ? ((m >> (rr * (int)w)) & rowmask) << a : 0ull;
            u |= rows[j];
          }
          unsigned long long g = S == 1 ? ((u | (u >> 1)) & 0x5555555555555555ull) : ((u | (u >> 1) | (u >> 2) | (u >> 3)) & 0x1111111111111111ull);
          while (g) {
            const int b = __ffsll(g) - 1;
            g &= g - 1;
            uint32_t cm = 0;
#pragma unroll
            for (int j = 0; j <= M; j++) cm |= (uint32_t)((rows[j] >> b) & (unsigned long long)((1 << (1 << S)) - 1)) << (j << S);
            if (S == 1) cm = tile_to_quad_mask(cm);       // policy 2 keys carry quadrant bits
            EMIT(pos, (pr * (uint32_t)pgx + (x0 >> S) + (uint32_t)(b >> S)) | (cm << GM_KEY_MASK_SHIFT), gid[i]);
            pos++;
          }
        }
      }
    }
  }
  // (2) the other rectangles: the wave walks them one at a time, 64 candidate (parent) tiles per step
#pragma unroll
  for (int i = 0; i < BN_PER_THREAD; i++) {
    const uint32_t wi = bin_w(rc[i]), nci = wi * bin_h(rc[i]);
    unsigned long long todo = __ballot(cnt[i] != 0 && !(nci <= 64 && (S == 0 || wi <= 60)));
    while (todo) {
      const int j = __ffsll(todo) - 1;
      todo &= todo - 1;
      const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)offs[i], j);
      const uint32_t rx = (uint32_t)__builtin_amdgcn_readlane((int)rc[i].x, j);
      const uint32_t ry = (uint32_t)__builtin_amdgcn_readlane((int)rc[i].y, j);
      const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)gid[i], j);
      const uint32_t x0 = rx & 0xFFFu, y0 = (rx >> 16) & 0xFFFu, w = ry & 0xFFFu, h = (ry >> 16) & 0xFFFu;
      const uint32_t px0 = x0 >> S, py0 = y0 >> S, pw = ((x0 + w - 1) >> S) - px0 + 1, ph = ((y0 + h - 1) >> S) - py0 + 1;
      const float inv_w = 1.0f / (float)pw;
      const float4 s0 = splat_row(splat, (size_t)g, 0), s1 = splat_row(splat, (size_t)g, 1);
      const TileCull tc = tile_cull_setup(s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, (float)(x0 * GM_TILE), (float)((x0 + w) * GM_TILE - 1),
                                          (float)(y0 * GM_TILE), (float)((y0 + h) * GM_TILE - 1));
      uint32_t run = o;
      // The row spans are computed ONCE per tile row (lane = tile row; 64 tile rows = 64 >> S parent rows per chunk) and handed to the
      // parents' lanes by cross-lane reads - every lane of a parent row walking the same 2^S rows again cost ~4x the instructions for
      // the rectangles of near Gaussians (C5's emission: 204 us).  Parents are taken in row-major order, 64 per step, as before.
      constexpr uint32_t PR_CHUNK = 64u >> S;
      for (uint32_t pr0 = 0; pr0 < ph; pr0 += PR_CHUNK) {
        int ta = GM_ROW_EMPTY_LO, tb = -1;                        // span of tile row ((py0 + pr0) << S) + lane
        {
          const int ty = (int)((py0 + pr0) << S) + lane;
          if (ty >= (int)y0 && ty < (int)(y0 + h)) {
            int ra = (int)x0, rb = (int)(x0 + w) - 1;
            if (!tile_cull || row_tiles(tc, s0.x, s0.y, ty, (int)x0, (int)(x0 + w), ra, rb)) { ta = ra; tb = rb; }
          }
        }
        const uint32_t nrows = min(PR_CHUNK, ph - pr0), nchunk = nrows * pw;
        for (uint32_t c0 = 0; c0 < nchunk; c0 += 64) {
          const uint32_t k = c0 + lane;
          const uint32_t row = min((uint32_t)(((float)k + 0.5f) * inv_w), nrows - 1u), col = k - row * pw;
          const uint32_t pcx = px0 + col, pcy = py0 + pr0 + row;
          uint32_t cm = 0;
          int lo = 0x7fffffff, hi = -1;                           // hull over the child rows, as preprocess counted
#pragma unroll
          for (int jr = 0; jr <= M; jr++) {
            const int src = (int)(row << S) + jr;                 // the lane that holds that tile row (< 64)
            const int ra = __shfl(ta, src), rb = __shfl(tb, src);
            if (rb >= ra) {
              lo = min(lo, ra); hi = max(hi, rb);
#pragma unroll
              for (int jc = 0; jc <= M; jc++) {
                const int tx = (int)(pcx << S) + jc;
                if (tx >= ra && tx <= rb) cm |= 1u << ((jr << S) + jc);
              }
            }
          }
          const bool pass = k < nchunk && hi >= 0 && (int)pcx >= (lo >> S) && (int)pcx <= (hi >> S);
          const unsigned long long bal = __ballot(pass);
          if (pass) {
            const uint32_t pos = run + lanes_below(bal);
            if (S == 1) cm = tile_to_quad_mask(cm);
            EMIT(pos, (pcy * (uint32_t)pgx + pcx) | (cm << GM_KEY_MASK_SHIFT), g);
          }
          run += (uint32_t)__popcll(bal);
        }
      }
    }
  }
  if (staged) {
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < total; j += BN_THREADS) pairs_out[block_base + j] = make_uint2(stage_k[j], stage_v[j]);
  }
#undef EMIT
  ibase += total;
  __syncthreads();                                 // the stage and wsum are reused by the next chunk
  }
}

int launch_duplicate(GeomState& g, BinningState& b, int P, int W, int H, int mode, size_t capacity, int debug, hipStream_t s, const BatchOfs* bt) {
  StageScope sc(ST_DUPLICATE, s);
  const TileGrid tg(W, H, mode);
  const BatchOfs one = single_frame();
  const BatchOfs& B = bt ? *bt : one;
  const uint32_t cap = capacity > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)capacity;
  if (P > 0) {
    // two Gaussians per thread while a workgroup's instances still fit its LDS stage (a run of 512 at C3's ~3 instances per
    // Gaussian), one when the cloud emits more per Gaussian (4K, near cameras): unstaged runs store partial lines
    const bool two = capacity <= (size_t)P * 5;
#define GM_DUP(SH, PER) hipLaunchKernelGGL((duplicate_kernel<SH, PER>), dim3((P + PER * BN_THREADS - 1) / (PER * BN_THREADS), 1, (uint32_t)B.frames), dim3(BN_THREADS), 0, s, g.order, \
                                           g.tiles_touched, g.bin_sorted, g.splat, g.counters, g.chunk_inst, tg.gx, tg.pgx, mode, cap, b.pairs[0], b.acc, \
                                           (uint32_t)bk_acc_words(capacity), B.geom, B.binning)
    if (two) { if (tg.s == 0) GM_DUP(0, 2); else if (tg.s == 1) GM_DUP(1, 2); else GM_DUP(2, 2); }
    else { if (tg.s == 0) GM_DUP(0, 1); else if (tg.s == 1) GM_DUP(1, 1); else GM_DUP(2, 1); }
#undef GM_DUP
  }
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint2* __restrict__ pairs, int R_host, const uint32_t* __restrict__ R_dev,
                                                          uint32_t tiles, const uint32_t* __restrict__ counters, uint2* __restrict__ ranges) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int R = R_dev ? (int)min((uint32_t)R_host, *R_dev) : R_host;
  if (i >= R || counters[GM_CNT_REFUSED] != 0u) return;        // emission was refused -> every list stays empty
  const uint32_t cur = pairs[i].x & GM_KEY_TILE_MASK;
  if (cur >= tiles) return;
  if (i == 0) ranges[cur].x = 0;
  else {
    const uint32_t prev = pairs[i - 1].x & GM_KEY_TILE_MASK;
    if (cur != prev) { if (prev < tiles) ranges[prev].y = (uint32_t)i; ranges[cur].x = (uint32_t)i; }
  }
  if (i == R - 1) ranges[cur].y = (uint32_t)R;
}

// only for more than 2048 list tiles (two-pass tile sort); otherwise bk_scan_kernel writes the ranges
int launch_tile_ranges(const GeomState& g, BinningState& b, int slot, ImageState& img, int R, const uint32_t* R_dev, int tiles, int debug,
                       hipStream_t s) {
  StageScope sc(ST_RANGES, s);
  GM_HIP(hipMemsetAsync(img.ranges, 0, sizeof(uint2) * (size_t)tiles, s));
  if (R > 0) hipLaunchKernelGGL(tile_ranges_kernel, dim3((R + 255) / 256), dim3(256), 0, s, b.pairs[slot], R, R_dev, (uint32_t)tiles, g.counters, img.ranges);
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

}  // namespace gm
