// gm_render.hip -- per-tile alpha blending, forward and backward.
//
// Replaces (reference, RAST = gaussian_renderer/diff_gaussian_rasterizater/cuda_rasterizer):
//   RAST/forward.cu:261-374   renderCUDA (forward)
//   RAST/backward.cu:399-557  renderCUDA (backward)
//
// CDNA4 mapping: the reference gives each 16x16 tile to a 256-thread block that stages
// 256-entry batches in shared memory behind two block barriers and lets every pixel thread re-read the batch from LDS
// (and the colour from global memory).  Here a 16x16 tile is four independent wave64s, each owning one 8x8 pixel
// quadrant (one pixel per lane, lane = y * 8 + x):
//   * the workgroup walks the list of its PARENT tile (gm_common.h: emission policies) and takes the entries whose key
//     carries its child bit; a wave gathers 64 list entries at a time (lane j <- entry j: key, id, three 16-byte splat
//     loads) through a three-deep software pipeline with a constant number of loads per iteration;
//   * while lane j holds entry j it tests, once per batch and for all 64 entries in parallel, whether the entry can
//     reach alpha >= 1/255 anywhere inside the bounding box of the quadrant's still-live pixels (exact minimum of the
//     conic's quadratic form over the rectangle, with a rounding margin).  A 64-bit ballot of the survivors drives the
//     inner loop (s_ff1 over set bits).  The cull is conservative: a culled entry would have been skipped by every
//     live pixel (alpha < 1/255), so results and n_contrib are unchanged;
//   * survivors are re-read as LDS broadcasts from a wave-private copy of the batch (conic pre-multiplied for the
//     exp2 argument) and processed two at a time; no s_barrier, no per-pixel global colour read;
//   * "is every pixel done" is a ballot instead of __syncthreads_count; an entry that no pixel of the wave accepts
//     is skipped with one __any.
// Discrete semantics are the reference's: skip power>0, skip alpha<1/255, stop (without applying the
// entry) when T(1-alpha)<1e-4, n_contrib = 1-based list position of the last accepted entry.
// FMA contraction is allowed here and exp() is v_exp_f32 on power*log2(e); see DESIGN.md for the
// tolerance argument.
// PPL (pixels per lane) and QUAD (8x8 quadrant instead of a 16x4 strip) are fixed at 1 / true: the other shapes were
// measured slower and are no longer instantiated; the loops over k < PPL are kept trivial.
#include "gm_common.h"
#include "gm_cull.h"
#include <cstdlib>

namespace gm {

#define LOG2E 1.4426950408889634f
constexpr int PPL = 1;          // pixels per lane
constexpr bool QUAD = true;     // a wave owns an 8x8 pixel quadrant of its tile

__device__ __forceinline__ float bcast(float v, int j) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}

// Workgroup -> 16-px tile, its parent tile (whose list it walks) and its bit in the instance keys' child mask.
// Consecutive workgroup ids go to consecutive XCDs (8 of them, each with its own L2), so the 2^s x 2^s children of
// a parent are given ids that are 8 apart: they run on the same XCD and share the parent's list and records in L2.
struct TileMap {
  int gx, gy, pgx, pgy, s;
  __device__ __forceinline__ bool locate(int b, int& tx, int& ty, int& parent, uint32_t& child_bit) const {
    const int nch = 1 << (2 * s);
    const int xcd = b & 7, j = b >> 3;
    const int p = (j >> (2 * s)) * 8 + xcd, c = j & (nch - 1);
    if (p >= pgx * pgy) return false;
    const int py = p / pgx, px = p - py * pgx;
    const int cy = c >> s, cx = c & ((1 << s) - 1);
    tx = (px << s) + cx; ty = (py << s) + cy;
    parent = p;
    child_bit = 1u << (GM_KEY_MASK_SHIFT + c);
    return tx < gx && ty < gy;
  }
  int blocks() const { return ((pgx * pgy + 7) / 8) * 8 << (2 * s); }
};

struct Batch {      // lane j holds list entry j of the current 64-entry batch
  float4 a;         // x, y, conic.x, conic.y
  float4 b;         // conic.z, opacity, r, g
  float c;          // b
  uint32_t id;
};

// Two-deep software pipeline of the dependent gather (list position -> Gaussian id -> 48-byte splat record):
// ids are fetched two batches ahead and records one batch ahead, so neither latency sits on the critical path
// of a wave that walks a multi-thousand-entry list.  WAIT_ALL_LOADS() at the top of each iteration retires
// the loads issued one iteration earlier (they had a whole batch of arithmetic to land); placing the wait
// there, explicitly, also stops the compiler from emitting vmcnt(0) in the middle of the batch, which would
// serialise the freshly issued prefetch.
#define WAIT_ALL_LOADS() __builtin_amdgcn_s_waitcnt(0x0F70)   /* vmcnt(0), expcnt/lgkmcnt untouched */

__device__ __forceinline__ uint32_t load_id(const uint32_t* __restrict__ list, int e, int n) {
  return (e >= 0 && e < n) ? list[e] : 0u;
}
__device__ __forceinline__ Batch load_records(const float4* __restrict__ splat, uint32_t id, bool live) {
  Batch t;
  t.a = make_float4(0.f, 0.f, 0.f, 0.f); t.b = t.a; t.c = 0.f; t.id = id;
  if (live) {
    t.a = splat[3 * (size_t)id];
    t.b = splat[3 * (size_t)id + 1];
    t.c = splat[3 * (size_t)id + 2].x;
  }
  return t;
}


__global__ __launch_bounds__(256) void render_fwd_kernel(const uint2* __restrict__ ranges,
                                                               const uint2* __restrict__ pairs,
                                                               const float4* __restrict__ splat, int W, int H, TileMap tm,
                                                               const float* __restrict__ bg, float* __restrict__ out_color,
                                                               float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int tx, ty, parent;
  uint32_t child_bit;
  if (!tm.locate(blockIdx.x, tx, ty, parent, child_bit)) return;
  const uint2 range = ranges[parent];
  const int n = (int)(range.y - range.x);
  const uint2* list = pairs + range.x;           // (key, Gaussian id) per list entry

  const int px = QUAD ? tx * GM_TILE + (wave & 1) * 8 + (lane & 7) : tx * GM_TILE + (lane & 15);
  const float pixx = (float)px;
  int py[PPL];
  float pixy[PPL], T[PPL], Cr[PPL], Cg[PPL], Cb[PPL];
  uint32_t last[PPL];
  bool done[PPL];
#pragma unroll
  for (int k = 0; k < PPL; k++) {
    py[k] = QUAD ? ty * GM_TILE + (wave >> 1) * 8 + (lane >> 3) : ty * GM_TILE + (wave * PPL + k) * 4 + (lane >> 4);
    pixy[k] = (float)py[k];
    T[k] = 1.0f; Cr[k] = Cg[k] = Cb[k] = 0.f; last[k] = 0;
    done[k] = !(px < W && py[k] < H);
  }

  // pixel-centre rectangle owned by this wave
  const float rx0 = QUAD ? (float)(tx * GM_TILE + (wave & 1) * 8) : (float)(tx * GM_TILE);
  const float rx1 = rx0 + (QUAD ? 7.0f : (float)(GM_TILE - 1));
  const float ry0 = QUAD ? (float)(ty * GM_TILE + (wave >> 1) * 8) : (float)(ty * GM_TILE + wave * PPL * 4);
  const float ry1 = ry0 + (QUAD ? 7.0f : (float)(PPL * 4 - 1));

  // wave-private LDS copy of the current batch: survivors are re-read from here as LDS broadcasts (3 LDS
  // instructions, no VALU issue slots) instead of 9 v_readlane_b32 per survivor
  __shared__ float4 l_rec[4 / PPL][64][3];
  float4 (*rec)[3] = l_rec[wave];

  // Three-deep software pipeline of the dependent gather (list position -> id / key -> 48-byte splat record): ids and
  // keys run three batches ahead, records two, so a wave that finds few entries of its own per batch (little
  // arithmetic per iteration) does not advance at one memory round trip per batch.  Every iteration issues exactly
  // 2 + 3 loads (positions past the end re-read the last entry, entries of other child tiles read record 0), ids
  // first, so "all but the three youngest loads have landed" (vmcnt(3)) is exactly "this batch's records and the ids
  // needed to issue the next gathers are here; the gathers issued last iteration may still be in flight".
  // An entry of the parent's list concerns this tile iff its key has the tile's child bit.
  const int nlast = n - 1;
  auto ld_kv = [&](int e) { return list[min(e, nlast)]; };
  auto is_mine = [&](uint2 kv, int e) { return (e <= nlast) & ((kv.x & child_bit) != 0); };
  auto ld_rec = [&](uint32_t id, bool m) { return load_records(splat, m ? id : 0u, true); };
  if (n <= 0) {                                   // empty list: background only
    const size_t HW_ = (size_t)H * W;
    if (px < W && py[0] < H) {
      const size_t pid = (size_t)W * py[0] + px;
      final_T[pid] = 1.0f; n_contrib[pid] = 0u;
      out_color[pid] = bg[0]; out_color[HW_ + pid] = bg[1]; out_color[2 * HW_ + pid] = bg[2];
    }
    return;
  }
  const uint2 kv0 = ld_kv(lane), kv1 = ld_kv(64 + lane), kv2 = ld_kv(128 + lane);
  bool mine = is_mine(kv0, lane);
  Batch cur = ld_rec(kv0.y, mine);
  bool mine_1 = is_mine(kv1, 64 + lane);
  Batch nx1 = ld_rec(kv1.y, mine_1);
  uint32_t id_2 = kv2.y;
  bool mine_2 = is_mine(kv2, 128 + lane);
  const float qx0 = rx0, qy0 = ry0;
  for (int base = 0; base < n; base += 64) {
    bool all_done = true;
#pragma unroll
    for (int k = 0; k < PPL; k++) all_done = all_done && done[k];
    const unsigned long long live = __ballot(!all_done);
    if (live == 0ull) break;
    // Entries are culled against the bounding box of the pixels that are still live, not the whole 8x8 quadrant
    // (a quadrant kept alive by a few unsaturated pixels would otherwise evaluate every entry that touches any of
    // its 64 pixels).  Lane = y * 8 + x; scalar bit arithmetic.
    float cx0 = rx0, cx1 = rx1, cy0 = ry0, cy1 = ry1;
    if (QUAD && PPL == 1) {
      uint32_t cols = (uint32_t)live | (uint32_t)(live >> 32);
      cols |= cols >> 16; cols |= cols >> 8; cols &= 0xFFu;
      cx0 = qx0 + (float)(__ffs((int)cols) - 1);
      cx1 = qx0 + (float)(31 - __clz((int)cols));
      cy0 = qy0 + (float)((__ffsll(live) - 1) >> 3);
      cy1 = qy0 + (float)((63 - __clzll((long long)live)) >> 3);
    }
    __builtin_amdgcn_s_waitcnt(0x0F73);                                    // vmcnt(3)
    const uint2 kv3 = ld_kv(base + 192 + lane);                            // (key, id) pairs three batches ahead ...
    const uint32_t id_3 = kv3.y;
    const bool mine_3 = is_mine(kv3, base + 192 + lane);
    const Batch nx2 = ld_rec(id_2, mine_2);                                // ... records two batches ahead
    const bool keep = mine && may_touch(cur.a.x, cur.a.y, cur.a.z, cur.a.w, cur.b.x, cur.b.y, cx0, cx1, cy0, cy1);
    // staged record: the conic is stored pre-multiplied (lane-parallel, 3 multiplies per 64 entries) so that the
    // per-survivor exponent is e = dx (a' dx + b' dy) + (c' dy) dy = power * log2(e): 5 instructions instead of 9
    rec[lane][0] = make_float4(cur.a.x, cur.a.y, (-0.5f * LOG2E) * cur.a.z, (-LOG2E) * cur.a.w);
    rec[lane][1] = make_float4((-0.5f * LOG2E) * cur.b.x, cur.b.y, cur.b.z, cur.b.w);
    rec[lane][2] = make_float4(cur.c, 0.f, 0.f, 0.f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    unsigned long long todo = __ballot(keep);
    // Survivors are taken two at a time: the two alpha evaluations (the long dependent chain: quadratic form,
    // exp, min) are independent and interleave, only the short T/C recurrence is applied in order.  A wave that
    // is alone on its SIMD at the tail of a 10k-entry tile otherwise issues one dependent instruction every ~6 cycles.
    while (todo) {
      const int j0 = __ffsll(todo) - 1;
      todo &= todo - 1;
      const bool has1 = todo != 0;
      const int j1 = has1 ? __ffsll(todo) - 1 : j0;
      todo &= todo - 1;
      const float4 A0 = rec[j0][0], B0 = rec[j0][1], A1 = rec[j1][0], B1 = rec[j1][1];
      const float sx0 = A0.x, sy0 = A0.y, qa0 = A0.z, qb0 = A0.w, qc0 = B0.x, op0 = B0.y;
      const float sx1 = A1.x, sy1 = A1.y, qa1 = A1.z, qb1 = A1.w, qc1 = B1.x, op1 = B1.y;
      const float dx0 = sx0 - pixx, dx1 = sx1 - pixx;
      float wa0[PPL], wa1[PPL], t0[PPL], t1[PPL];
      bool v0[PPL], v1[PPL], anyv = false;
#pragma unroll
      for (int k = 0; k < PPL; k++) {
        const float dy0 = sy0 - pixy[k], dy1 = sy1 - pixy[k];
        const float e0 = dx0 * (qa0 * dx0 + qb0 * dy0) + (qc0 * dy0) * dy0;      // power * log2(e); same sign as power
        const float e1 = dx1 * (qa1 * dx1 + qb1 * dy1) + (qc1 * dy1) * dy1;
        const float alpha0 = fminf(0.99f, op0 * __builtin_amdgcn_exp2f(e0));
        const float alpha1 = fminf(0.99f, op1 * __builtin_amdgcn_exp2f(e1));
        // entry j0: weight alpha T, new transmittance T - alpha T (= T (1 - alpha))
        v0[k] = !done[k] && (e0 <= 0.0f) && (alpha0 >= 1.0f / 255.0f);
        wa0[k] = alpha0 * T[k];
        t0[k] = T[k] - wa0[k];
        const bool stop0 = v0[k] && (t0[k] < 0.0001f);
        done[k] = done[k] || stop0;
        v0[k] = v0[k] && !stop0;
        const float Tm = v0[k] ? t0[k] : T[k];
        // entry j1 (in list order after j0)
        v1[k] = has1 && !done[k] && (e1 <= 0.0f) && (alpha1 >= 1.0f / 255.0f);
        wa1[k] = alpha1 * Tm;
        t1[k] = Tm - wa1[k];
        const bool stop1 = v1[k] && (t1[k] < 0.0001f);
        done[k] = done[k] || stop1;
        v1[k] = v1[k] && !stop1;
        anyv = anyv || v0[k] || v1[k];
      }
      if (__any(anyv)) {
        const float r0 = B0.z, g0 = B0.w, b0 = rec[j0][2].x;
        const float r1 = B1.z, g1 = B1.w, b1 = rec[j1][2].x;
        const uint32_t c0 = (uint32_t)(base + j0 + 1), c1 = (uint32_t)(base + j1 + 1);
#pragma unroll
        for (int k = 0; k < PPL; k++) {
          const float w0 = v0[k] ? wa0[k] : 0.0f;
          const float Tm = v0[k] ? t0[k] : T[k];
          const float w1 = v1[k] ? wa1[k] : 0.0f;
          Cr[k] += r0 * w0; Cg[k] += g0 * w0; Cb[k] += b0 * w0;
          Cr[k] += r1 * w1; Cg[k] += g1 * w1; Cb[k] += b1 * w1;
          T[k] = v1[k] ? t1[k] : Tm;
          last[k] = v1[k] ? c1 : (v0[k] ? c0 : last[k]);
        }
      }
    }
    cur = nx1; nx1 = nx2; mine = mine_1; mine_1 = mine_2; id_2 = id_3; mine_2 = mine_3;
  }

  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t HW = (size_t)H * W;
#pragma unroll
  for (int k = 0; k < PPL; k++) {
    if (px < W && py[k] < H) {
      const size_t pid = (size_t)W * py[k] + px;
      final_T[pid] = T[k];
      n_contrib[pid] = last[k];
      out_color[pid] = Cr[k] + T[k] * bg0;
      out_color[HW + pid] = Cg[k] + T[k] * bg1;
      out_color[2 * HW + pid] = Cb[k] + T[k] * bg2;
    }
  }
}

int launch_render_fwd(const GeomState& g, const uint2* pairs, ImageState& img, int W, int H, int mode,
                      const float* background, float* out_color, int debug, hipStream_t s) {
  StageScope sc(ST_RENDER, s);
  const TileGrid tg(W, H, mode);
  const TileMap tm{tg.gx, tg.gy, tg.pgx, tg.pgy, tg.s};
  if (tg.ptiles > 0)
    hipLaunchKernelGGL(render_fwd_kernel, dim3(tm.blocks()), dim3(256), 0, s, img.ranges, pairs, g.splat, W, H, tm,
                       background, out_color, img.final_T, img.n_contrib);
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Backward blend.
//
// Per (wave, surviving entry) every lane holds nine partial sums (its pixels' contributions to dL/dcolor rgb,
// dL/dmean2D xy, dL/dconic x,y,w and dL/dopacity of that Gaussian).  The reference issues nine global atomics per
// (pixel, entry); here the wave first reduces them, with the gfx950 lane-swap instructions doing a TRANSPOSED
// reduction of eight values at once:
//   v_permlane32_swap + add : (q0,q1) -> one register holding q0's 32-lane partials in lanes 0-31 and q1's in 32-63
//   v_permlane16_swap + add : two such registers -> one register, one value per 16-lane row
//   row_ror:8 add + select  : two such registers -> one register, one value per 8-lane group
//   row_half_mirror, quad_perm[3,2,1,0], quad_perm[1,0,3,2] adds: finish inside the 8-lane groups
// = 18 VALU instructions for eight totals (48 with a plain 6-step DPP tree each), and the eight totals sit in eight
// different lanes, so ONE global_atomic_add_f32 instruction commits them.  The ninth value takes the plain DPP tree.
// Accumulation goes to a packed per-Gaussian record grad_acc[P][12] (one cache line instead of four arrays);
// preprocess_bwd_kernel unpacks it into the API's dL_dmean2D / dL_dconic / dL_dopacity / dL_dcolor.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int sh = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
  return v + __int_as_float(sh);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_add<0x111, 0xf>(v);   // row_shr:1
  v = dpp_add<0x112, 0xf>(v);   // row_shr:2
  v = dpp_add<0x114, 0xf>(v);   // row_shr:4
  v = dpp_add<0x118, 0xf>(v);   // row_shr:8
  v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1,3
  v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2,3
  return v;
}
__device__ __forceinline__ float swap32_add(float a, float b) {   // lanes 0-31: a[l]+a[l+32]; lanes 32-63: b[l-32]+b[l]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add(float a, float b) {   // rows: a.r0+a.r1 | b.r0+b.r1 | a.r2+a.r3 | b.r2+b.r3
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// totals of q0..q7; the total of q[slot] is returned in every lane of the 8-lane group with
// slot == reduce8_slot(lane)
__device__ __forceinline__ float reduce8(const float* q, int lane) {
  const float s01 = swap32_add(q[0], q[1]), s23 = swap32_add(q[2], q[3]);
  const float s45 = swap32_add(q[4], q[5]), s67 = swap32_add(q[6], q[7]);
  const float ta = swap16_add(s01, s23);            // rows: q0 q2 q1 q3
  const float tb = swap16_add(s45, s67);            // rows: q4 q6 q5 q7
  const float x = dpp_add<0x128, 0xf>(ta);          // row_ror:8 -> 8-lane partials, duplicated in both halves
  const float y = dpp_add<0x128, 0xf>(tb);
  float u = (lane & 8) ? y : x;
  u = dpp_add<0x141, 0xf>(u);                       // row_half_mirror
  u = dpp_add<0x1B, 0xf>(u);                        // quad_perm [3,2,1,0]
  u = dpp_add<0xB1, 0xf>(u);                        // quad_perm [1,0,3,2]
  return u;
}
__device__ __forceinline__ int reduce8_slot(int lane) {
  const int r = lane >> 4, h = (lane >> 3) & 1;
  const int a_idx = (r == 0) ? 0 : (r == 1) ? 2 : (r == 2) ? 1 : 3;
  return h ? a_idx + 4 : a_idx;                     // h = 1: q4 q6 q5 q7
}

#define GM_ACC_STRIDE 12   // floats per Gaussian in grad_acc: dcolor rgb (0-2), moments of h: 1, dx, dy, dx^2, dx dy, dy^2 (3-8)

__global__ __launch_bounds__(256) void render_bwd_kernel(const uint2* __restrict__ ranges,
                                                               const uint2* __restrict__ pairs,
                                                               const float4* __restrict__ splat, int W, int H, TileMap tm,
                                                               const float* __restrict__ bg, const float* __restrict__ final_T,
                                                               const uint32_t* __restrict__ n_contrib,
                                                               const float* __restrict__ dL_dpix, float* __restrict__ grad_acc,
                                                               const uint32_t* __restrict__ counters, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int tx, ty, parent;
  uint32_t child_bit;
  if ((int)counters[GM_CNT_POLICY] != mode || counters[GM_CNT_REFUSED] != 0u) return;   // lists were built under another emission policy: contribute nothing
  if (!tm.locate(blockIdx.x, tx, ty, parent, child_bit)) return;
  const uint2 range = ranges[parent];
  const int n = (int)(range.y - range.x);
  if (n == 0) return;
  const uint2* list = pairs + range.x;           // (key, Gaussian id) per list entry
  const size_t HW = (size_t)H * W;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];

  const int px = QUAD ? tx * GM_TILE + (wave & 1) * 8 + (lane & 7) : tx * GM_TILE + (lane & 15);
  const float pixx = (float)px;
  float pixy[PPL], T[PPL], T_final[PPL], last_alpha[PPL], bg_dot[PPL];
  float dpr[PPL], dpg[PPL], dpb[PPL], lcr[PPL], lcg[PPL], lcb[PPL], arr[PPL], arg_[PPL], arb[PPL];
  int last[PPL];
  int max_last = 0;
#pragma unroll
  for (int k = 0; k < PPL; k++) {
    const int py = QUAD ? ty * GM_TILE + (wave >> 1) * 8 + (lane >> 3) : ty * GM_TILE + (wave * PPL + k) * 4 + (lane >> 4);
    pixy[k] = (float)py;
    const bool inside = px < W && py < H;
    const size_t pid = inside ? (size_t)W * py + px : 0;
    T_final[k] = inside ? final_T[pid] : 0.f;
    T[k] = T_final[k];
    last[k] = inside ? (int)n_contrib[pid] : 0;
    dpr[k] = inside ? dL_dpix[pid] : 0.f;
    dpg[k] = inside ? dL_dpix[HW + pid] : 0.f;
    dpb[k] = inside ? dL_dpix[2 * HW + pid] : 0.f;
    bg_dot[k] = bg0 * dpr[k] + bg1 * dpg[k] + bg2 * dpb[k];
    last_alpha[k] = 0.f; lcr[k] = lcg[k] = lcb[k] = 0.f; arr[k] = arg_[k] = arb[k] = 0.f;
    max_last = max(max_last, last[k]);
  }
  // entries at list positions >= max over the wave of n_contrib are never used: start there
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) max_last = max(max_last, __shfl_xor(max_last, d));
  const int start = max_last;           // number of list entries this wave has to visit (positions start-1 .. 0)
  if (start == 0) return;

  const int my_slot = reduce8_slot(lane);
  const bool committer = (lane & 7) == 0;
  const float rx0 = QUAD ? (float)(tx * GM_TILE + (wave & 1) * 8) : (float)(tx * GM_TILE);
  const float rx1 = rx0 + (QUAD ? 7.0f : (float)(GM_TILE - 1));
  const float ry0 = QUAD ? (float)(ty * GM_TILE + (wave >> 1) * 8) : (float)(ty * GM_TILE + wave * PPL * 4);
  const float ry1 = ry0 + (QUAD ? 7.0f : (float)(PPL * 4 - 1));
  __shared__ float4 l_rec[4 / PPL][64][3];       // wave-private LDS copy of the batch (see render_fwd_kernel)
  float4 (*rec)[3] = l_rec[wave];
  // batch b covers list positions start-1-b*64-j (j = lane), i.e. back to front; same three-deep gather pipeline and
  // constant load count per iteration as render_fwd_kernel (positions below 0 re-read entry 0 and are not "mine")
  auto ld_kv = [&](int e) { return list[max(e, 0)]; };
  auto is_mine = [&](uint2 kv, int e) { return (e >= 0) & ((kv.x & child_bit) != 0); };
  auto ld_rec = [&](uint32_t id, bool m) { return load_records(splat, m ? id : 0u, true); };
  const uint2 kv0 = ld_kv(start - 1 - lane), kv1 = ld_kv(start - 1 - 64 - lane), kv2 = ld_kv(start - 1 - 128 - lane);
  bool mine = is_mine(kv0, start - 1 - lane);
  Batch cur = ld_rec(kv0.y, mine);
  bool mine_1 = is_mine(kv1, start - 1 - 64 - lane);
  Batch nx1 = ld_rec(kv1.y, mine_1);
  uint32_t id_2 = kv2.y;
  bool mine_2 = is_mine(kv2, start - 1 - 128 - lane);
  for (int base = 0; base < start; base += 64) {
    // A pixel takes part in this batch only if its last contributor lies above the batch's lowest position: cull
    // against the bounding box of those pixels (at the deep end of the walk only the few pixels that reached far
    // into the list are still in play).  Lane = y * 8 + x.
    float cx0 = rx0, cx1 = rx1, cy0 = ry0, cy1 = ry1;
    if (QUAD && PPL == 1) {
      const unsigned long long live = __ballot(last[0] > start - 64 - base);
      uint32_t cols = (uint32_t)live | (uint32_t)(live >> 32);
      cols |= cols >> 16; cols |= cols >> 8; cols &= 0xFFu;
      if (live != 0ull) {
        cx0 = rx0 + (float)(__ffs((int)cols) - 1);
        cx1 = rx0 + (float)(31 - __clz((int)cols));
        cy0 = ry0 + (float)((__ffsll(live) - 1) >> 3);
        cy1 = ry0 + (float)((63 - __clzll((long long)live)) >> 3);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F73);                                    // vmcnt(3)
    const uint2 kv3 = ld_kv(start - 1 - (base + 192) - lane);
    const uint32_t id_3 = kv3.y;
    const bool mine_3 = is_mine(kv3, start - 1 - (base + 192) - lane);
    const Batch nx2 = ld_rec(id_2, mine_2);
    const bool keep = mine && may_touch(cur.a.x, cur.a.y, cur.a.z, cur.a.w, cur.b.x, cur.b.y, cx0, cx1, cy0, cy1);
    // conic pre-multiplied for the exp2 argument, as in render_fwd_kernel (the moments below only need dx, dy)
    rec[lane][0] = make_float4(cur.a.x, cur.a.y, (-0.5f * LOG2E) * cur.a.z, (-LOG2E) * cur.a.w);
    rec[lane][1] = make_float4((-0.5f * LOG2E) * cur.b.x, cur.b.y, cur.b.z, cur.b.w);
    rec[lane][2] = make_float4(cur.c, __uint_as_float(cur.id), 0.f, 0.f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    unsigned long long todo = __ballot(keep);
    while (todo) {
      const int j = __ffsll(todo) - 1;
      todo &= todo - 1;
      const int pos = start - 1 - base - j;          // 0-based list position == reference `contributor`
      const float4 RA = rec[j][0], RB = rec[j][1];
      const float sx = RA.x, sy = RA.y, qa = RA.z, qb = RA.w, qc = RB.x, op = RB.y;
      const float dx = sx - pixx;
      float G[PPL], alpha[PPL], dy[PPL];
      bool valid[PPL], anyv = false;
#pragma unroll
      for (int k = 0; k < PPL; k++) {
        dy[k] = sy - pixy[k];
        const float e = dx * (qa * dx + qb * dy[k]) + (qc * dy[k]) * dy[k];      // power * log2(e); same sign as power
        G[k] = __builtin_amdgcn_exp2f(e);
        alpha[k] = fminf(0.99f, op * G[k]);
        valid[k] = (pos < last[k]) && (e <= 0.0f) && (alpha[k] >= 1.0f / 255.0f);
        anyv = anyv || valid[k];
      }
      if (!__any(anyv)) continue;
      const float4 RC = rec[j][2];
      const float r = RB.z, g = RB.w, b = RC.x;
      const uint32_t gid = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(RC.y));
      // per-lane partial sums of: dL/dcolor rgb (q0-2) and the six moments of h = G * dL/dG over the wave's pixels
      // (q3 = sum h, q4 = sum h dx, q5 = sum h dy, q6 = sum h dx^2, q7 = sum h dx dy, q8 = sum h dy^2).
      // preprocess_bwd_kernel turns the moments into dL/dopacity, dL/dmean2D and dL/dconic (backward.cu:538-554):
      //   dL/dopacity = q3 / opacity, dL/dmean2D = -(W/2)(cx q4 + cy q5), -(H/2)(cz q5 + cy q4), dL/dconic = -q6/2, -q7/2, -q8/2
      float q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, q8 = 0.f;
#pragma unroll
      for (int k = 0; k < PPL; k++) {
        if (valid[k]) {
          const float inv = __builtin_amdgcn_rcpf(1.f - alpha[k]);      // 1/(1-alpha): T recovery and the bg term
          T[k] = T[k] * inv;
          const float dchannel_dcolor = alpha[k] * T[k];
          arr[k] += last_alpha[k] * (lcr[k] - arr[k]);                  // accum_rec = la*lc + (1-la)*accum_rec
          arg_[k] += last_alpha[k] * (lcg[k] - arg_[k]);
          arb[k] += last_alpha[k] * (lcb[k] - arb[k]);
          lcr[k] = r; lcg[k] = g; lcb[k] = b;
          float dL_dalpha = (r - arr[k]) * dpr[k] + (g - arg_[k]) * dpg[k] + (b - arb[k]) * dpb[k];
          q[0] += dchannel_dcolor * dpr[k]; q[1] += dchannel_dcolor * dpg[k]; q[2] += dchannel_dcolor * dpb[k];
          last_alpha[k] = alpha[k];
          dL_dalpha = dL_dalpha * T[k] - (T_final[k] * inv) * bg_dot[k];
          const float h = (op * G[k]) * dL_dalpha;                      // G * dL/dG with dL/dG = opacity * dL/dalpha
          const float hx = h * dx, hy = h * dy[k];
          q[3] += h; q[4] += hx; q[5] += hy;
          q[6] += hx * dx; q[7] += hx * dy[k]; q8 += hy * dy[k];
        }
      }
      const float tot = reduce8(q, lane);
      q8 = wave_sum_to_lane63(q8);
      // eight totals sit in the lanes with (lane & 7) == 0, the ninth in lane 63: one atomic instruction commits all nine
      const bool last_lane = lane == 63;
      if (committer || last_lane)
        atomicAdd(grad_acc + (size_t)gid * GM_ACC_STRIDE + (last_lane ? 8 : my_slot), last_lane ? q8 : tot);
    }
    cur = nx1; nx1 = nx2; mine = mine_1; mine_1 = mine_2; id_2 = id_3; mine_2 = mine_3;
  }
}

int launch_render_bwd(const GeomState& g, const uint2* pairs, ImageState& img, int W, int H, int mode,
                      const float* background, const float* dL_dpix, int debug, hipStream_t s) {
  StageScope sc(ST_RENDER_BWD, s);
  const TileGrid tg(W, H, mode);
  const TileMap tm{tg.gx, tg.gy, tg.pgx, tg.pgy, tg.s};
  if (tg.ptiles > 0)
    hipLaunchKernelGGL(render_bwd_kernel, dim3(tm.blocks()), dim3(256), 0, s, img.ranges, pairs, g.splat, W, H, tm,
                       background, img.final_T, img.n_contrib, dL_dpix, g.grad_acc, g.counters, mode);
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

}  // namespace gm
