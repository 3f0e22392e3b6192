// gm_pre_body.h -- the per-Gaussian forward geometry of preprocessCUDA (RAST/forward.cu:155-256) as inline device
// functions, shared by preprocess_fwd_kernel (gm_preprocess.hip) and the fused deform+shade+preprocess kernel of the edit
// loop (gm_deform.hip).  ARITHMETIC CONTRACT: every function body here switches FMA contraction off and keeps the
// reference's association order, whatever the including translation unit's default is, so both users (and the CPU
// oracle) produce bit-identical radii, rectangles, conics, depth keys and instance counts.
#pragma once
#include "gm_common.h"
#include "gm_cull.h"

namespace gm {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 xform4x3(const V3 p, const float* __restrict__ m) {
#pragma clang fp contract(off)
  V3 o;
  o.x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
  o.y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
  o.z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
  return o;
}

__device__ __forceinline__ float ndc2pix(float v, int S) {
#pragma clang fp contract(off)
  return (float)(((v + 1.0) * S - 1.0) * 0.5);
}

__device__ __forceinline__ void get_rect(float px, float py, int r, int gx, int gy, int& x0, int& y0, int& x1, int& y1) {
#pragma clang fp contract(off)
  x0 = min(gx, max(0, (int)((px - r) / GM_TILE)));
  y0 = min(gy, max(0, (int)((py - r) / GM_TILE)));
  x1 = min(gx, max(0, (int)((px + r + GM_TILE - 1) / GM_TILE)));
  y1 = min(gy, max(0, (int)((py + r + GM_TILE - 1) / GM_TILE)));
}

// GLM-argument-order rotation entries: Rg[3*c+r] = column c, row r of glm::mat3 R (forward.cu:134-138)
__device__ __forceinline__ void quat_cols(float r, float x, float y, float z, float* Rg) {
#pragma clang fp contract(off)
  Rg[0] = 1.f - 2.f * (y * y + z * z); Rg[1] = 2.f * (x * y - r * z); Rg[2] = 2.f * (x * z + r * y);
  Rg[3] = 2.f * (x * y + r * z); Rg[4] = 1.f - 2.f * (x * x + z * z); Rg[5] = 2.f * (y * z - r * x);
  Rg[6] = 2.f * (x * z - r * y); Rg[7] = 2.f * (y * z + r * x); Rg[8] = 1.f - 2.f * (x * x + y * y);
}

// T = W*J (two non-zero GLM columns), clamped t and clamp masks.  forward.cu:80-99 / backward.cu:166-192
__device__ __forceinline__ void cov2d_T(V3 mean, float fx, float fy, float tanx, float tany, const float* __restrict__ v,
                                        V3& t, float* T0, float* T1, float& xmul, float& ymul) {
#pragma clang fp contract(off)
  t = xform4x3(mean, v);
  const float limx = 1.3f * tanx, limy = 1.3f * tany;
  const float txtz = t.x / t.z, tytz = t.y / t.z;
  t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
  t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
  xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  const float j00 = fx / t.z, j02 = -(fx * t.x) / (t.z * t.z);
  const float j11 = fy / t.z, j12 = -(fy * t.y) / (t.z * t.z);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    T0[i] = (v[4 * i] * j00 + v[4 * i + 1] * 0.0f) + v[4 * i + 2] * j02;
    T1[i] = (v[4 * i] * 0.0f + v[4 * i + 1] * j11) + v[4 * i + 2] * j12;
  }
}

// cov = T^T Vrk^T T; entries [0][0], [0][1], [1][1]; no low-pass.  forward.cu:101-106
__device__ __forceinline__ void cov2d_from_T(const float* T0, const float* T1, const float* c, float& a, float& b, float& cc) {
#pragma clang fp contract(off)
  const float V[9] = {c[0], c[1], c[2], c[1], c[3], c[4], c[2], c[4], c[5]};
  float A0[3], A1[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    A0[k] = (T0[0] * V[0 + k] + T0[1] * V[3 + k]) + T0[2] * V[6 + k];
    A1[k] = (T1[0] * V[0 + k] + T1[1] * V[3 + k]) + T1[2] * V[6 + k];
  }
  a = (A0[0] * T0[0] + A0[1] * T0[1]) + A0[2] * T0[2];
  b = (A1[0] * T0[0] + A1[1] * T0[1]) + A1[2] * T0[2];
  cc = (A1[0] * T1[0] + A1[1] * T1[1]) + A1[2] * T1[2];
}


struct PreCam {                 // camera / raster constants of one forward (wave-uniform)
  int W, H, gx, gy, tile_cull;
  const float *view, *proj;
  float tanx, tany, fx, fy;
};

struct PreGeom {                // result of pre_project for a visible Gaussian
  float pix, piy, conx, cony, conz, depth, radius;
  int x0, y0, x1, y1;
};

// projection, EWA covariance, conic, radius, tile rectangle.  false = culled (reference early-outs: behind the near
// plane, det == 0, empty rectangle).
__device__ __forceinline__ bool pre_project(const PreCam& a, const V3 p, const float* c3, PreGeom& o) {
#pragma clang fp contract(off)
  const float* pm = a.proj;
  const float hx = pm[0] * p.x + pm[4] * p.y + pm[8] * p.z + pm[12];
  const float hy = pm[1] * p.x + pm[5] * p.y + pm[9] * p.z + pm[13];
  const float hw = pm[3] * p.x + pm[7] * p.y + pm[11] * p.z + pm[15];
  const float p_w = 1.0f / (hw + 0.0000001f);
  const float prx = hx * p_w, pry = hy * p_w;
  const V3 pv = xform4x3(p, a.view);
  if (pv.z <= 0.2f) return false;                   // in_frustum, auxiliary.h:153
  V3 t; float T0[3], T1[3], xm, ym, ca, cb, cc;
  cov2d_T(p, a.fx, a.fy, a.tanx, a.tany, a.view, t, T0, T1, xm, ym);
  cov2d_from_T(T0, T1, c3, ca, cb, cc);
  ca += 0.3f; cc += 0.3f;
  const float det = ca * cc - cb * cb;
  if (det == 0.0f) return false;
  const float det_inv = 1.f / det;
  o.conx = cc * det_inv; o.cony = -cb * det_inv; o.conz = ca * det_inv;
  const float mid = 0.5f * (ca + cc);
  const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
  const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
  o.radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
  o.pix = ndc2pix(prx, a.W); o.piy = ndc2pix(pry, a.H);
  get_rect(o.pix, o.piy, (int)o.radius, a.gx, a.gy, o.x0, o.y0, o.x1, o.y1);
  if ((o.x1 - o.x0) * (o.y1 - o.y0) == 0) return false;
  o.depth = pv.z;
  return true;
}

// Instances to emit.  The reference emits one per tile of the rectangle (rasterizer_impl.cu:98-109).  With
// tile_cull the tiles the Gaussian cannot reach with alpha >= 1/255 are dropped here: every pixel of such a
// tile skips the entry anyway (forward.cu:344), so images and gradients are unchanged while the instance
// count - and with it the sort, the tile lists and the blend work - shrinks by ~2-3x.
__device__ __forceinline__ void pre_emit(const PreCam& a, const PreGeom& g, float opac, uint32_t& tiles, uint4& bin) {
#pragma clang fp contract(off)
  const int x0 = g.x0, y0 = g.y0, x1 = g.x1, y1 = g.y1;
  const int rw = x1 - x0, rh = y1 - y0, ncand = rw * rh;
  unsigned long long mask = 0ull;
  if (!a.tile_cull) {
    tiles = (uint32_t)ncand;
    mask = ncand >= 64 ? ~0ull : ((1ull << ncand) - 1ull);
  } else {
    const TileCull tc = tile_cull_setup(g.pix, g.piy, g.conx, g.cony, g.conz, opac, (float)(x0 * GM_TILE), (float)(x1 * GM_TILE - 1),
                                        (float)(y0 * GM_TILE), (float)(y1 * GM_TILE - 1));
    // s > 0: instances are (Gaussian, parent tile) pairs, a parent = 2^s x 2^s tiles.  Small rectangles: the parents
    // with at least one reached child (exact, from the child mask); others: per parent row the hull of the child
    // rows' spans (duplicate_kernel applies the same two rules).
    const int s = a.tile_cull >= 2 ? a.tile_cull - 1 : 0;
    const bool small = ncand <= 64 && rw <= 60;
    uint32_t cnt = 0;
    if (s == 1) {
      // 2 x 2-tile parents (the default policy): one iteration per parent row, its two tile rows evaluated side by side
      // without a branch, and the two counting rules in interval arithmetic instead of 64-bit column masks:
      //   small rectangles  parents with a reached child = |PA u PB| = |PA| + |PB| - |PA n PB|  (PA, PB: the rows' spans in parent columns)
      //   others            the hull of the two spans
      // (an empty span is (GM_ROW_EMPTY_LO, -1): every length below comes out <= 0 and clamps to 0).
      // Rectangles inside 2 x 2 parents (quad_rect, gm_cull.h) record their reach per 8x8 QUADRANT instead of per tile - the blend
      // kernels run one wave per quadrant: the quadrants of the reached tiles that the bounding box of the alpha >= 1/255 ellipse
      // (tile_cull_setup's xext x dymax) touches.  A parent none of whose quadrants passes is not emitted.
      const bool quad = quad_rect((uint32_t)x0, (uint32_t)y0, (uint32_t)rw, (uint32_t)rh);
      // a wave whose Gaussians ALL have such a rectangle (the usual case: neighbours in memory are neighbours on the mesh and of
      // one size) takes count and mask from the quadrant words alone: the per-tile count and column mask of the general rule, which
      // every lane would otherwise compute and a quad lane then throws away, are skipped by a wave-uniform branch
      const bool all_quad = __all(quad);
      const int px0 = x0 >> 1, py0 = y0 >> 1;
      int bqx0 = 0, bqx1 = 7, bqy0 = 0, bqy1 = 7;                       // the box in quadrant columns / rows relative to parent (px0, py0)
      if (quad && tc.mode == 1) {
        const float ex = tc.xext + 2e-3f, ey = tc.dymax + 2e-3f;         // quadrant q covers pixel centres [8 q, 8 q + 7]
        bqx0 = (int)fmaxf(ceilf((g.pix - ex - 7.f) * 0.125f), -1e9f) - 4 * px0; bqx1 = (int)fminf(floorf((g.pix + ex) * 0.125f), 1e9f) - 4 * px0;
        bqy0 = (int)fmaxf(ceilf((g.piy - ey - 7.f) * 0.125f), -1e9f) - 4 * py0; bqy1 = (int)fminf(floorf((g.piy + ey) * 0.125f), 1e9f) - 4 * py0;
      }
      uint32_t qw[2] = {0u, 0u};                                        // qw[pj]: quadrant masks of parents (px0, py0 + pj) | (px0 + 1, py0 + pj) << 16
      const int pr_last = tc.mode != 0 ? (y1 - 1) >> 1 : (y0 >> 1) - 1;
      for (int pr = y0 >> 1; pr <= pr_last; pr++) {
        const int ty = 2 * pr;
        int ta[2], tb[2];
        row_tiles_pair(tc, g.pix, g.piy, ty, x0, x1, ta, tb);
        if (ty < y0) { ta[0] = GM_ROW_EMPTY_LO; tb[0] = -1; }            // rows of the parent outside the rectangle
        if (ty + 1 >= y1) { ta[1] = GM_ROW_EMPTY_LO; tb[1] = -1; }
        if (!all_quad) {                                                   // (wave-uniform: see above)
          const int la = ta[0] >> 1, ha = tb[0] >> 1, lb = ta[1] >> 1, hb = tb[1] >> 1;
          const int uni = max(ha - la + 1, 0) + max(hb - lb + 1, 0) - max(min(ha, hb) - max(la, lb) + 1, 0);
          const int hull = max((max(tb[0], tb[1]) >> 1) - (min(ta[0], ta[1]) >> 1) + 1, 0);
          cnt += (uint32_t)(small ? uni : hull);
#pragma unroll
          for (int r = 0; r < 2; r++) {                                    // child mask of a rectangle of <= 64 tiles
            const int len = max(tb[r] - ta[r] + 1, 0);
            const unsigned long long run = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
            mask |= run << (((ty + r - y0) * rw + (ta[r] - x0)) & 63);
          }
        }
        if (quad) {
          uint32_t word = 0;
#pragma unroll
          for (int r = 0; r < 2; r++) {
            const int qa = max(2 * ta[r] - 4 * px0, bqx0), qb = min(2 * tb[r] + 1 - 4 * px0, bqx1);     // quadrant columns 0..7 of the two parents
            const uint32_t r8 = qa <= qb ? ((1u << (qb - qa + 1)) - 1u) << qa : 0u;
            const uint32_t two = (r8 & 0xFu) | ((r8 >> 4) << 16);
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const int ql = 2 * r + h, qrow = 4 * (pr - py0) + ql;                                     // quadrant row in the parent / relative to parent row py0
              if (qrow >= bqy0 && qrow <= bqy1) word |= two << (4 * ql);
            }
          }
          if (pr == py0) qw[0] = word; else qw[1] = word;
        }
      }
      if (ncand > 64) mask = 0ull;
      if (quad) {
        mask = (unsigned long long)qw[0] | ((unsigned long long)qw[1] << 32);
        cnt = ((qw[0] & 0xFFFFu) ? 1u : 0u) + ((qw[0] >> 16) ? 1u : 0u) + ((qw[1] & 0xFFFFu) ? 1u : 0u) + ((qw[1] >> 16) ? 1u : 0u);
      }
    } else {
    int cur_pr = -1, hull_lo = 0x7fffffff, hull_hi = -1;           // hull empty while hull_hi < 0
    unsigned long long prow = 0ull;
    for (int ry = 0; ry < rh; ry++) {            // per tile row: the span of tiles the alpha >= 1/255 region reaches
      const int pr = (y0 + ry) >> s;
      if (s > 0 && pr != cur_pr) {
        cnt += small ? parents_in_row(prow, x0, s) : (hull_hi >= 0 ? (uint32_t)((hull_hi >> s) - (hull_lo >> s) + 1) : 0u);
        cur_pr = pr; prow = 0ull; hull_lo = 0x7fffffff; hull_hi = -1;
      }
      int ta, tb;
      if (!row_tiles(tc, g.pix, g.piy, y0 + ry, x0, x1, ta, tb)) continue;
      const int len = tb - ta + 1, bit0 = ry * rw + (ta - x0);
      const unsigned long long run = len >= 64 ? ~0ull : ((1ull << len) - 1ull);
      if (ncand <= 64) mask |= run << bit0;
      if (s == 0) cnt += (uint32_t)len;
      else if (small) prow |= run << (ta - x0);
      else { hull_lo = min(hull_lo, ta); hull_hi = max(hull_hi, tb); }
    }
    if (s > 0) cnt += small ? parents_in_row(prow, x0, s) : (hull_hi >= 0 ? (uint32_t)((hull_hi >> s) - (hull_lo >> s) + 1) : 0u);
    }
    tiles = cnt;
  }
  bin = bin_pack((uint32_t)x0, (uint32_t)y0, (uint32_t)rw, (uint32_t)rh, tiles, mask);
}

// Per-wave partials of the instance total, the visible count and the range of coarse bins in use -> one of GM_SLOTS atomic slots (GeomState::slots), and the wave's visible depth keys
// into the coarse histogram (GeomState::coarse; one atomic per distinct coarse bin of the wave, spread over GM_COARSE_COPIES
// copies by workgroup id so that no address sees more than a few thousand atomics per launch).  Every lane of the wave must
// call this (culled / out-of-range lanes with tiles = 0, dkey = 0xFFFFFFFF).  The ordering kernels (gm_bucket.hip) reduce
// the slots to num_rendered and turn the histogram into the depth-bucket mapping.
__device__ __forceinline__ void slot_accumulate(uint32_t* __restrict__ slots, uint32_t* __restrict__ coarse, uint32_t tiles, uint32_t dkey) {
  const bool vis = dkey != 0xFFFFFFFFu;
  const int lane = threadIdx.x & 63;
  uint32_t s = tiles;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += (uint32_t)__shfl_xor((int)s, d);
  const unsigned long long visb = __ballot(vis);
  const uint32_t c = (dkey >> GM_COARSE_SHIFT) & (GM_COARSE_BINS - 1);          // (keys are positive float bits: < 2^31)
  uint32_t ncmin = vis ? (GM_COARSE_BINS - 1u) - c : 0u, cmax = vis ? c : 0u;   // wave extremes of the coarse bin (as two maxima)
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    ncmin = max(ncmin, (uint32_t)__shfl_xor((int)ncmin, d));
    cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, d));
  }
  if (lane == 0 && visb) {
    uint32_t* slot = slots + GM_SLOT_STRIDE * ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (GM_SLOTS - 1));
    if (s) atomicAdd(slot, s);                                         // instances
    atomicAdd(slot + 1, (uint32_t)__popcll(visb));                     // visible Gaussians
    atomicMax(slot + 2, ncmin);                                        // 2047 - (first coarse bin in use)
    atomicMax(slot + 3, cmax);                                         // last coarse bin in use
  }
  const uint32_t copy = blockIdx.x & (GM_COARSE_COPIES - 1);
  unsigned long long todo = visb;
  while (todo) {                                                       // wave-uniform loop over the distinct coarse bins
    const int l = __ffsll(todo) - 1;
    const uint32_t cl = (uint32_t)__builtin_amdgcn_readlane((int)c, l);
    const unsigned long long same = __ballot(vis && c == cl);
    if (lane == l) atomicAdd(coarse + coarse_index(cl, copy), (uint32_t)__popcll(same));
    todo &= ~same;
  }
}

}  // namespace gm
