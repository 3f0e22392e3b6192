// gm_cull.h -- exact, conservative "can this Gaussian reach alpha >= 1/255 inside this pixel rectangle" test.
// Used at two granularities: 16x16 tiles when instances are emitted (gm_preprocess.hip / gm_binning.hip) and the
// 16x4 strip of a wave inside the blend kernels (gm_render.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace gm {

// Can entry (centre sx,sy; conic a,b,c; opacity op) reach alpha >= 1/255 at any pixel centre of the rectangle
// [x0,x1] x [y0,y1]?  alpha >= 1/255  <=>  q(d) = a dx^2 + 2 b dx dy + c dy^2 <= 2 ln(255 op), d = centre - pixel.
// q is a positive-definite form, so its minimum over the rectangle is 0 if the centre is inside and otherwise
// lies on one of the four edges (1-D clamped minimum per edge).  The margin covers float rounding of both this
// test and the per-pixel evaluation (proportional to the magnitude of the cancelling terms).
__device__ __forceinline__ bool may_touch(float sx, float sy, float a, float b, float c, float op,
                                          float x0, float x1, float y0, float y1) {
  if (!(op >= 0.0039f)) return false;                 // op < 1/255 (1/255 = 0.0039216): alpha = op*G < 1/255 everywhere
  // d = centre - pixel ranges over [dxl, dxh] x [dyl, dyh]; q is minimal at d = 0
  const float dxl = sx - x1, dxh = sx - x0, dyl = sy - y1, dyh = sy - y0;
  const float thr = 1.3862943611f * __builtin_amdgcn_logf(255.0f * op);   // 2 ln(255 op) = 2 ln2 log2(255 op)
  const float mx = fmaxf(fabsf(dxl), fabsf(dxh)), my = fmaxf(fabsf(dyl), fabsf(dyh));
  const float margin = 4e-6f * (a * mx * mx + c * my * my + 2.0f * fabsf(b) * mx * my) + 1e-3f;
  // Round 5: TWO edges instead of four.  The constrained minimum of a convex form whose free minimum (d = 0) lies outside the
  // rectangle is on an edge that faces it: dx = ex, the end of [dxl, dxh] nearest 0, and dy = ey likewise (a point of a far edge
  // could slide towards 0 inside the rectangle and lower q).  With 0 inside a range (ex or ey = 0) that candidate is a feasible
  // interior line whose clamped minimum is still a point of the rectangle - never below the true minimum - and with both 0 the
  // centre is inside and both candidates are q(0, 0) = 0.  24 vector instructions less per batch of 64 candidates in either blend.
  const float ex = __builtin_amdgcn_fmed3f(0.f, dxl, dxh), ey = __builtin_amdgcn_fmed3f(0.f, dyl, dyh);
  const float nb_c = -b * __builtin_amdgcn_rcpf(c), nb_a = -b * __builtin_amdgcn_rcpf(a);
  const float y = __builtin_amdgcn_fmed3f(nb_c * ex, dyl, dyh);
  const float x = __builtin_amdgcn_fmed3f(nb_a * ey, dxl, dxh);
  const float q1 = a * ex * ex + 2.f * b * ex * y + c * y * y;
  const float q2 = a * x * x + 2.f * b * x * ey + c * ey * ey;
  const float qmin = fminf(q1, q2);
  return !(qmin > thr + margin);                       // NaN-safe: keep the entry unless it is provably out of reach
}


// ---------------------------------------------------------------------------------------------
// Tile-row form of the same test, used when instances are emitted: instead of testing the tiles of the candidate
// rectangle one by one, compute per tile ROW the pixel-x interval in which the region {alpha >= 1/255} (the ellipse
// q(d) <= lim) exists inside that row's pixel band.  A tile of the row is hit iff its pixel range meets the
// interval (both sets are convex and the tile spans the whole band), so this yields exactly the tiles may_touch()
// would keep (up to the conservative epsilons) at a cost per row instead of per tile.
// With d = centre - pixel, for fixed dy the ellipse spans dx in (-b dy -+ sqrt(a lim - det dy^2)) / a; over a band the
// upper end is concave in dy (max at an end point or at dy = -dystar where it equals xext), the lower end convex.
// The three functions below run with FMA contraction off in every translation unit: preprocess (instance counts) and
// duplicate_kernel (instance emission) must make identical decisions.
struct TileCull {
  float a, b, inv_a, lim, det, dymax, xext, dystar;
  int mode;          // 0 = nothing reachable, 1 = use row_span, 2 = keep every tile (degenerate conic / NaN)
};

__device__ __forceinline__ TileCull tile_cull_setup(float sx, float sy, float a, float b, float c, float op,
                                                    float rx0, float rx1, float ry0, float ry1) {
#pragma clang fp contract(off)
  TileCull t;
  t.a = a; t.b = b; t.inv_a = __builtin_amdgcn_rcpf(a);
  t.mode = 1;
  const float mx = fmaxf(fabsf(sx - rx0), fabsf(sx - rx1)), my = fmaxf(fabsf(sy - ry0), fabsf(sy - ry1));
  const float thr = 1.3862943611f * __builtin_amdgcn_logf(255.0f * op);
  t.lim = thr + 4e-6f * (a * mx * mx + c * my * my + 2.0f * fabsf(b) * mx * my) + 1e-3f;
  t.det = a * c - b * b;
  if (!(op >= 0.0039f) || t.lim <= 0.f) { t.mode = 0; return t; }
  // hardware sqrt / rcp (1 ulp) instead of the correctly rounded sequences: this test only has to be conservative and
  // identical wherever it is evaluated; the 1e-4 relative / 1e-3 absolute margins dwarf the approximation error
  const float r = t.lim * __builtin_amdgcn_rcpf(t.det);
  t.dymax = __builtin_amdgcn_sqrtf(a * r) * 1.0001f + 1e-3f;
  t.xext = __builtin_amdgcn_sqrtf(c * r) * 1.0001f + 1e-3f;
  t.dystar = b * __builtin_amdgcn_sqrtf(r * __builtin_amdgcn_rcpf(c));
  if (!(t.det > 0.f) || !(t.dymax < 1e30f) || !(t.xext < 1e30f) || !(t.dystar == t.dystar)) t.mode = 2;
  return t;
}

// pixel-x interval [A, B] reachable inside pixel rows [py0, py1]; false if the band is out of reach
__device__ __forceinline__ bool row_span(const TileCull& t, float sx, float sy, float py0, float py1, float& A, float& B) {
#pragma clang fp contract(off)
  float dyl = fmaxf(sy - py1, -t.dymax), dyh = fminf(sy - py0, t.dymax);
  if (!(dyl <= dyh)) return false;
  const float sl = __builtin_amdgcn_sqrtf(fmaxf(t.a * t.lim - t.det * dyl * dyl, 0.f)), sh = __builtin_amdgcn_sqrtf(fmaxf(t.a * t.lim - t.det * dyh * dyh, 0.f));
  float hi = fmaxf(-t.b * dyl + sl, -t.b * dyh + sh) * t.inv_a;
  float lo = fminf(-t.b * dyl - sl, -t.b * dyh - sh) * t.inv_a;
  if (-t.dystar >= dyl && -t.dystar <= dyh) hi = t.xext;
  if (t.dystar >= dyl && t.dystar <= dyh) lo = -t.xext;
  hi = fminf(hi, t.xext); lo = fmaxf(lo, -t.xext);
  const float eps = 2e-3f + 1e-4f * (fabsf(hi) + fabsf(lo));
  A = sx - hi - eps; B = sx - lo + eps;
  return true;
}

// tiles [ta, tb] (inclusive, clamped to [x0, x1)) of tile row ty that are emitted; false if none
__device__ __forceinline__ bool row_tiles(const TileCull& t, float sx, float sy, int ty, int x0, int x1, int& ta, int& tb) {
#pragma clang fp contract(off)
  if (t.mode == 0) return false;
  ta = x0; tb = x1 - 1;
  if (t.mode == 2) return true;
  float A, B;
  const float py0 = (float)(ty * 16);
  if (!row_span(t, sx, sy, py0, py0 + 15.f, A, B)) return false;
  // tile tx covers pixel centres [16 tx, 16 tx + 15]
  const float fa = ceilf((A - 15.f) * 0.0625f), fb = floorf(B * 0.0625f);
  ta = max(x0, (int)fmaxf(fa, -1e9f)); tb = min(x1 - 1, (int)fminf(fb, 1e9f));
  return ta <= tb;
}

// row_tiles for the two tile rows ty, ty + 1 at once, straight-line (no branch, the two rows' arithmetic side by side so that
// the multiplies / adds pair up on the packed-f32 pipe): the per-row operations and their order are those of row_span /
// row_tiles, so the spans are bit-identical.  A row without tiles comes back as the empty interval (GM_ROW_EMPTY_LO, -1).
// Precondition: t.mode != 0.
#define GM_ROW_EMPTY_LO 0x3fffffff
typedef float cull_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cull_v2f v2_max(cull_v2f a, float b) { return cull_v2f{fmaxf(a.x, b), fmaxf(a.y, b)}; }
__device__ __forceinline__ cull_v2f v2_min(cull_v2f a, float b) { return cull_v2f{fminf(a.x, b), fminf(a.y, b)}; }
__device__ __forceinline__ cull_v2f v2_sqrt(cull_v2f a) { return cull_v2f{__builtin_amdgcn_sqrtf(a.x), __builtin_amdgcn_sqrtf(a.y)}; }
__device__ __forceinline__ void row_tiles_pair(const TileCull& t, float sx, float sy, int ty, int x0, int x1, int* ta, int* tb) {
#pragma clang fp contract(off)
  const float al = t.a * t.lim, nb = -t.b;
  const float p = (float)(ty * 16);
  const cull_v2f py0 = {p, p + 16.f};                       // (exact: multiples of 16 far below 2^24)
  const cull_v2f py1 = py0 + 15.f;
  const cull_v2f dyl = v2_max(sy - py1, -t.dymax), dyh = v2_min(sy - py0, t.dymax);
  const cull_v2f sl = v2_sqrt(v2_max(al - t.det * dyl * dyl, 0.f)), sh = v2_sqrt(v2_max(al - t.det * dyh * dyh, 0.f));
  const cull_v2f bl = nb * dyl, bh = nb * dyh;
  const cull_v2f hl = bl + sl, hh = bh + sh, ll = bl - sl, lh = bh - sh;
  cull_v2f hi = cull_v2f{fmaxf(hl.x, hh.x), fmaxf(hl.y, hh.y)} * t.inv_a;
  cull_v2f lo = cull_v2f{fminf(ll.x, lh.x), fminf(ll.y, lh.y)} * t.inv_a;
  // dyl <= x <= dyh as med3(dyl, x, dyh) == x (a row with dyl > dyh is dropped below whatever this gives)
  const float nd = -t.dystar;
  if (__builtin_amdgcn_fmed3f(dyl.x, nd, dyh.x) == nd) hi.x = t.xext;
  if (__builtin_amdgcn_fmed3f(dyl.y, nd, dyh.y) == nd) hi.y = t.xext;
  if (__builtin_amdgcn_fmed3f(dyl.x, t.dystar, dyh.x) == t.dystar) lo.x = -t.xext;
  if (__builtin_amdgcn_fmed3f(dyl.y, t.dystar, dyh.y) == t.dystar) lo.y = -t.xext;
  hi = v2_min(hi, t.xext); lo = v2_max(lo, -t.xext);
  const cull_v2f mag = {fabsf(hi.x) + fabsf(lo.x), fabsf(hi.y) + fabsf(lo.y)};
  const cull_v2f eps = 2e-3f + 1e-4f * mag;
  const cull_v2f A = sx - hi - eps, B = sx - lo + eps;
  const cull_v2f fa = (A - 15.f) * 0.0625f, fb = B * 0.0625f;
  const float fav[2] = {ceilf(fa.x), ceilf(fa.y)}, fbv[2] = {floorf(fb.x), floorf(fb.y)};
  const bool in[2] = {dyl.x <= dyh.x, dyl.y <= dyh.y};
#pragma unroll
  for (int r = 0; r < 2; r++) {
    int a = max(x0, (int)fmaxf(fav[r], -1e9f)), b = min(x1 - 1, (int)fminf(fbv[r], 1e9f));
    if (t.mode == 2) { a = x0; b = x1 - 1; }
    const bool ok = t.mode == 2 || (in[r] && a <= b);
    ta[r] = ok ? a : GM_ROW_EMPTY_LO; tb[r] = ok ? b : -1;
  }
}

// ---------------------------------------------------------------------------------------------
// Policy 2 (2 x 2-tile parents = 4 x 4 quadrants of 8 x 8 pixels, one blend wave per quadrant) records the reach of a Gaussian per
// QUADRANT when its rectangle (x0, y0, w, h in tiles) meets at most 2 x 2 parents - every rectangle of up to 3 x 3 tiles, i.e.
// nearly every Gaussian of an object seen from outside: the emission record's 64-bit mask then holds the four parents' 16-bit
// quadrant masks, parent (x0 / 2 + pi, y0 / 2 + pj) in bits [16 (2 pj + pi), +16), quadrant (qx, qy) of a parent in bit 4 qy + qx
// (gm_pre_body.h pre_emit builds them, duplicate_kernel<1> turns each non-empty one into an instance).
__device__ __forceinline__ bool quad_rect(uint32_t x0, uint32_t y0, uint32_t w, uint32_t h) {
  return ((x0 + w - 1u) >> 1) - (x0 >> 1) <= 1u && ((y0 + h - 1u) >> 1) - (y0 >> 1) <= 1u;
}
// the 2 x 2 quadrant block of every set tile bit of a 2 x 2-tile parent (tile bit cy * 2 + cx -> quadrant bits (2 cy + {0,1}) * 4 + 2 cx + {0,1})
__device__ __forceinline__ uint32_t tile_to_quad_mask(uint32_t cm) {
  return ((cm & 1u) ? 0x0033u : 0u) | ((cm & 2u) ? 0x00CCu : 0u) | ((cm & 4u) ? 0x3300u : 0u) | ((cm & 8u) ? 0xCC00u : 0u);
}

// ---------------------------------------------------------------------------------------------
// Parent tiles (2^s x 2^s tiles, s = 1 or 2).  `bits`: bit c = tile column x0 + c of one parent ROW is reached
// (union over the row's child rows; at most 60 columns).  Returns the number of parent columns with a reached child.
__device__ __forceinline__ unsigned long long fold_parent_bits(unsigned long long bits, int x0, int s) {
  const unsigned long long f = bits << (x0 & ((1 << s) - 1));      // bit position = column - (x0 rounded down to a parent edge)
  return s == 1 ? ((f | (f >> 1)) & 0x5555555555555555ull) : ((f | (f >> 1) | (f >> 2) | (f >> 3)) & 0x1111111111111111ull);
}
__device__ __forceinline__ uint32_t parents_in_row(unsigned long long bits, int x0, int s) {
  return (uint32_t)__popcll(fold_parent_bits(bits, x0, s));
}

}  // namespace gm
