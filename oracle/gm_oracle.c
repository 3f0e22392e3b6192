/*
 * gm_oracle.c -- CPU restatement of the GaussianMesh hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (gaussianmesh_amd/) never links, imports or calls anything in oracle/.
 *
 * PARITY STATUS: "parity unpinned" for the rasterizer core (functions marked [core] below).
 * The reference's implementation of that core is CUDA (+cub, +Jittor JIT); it cannot be
 * imported (no jittor) nor built in this image without writing stand-in CUDA headers, which
 * the build rules forbid, and the reference ships no tests / golden vectors.
 * What IS pinned against reference code executed in the dev container (tests/golden/, made by
 * tests/golden/make_golden.py importing /root/reference python modules):
 *   - the SH polynomial (orc_sh_to_rgb)      vs  utils/sh_utils.py:eval_sh, edittool/sh_utils.py:eval_sh
 *   - barycentric weights (orc_bary_weights)  vs  edittool/general_utils.py:get_barycentric_coordinate
 *   - camera matrices (python side)           vs  utils/graphics_utils.py:getWorld2View2
 * The core is additionally cross-checked by an independent dense numpy restatement
 * (oracle/torch_dense.py: float64, gradients from autograd), by central finite differences of its own forward, and by structural
 * invariants (tests/test_oracle_*.py).
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference,
 * RAST = gaussian_renderer/diff_gaussian_rasterizater/cuda_rasterizer).
 *
 * ARITHMETIC CONTRACT (shared with the HIP kernels, see DESIGN.md "arithmetic contract"):
 * per-Gaussian geometry (orc_preprocess) is evaluated in IEEE-754 binary32 with NO fused
 * multiply-add contraction, sums of products associated left-to-right exactly as the C
 * expressions of the reference read, correctly rounded / and sqrt, and ndc2Pix in binary64.
 * Compile with -ffp-contract=off.  Under this contract radii, tile rectangles, depth keys,
 * the sorted instance list and the tile ranges are bit-identical between this file and the
 * HIP path; blended colours and gradients agree to rounding (exp implementation, FMA).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16 /* RAST/config.h:15-16 BLOCK_X, BLOCK_Y */

/* RAST/auxiliary.h:21-38 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* RAST/rasterizer_impl.cu:35-50 getHigherMsb */
uint32_t orc_higher_msb(uint32_t n) {
  uint32_t msb = sizeof(n) * 4;
  uint32_t step = msb;
  while (step > 1) {
    step /= 2;
    if (n >> msb) msb += step; else msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}

/* RAST/auxiliary.h:40-43 ndc2Pix: evaluated in double, rounded to float on return */
static inline float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

/* RAST/auxiliary.h:45-55 getRect.  max_radius is an int (float radius converted by the caller);
 * (p - r)/16 is float arithmetic truncated toward zero by the (int) cast. */
static inline void get_rect(float px, float py, int max_radius, int gx, int gy,
                            int* x0, int* y0, int* x1, int* y1) {
  int v;
  v = (int)((px - max_radius) / TILE); v = v > 0 ? v : 0; *x0 = gx < v ? gx : v;
  v = (int)((py - max_radius) / TILE); v = v > 0 ? v : 0; *y0 = gy < v ? gy : v;
  v = (int)((px + max_radius + TILE - 1) / TILE); v = v > 0 ? v : 0; *x1 = gx < v ? gx : v;
  v = (int)((py + max_radius + TILE - 1) / TILE); v = v > 0 ? v : 0; *y1 = gy < v ? gy : v;
}

/* RAST/auxiliary.h:57-76 */
static inline void xform4x3(const float* p, const float* m, float* o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void xform4x4(const float* p, const float* m, float* o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
  o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* Quaternion (r,x,y,z) -> the nine numbers the reference passes to glm::mat3(...) in argument
 * order (RAST/forward.cu:134-138); Rg[3*c+r] is GLM column c, row r. */
static inline void quat_cols(const float* q, float* Rg) {
  float r = q[0], x = q[1], y = q[2], z = q[3];
  Rg[0] = 1.f - 2.f * (y * y + z * z); Rg[1] = 2.f * (x * y - r * z); Rg[2] = 2.f * (x * z + r * y);
  Rg[3] = 2.f * (x * y + r * z); Rg[4] = 1.f - 2.f * (x * x + z * z); Rg[5] = 2.f * (y * z - r * x);
  Rg[6] = 2.f * (x * z - r * y); Rg[7] = 2.f * (y * z + r * x); Rg[8] = 1.f - 2.f * (x * x + y * y);
}

/* RAST/forward.cu:118-152 computeCov3D: M = S*R (GLM), Sigma = M^T M; quaternion NOT normalised.
 * With GLM's column-major product, M[c][k] = s_k * Rg[c][k] and
 * Sigma[a][b] = (M[a][0]M[b][0] + M[a][1]M[b][1]) + M[a][2]M[b][2]. */
static inline void cov3d_from_scale_rot(const float* scale, float mod, const float* q, float* cov) {
  float Rg[9], Mc[9];
  float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
  quat_cols(q, Rg);
  for (int c = 0; c < 3; c++)
    for (int k = 0; k < 3; k++) Mc[3 * c + k] = s[k] * Rg[3 * c + k];
#define SIG(a, b) (Mc[3 * a + 0] * Mc[3 * b + 0] + Mc[3 * a + 1] * Mc[3 * b + 1] + Mc[3 * a + 2] * Mc[3 * b + 2])
  cov[0] = SIG(0, 0); cov[1] = SIG(0, 1); cov[2] = SIG(0, 2);
  cov[3] = SIG(1, 1); cov[4] = SIG(1, 2); cov[5] = SIG(2, 2);
#undef SIG
}

/* Shared by forward (RAST/forward.cu:74-113) and backward (RAST/backward.cu:166-199):
 * T = W*J with the frustum clamp; returns the two non-zero GLM columns of T (T0, T1, 3 each),
 * the clamped t, and the clamp masks. */
static inline void cov2d_T(const float* mean, float fx, float fy, float tanx, float tany,
                           const float* v, float* t, float* T0, float* T1, float* xmul, float* ymul) {
  xform4x3(mean, v, t);
  const float limx = 1.3f * tanx, limy = 1.3f * tany;
  const float txtz = t[0] / t[2], tytz = t[1] / t[2];
  t[0] = fminf(limx, fmaxf(-limx, txtz)) * t[2];
  t[1] = fminf(limy, fmaxf(-limy, tytz)) * t[2];
  *xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  *ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  const float j00 = fx / t[2], j02 = -(fx * t[0]) / (t[2] * t[2]);
  const float j11 = fy / t[2], j12 = -(fy * t[1]) / (t[2] * t[2]);
  for (int i = 0; i < 3; i++) {
    /* T[0][i] = W[0][i]*J[0][0] + W[1][i]*J[0][1] + W[2][i]*J[0][2], W[k][i] = v[4i+k] */
    T0[i] = (v[4 * i] * j00 + v[4 * i + 1] * 0.0f) + v[4 * i + 2] * j02;
    T1[i] = (v[4 * i] * 0.0f + v[4 * i + 1] * j11) + v[4 * i + 2] * j12;
  }
}

/* cov = T^T * Vrk^T * T, entries [0][0],[0][1],[1][1] (GLM [col][row]); low-pass NOT added here. */
static inline void cov2d_from_T(const float* T0, const float* T1, const float* c, float* a_, float* b_, float* c_) {
  const float V[9] = {c[0], c[1], c[2], c[1], c[3], c[4], c[2], c[4], c[5]}; /* V[3*col+row], symmetric */
  float A0[3], A1[3]; /* A[k][i] = sum_m T[i][m] * V[m][k]; Ai[k] */
  for (int k = 0; k < 3; k++) {
    A0[k] = (T0[0] * V[0 + k] + T0[1] * V[3 + k]) + T0[2] * V[6 + k];
    A1[k] = (T1[0] * V[0 + k] + T1[1] * V[3 + k]) + T1[2] * V[6 + k];
  }
  *a_ = (A0[0] * T0[0] + A0[1] * T0[1]) + A0[2] * T0[2]; /* cov[0][0] */
  *b_ = (A1[0] * T0[0] + A1[1] * T0[1]) + A1[2] * T0[2]; /* cov[0][1]: row 1 of A, column 0 of T */
  *c_ = (A1[0] * T1[0] + A1[1] * T1[1]) + A1[2] * T1[2]; /* cov[1][1] */
}

/* RAST/forward.cu:20-71 computeColorFromSH (forward).  dir may be supplied pre-rotated
 * (edit tool, edittool/__init__.py:442-448) by the caller through `dir`. */
static inline void sh_eval(int deg, const float* sh /* [M][3] */, const float* dir, float* out) {
  float x = dir[0], y = dir[1], z = dir[2];
  for (int ch = 0; ch < 3; ch++) {
#define S(i) sh[3 * (i) + ch]
    float r = SH_C0 * S(0);
    if (deg > 0) {
      r = r - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
      if (deg > 1) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        r = r + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) + SH_C2[2] * (2.0f * zz - xx - yy) * S(6) +
            SH_C2[3] * xz * S(7) + SH_C2[4] * (xx - yy) * S(8);
        if (deg > 2) {
          r = r + SH_C3[0] * y * (3.0f * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10) +
              SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11) +
              SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12) +
              SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) + SH_C3[5] * z * (xx - yy) * S(14) +
              SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
        }
      }
    }
#undef S
    out[ch] = r;
  }
}

/* Exposed for the golden-vector test against the reference's python eval_sh:
 * out = SH(dir) (no +0.5, no clamp).  sh layout [N][M][3], dirs [N][3] already unit. */
void orc_sh_to_rgb(int N, int deg, int M, const float* shs, const float* dirs, float* out) {
  for (int i = 0; i < N; i++) sh_eval(deg, shs + (size_t)i * M * 3, dirs + 3 * i, out + 3 * i);
}

/* ---------------------------------------------------------------------------------------------
 * [core] RAST/forward.cu:155-256 preprocessCUDA, including in_frustum (RAST/auxiliary.h:138-163).
 * Nullable: scales/rots (when cov3D_precomp given), shs (when colors_precomp given).
 * Outputs for culled Gaussians: radii=0, tiles=0, everything else left untouched (callers zero).
 * clamped is uint8 [P][3]; cov3D [P][6] is written only when computed from scale/rot.
 */
void orc_preprocess(int P, int D, int M, const float* means, const float* scales, float mod,
                    const float* rots, const float* opac, const float* shs, const float* cov3D_precomp,
                    const float* colors_precomp, const float* view, const float* proj,
                    const float* campos, int W, int H, float tanx, float tany, int* radii,
                    float* xy, float* depths, float* cov3D, float* rgb, float* conic_op,
                    uint32_t* tiles, uint8_t* clamped) {
  const float fy = H / (2.0f * tany), fx = W / (2.0f * tanx); /* RAST/rasterizer_impl.cu:359-360 */
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; idx++) {
    radii[idx] = 0;
    tiles[idx] = 0;
    const float* p = means + 3 * (size_t)idx;
    float hom[4], pv[3];
    xform4x4(p, proj, hom);
    const float p_w = 1.0f / (hom[3] + 0.0000001f);
    const float prx = hom[0] * p_w, pry = hom[1] * p_w;
    xform4x3(p, view, pv);
    if (pv[2] <= 0.2f) continue;
    float covtmp[6];
    const float* c3;
    if (cov3D_precomp) c3 = cov3D_precomp + 6 * (size_t)idx;
    else {
      cov3d_from_scale_rot(scales + 3 * (size_t)idx, mod, rots + 4 * (size_t)idx, covtmp);
      if (cov3D) memcpy(cov3D + 6 * (size_t)idx, covtmp, sizeof covtmp);
      c3 = covtmp;
    }
    float t[3], T0[3], T1[3], xm, ym, a, b, c;
    cov2d_T(p, fx, fy, tanx, tany, view, t, T0, T1, &xm, &ym);
    cov2d_from_T(T0, T1, c3, &a, &b, &c);
    a += 0.3f; c += 0.3f;
    const float det = a * c - b * b;
    if (det == 0.0f) continue;
    const float det_inv = 1.f / det;
    const float conx = c * det_inv, cony = -b * det_inv, conz = a * det_inv;
    const float mid = 0.5f * (a + c);
    const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
    const float pix = ndc2pix(prx, W), piy = ndc2pix(pry, H);
    int x0, y0, x1, y1;
    get_rect(pix, piy, (int)my_radius, gx, gy, &x0, &y0, &x1, &y1);
    if ((x1 - x0) * (y1 - y0) == 0) continue;
    if (!colors_precomp) {
      float dir[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
      const float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
      dir[0] = dir[0] / len; dir[1] = dir[1] / len; dir[2] = dir[2] / len;
      float col[3];
      sh_eval(D, shs + (size_t)idx * M * 3, dir, col);
      for (int ch = 0; ch < 3; ch++) {
        col[ch] += 0.5f;
        clamped[3 * (size_t)idx + ch] = col[ch] < 0;
        rgb[3 * (size_t)idx + ch] = fmaxf(col[ch], 0.0f);
      }
    }
    depths[idx] = pv[2];
    radii[idx] = (int)my_radius;
    xy[2 * (size_t)idx] = pix; xy[2 * (size_t)idx + 1] = piy;
    conic_op[4 * (size_t)idx + 0] = conx; conic_op[4 * (size_t)idx + 1] = cony;
    conic_op[4 * (size_t)idx + 2] = conz; conic_op[4 * (size_t)idx + 3] = opac[idx];
    tiles[idx] = (uint32_t)((y1 - y0) * (x1 - x0));
  }
}

/* RAST/rasterizer_impl.cu:54-66 checkFrustum / markVisible */
void orc_mark_visible(int P, const float* means, const float* view, const float* proj, uint8_t* present) {
  (void)proj;
  for (int i = 0; i < P; i++) {
    float pv[3];
    xform4x3(means + 3 * (size_t)i, view, pv);
    present[i] = !(pv[2] <= 0.2f);
  }
}

/* ---------------------------------------------------------------------------------------------
 * [core] Binning: inclusive scan (RAST/rasterizer_impl.cu:407), duplicateWithKeys (:70-111),
 * stable radix sort on bits [0, 32+bit) (:478-483), identifyTileRanges (:116-138, ranges zeroed
 * first :485).  Returns num_rendered.  keys/vals must hold R entries (call once with keys==NULL
 * to get R).  ranges is uint32 [T][2].
 */
int64_t orc_bin(int P, const float* xy, const float* depths, const int* radii, const uint32_t* tiles,
                int W, int H, uint64_t* keys, uint32_t* vals, uint32_t* ranges) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  int64_t R = 0;
  for (int i = 0; i < P; i++) R += tiles[i];
  if (!keys) return R;
  uint64_t* k0 = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(R ? R : 1));
  uint32_t* v0 = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(R ? R : 1));
  int64_t off = 0;
  for (int idx = 0; idx < P; idx++) {
    if (radii[idx] > 0) {
      int x0, y0, x1, y1;
      get_rect(xy[2 * (size_t)idx], xy[2 * (size_t)idx + 1], radii[idx], gx, gy, &x0, &y0, &x1, &y1);
      uint32_t dbits;
      memcpy(&dbits, depths + idx, 4);
      for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
          uint64_t key = (uint64_t)(y * gx + x);
          key <<= 32;
          key |= dbits;
          k0[off] = key; v0[off] = (uint32_t)idx; off++;
        }
    }
  }
  /* stable LSD radix sort, 8-bit digits, restricted to the low 32+bit bits */
  const int endbit = 32 + (int)orc_higher_msb((uint32_t)(gx * gy));
  uint64_t* ka = k0; uint32_t* va = v0; uint64_t* kb = keys; uint32_t* vb = vals;
  for (int sh = 0; sh < endbit; sh += 8) {
    const int nb = (endbit - sh) < 8 ? (endbit - sh) : 8;
    const uint32_t mask = (1u << nb) - 1;
    int64_t cnt[257] = {0};
    for (int64_t i = 0; i < R; i++) cnt[((ka[i] >> sh) & mask) + 1]++;
    for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
    for (int64_t i = 0; i < R; i++) {
      int64_t dst = cnt[(ka[i] >> sh) & mask]++;
      kb[dst] = ka[i]; vb[dst] = va[i];
    }
    uint64_t* tk = ka; ka = kb; kb = tk;
    uint32_t* tv = va; va = vb; vb = tv;
  }
  if (ka != keys) { memcpy(keys, ka, sizeof(uint64_t) * (size_t)R); memcpy(vals, va, sizeof(uint32_t) * (size_t)R); }
  free(k0); free(v0);
  memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
  for (int64_t i = 0; i < R; i++) {
    uint32_t cur = (uint32_t)(keys[i] >> 32);
    if (i == 0) ranges[2 * cur] = 0;
    else {
      uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
      if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)i; ranges[2 * cur] = (uint32_t)i; }
    }
    if (i == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
  }
  return R;
}

/* ---------------------------------------------------------------------------------------------
 * [core] RAST/forward.cu:261-374 renderCUDA (forward).  out_color planar [3][H][W].
 * `contributor` counts every visited entry; a pixel that reaches T(1-a) < 1e-4 stops WITHOUT
 * applying that entry.  The 256-entry batching of the reference only affects when a pixel stops
 * looking (per-pixel `done`), not the result, so the list is walked directly.
 */
void orc_render_fwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* xy,
                    const float* feat, const float* conic_op, const float* bg, float* out_color,
                    float* final_T, uint32_t* n_contrib) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 4)
  for (int tile = 0; tile < gx * gy; tile++) {
    const int tx = tile % gx, ty = tile / gx;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    for (int ly = 0; ly < TILE; ly++)
      for (int lx = 0; lx < TILE; lx++) {
        const int px = tx * TILE + lx, py = ty * TILE + ly;
        if (px >= W || py >= H) continue;
        const float pfx = (float)px, pfy = (float)py;
        float T = 1.0f, C[3] = {0, 0, 0};
        uint32_t contributor = 0, last = 0;
        for (uint32_t e = r0; e < r1; e++) {
          contributor++;
          const uint32_t g = point_list[e];
          const float dx = xy[2 * (size_t)g] - pfx, dy = xy[2 * (size_t)g + 1] - pfy;
          const float* co = conic_op + 4 * (size_t)g;
          const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0.0f) continue;
          const float alpha = fminf(0.99f, co[3] * expf(power));
          if (alpha < 1.0f / 255.0f) continue;
          const float test_T = T * (1 - alpha);
          if (test_T < 0.0001f) break;
          for (int ch = 0; ch < 3; ch++) C[ch] += feat[3 * (size_t)g + ch] * alpha * T;
          T = test_T;
          last = contributor;
        }
        const size_t pid = (size_t)W * py + px;
        final_T[pid] = T;
        n_contrib[pid] = last;
        for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pid] = C[ch] + T * bg[ch];
      }
  }
}

/* ---------------------------------------------------------------------------------------------
 * For every (Gaussian, tile) instance of a binned list: does ANY pixel of the tile accept the entry, i.e. pass both
 * skips of RAST/forward.cu:336-345 (power <= 0 and alpha >= 1/255) with the float arithmetic of orc_render_fwd?
 * Checker for the product's emission-time tile culling: every instance with needed == 1 must be emitted.
 * tile_of[i] = tile id of instance i (key >> 32), needed uint8 [R].
 */
void orc_instance_needed(int64_t R, int W, int H, const uint32_t* tile_of, const uint32_t* point_list, const float* xy,
                         const float* conic_op, uint8_t* needed) {
  const int gx = (W + TILE - 1) / TILE;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < R; i++) {
    const uint32_t g = point_list[i], t = tile_of[i];
    const int tx = (int)(t % (uint32_t)gx), ty = (int)(t / (uint32_t)gx);
    const float* co = conic_op + 4 * (size_t)g;
    uint8_t hit = 0;
    for (int ly = 0; ly < TILE && !hit; ly++)
      for (int lx = 0; lx < TILE; lx++) {
        const int px = tx * TILE + lx, py = ty * TILE + ly;
        if (px >= W || py >= H) continue;
        const float dx = xy[2 * (size_t)g] - (float)px, dy = xy[2 * (size_t)g + 1] - (float)py;
        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.0f) continue;
        if (fminf(0.99f, co[3] * expf(power)) >= 1.0f / 255.0f) { hit = 1; break; }
      }
    needed[i] = hit;
  }
}

/* ---------------------------------------------------------------------------------------------
 * [core] RAST/backward.cu:399-557 renderCUDA (backward).  Accumulators must be zero on entry:
 * dL_dmean2D [P][3] (x,y used), dL_dconic [P][4] (slots x,y,w used), dL_dopacity [P], dL_dcolor [P][3].
 * Accumulates in double (tile-parallel, atomic adds); the reference uses float atomics in
 * nondeterministic order.
 */
void orc_render_bwd(int P, int W, int H, const uint32_t* ranges, const uint32_t* point_list,
                    const float* bg, const float* xy, const float* conic_op, const float* colors,
                    const float* final_T, const uint32_t* n_contrib, const float* dL_dpix,
                    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  double* acc = (double*)calloc((size_t)P * 9, sizeof(double));
  const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
  /* tiles in parallel; the double accumulators are shared, so every update is an `omp atomic` (summation order is
   * then arbitrary, which perturbs a double sum of float terms at the 1e-16 level) */
#pragma omp parallel for schedule(dynamic, 4)
  for (int tile = 0; tile < gx * gy; tile++) {
    const int tx = tile % gx, ty = tile / gx;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    for (int ly = 0; ly < TILE; ly++)
      for (int lx = 0; lx < TILE; lx++) {
        const int px = tx * TILE + lx, py = ty * TILE + ly;
        if (px >= W || py >= H) continue;
        const size_t pid = (size_t)W * py + px;
        const float pfx = (float)px, pfy = (float)py;
        const float T_final = final_T[pid];
        float T = T_final;
        const uint32_t last_contributor = n_contrib[pid];
        float accum_rec[3] = {0, 0, 0}, dpx[3], last_color[3] = {0, 0, 0}, last_alpha = 0;
        for (int ch = 0; ch < 3; ch++) dpx[ch] = dL_dpix[(size_t)ch * H * W + pid];
        uint32_t contributor = r1 - r0;
        for (uint32_t e = r1; e-- > r0;) { /* back to front */
          contributor--;
          if (contributor >= last_contributor) continue;
          const uint32_t g = point_list[e];
          const float dx = xy[2 * (size_t)g] - pfx, dy = xy[2 * (size_t)g + 1] - pfy;
          const float* co = conic_op + 4 * (size_t)g;
          const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0.0f) continue;
          const float G = expf(power);
          const float alpha = fminf(0.99f, co[3] * G);
          if (alpha < 1.0f / 255.0f) continue;
          T = T / (1.f - alpha);
          const float dchannel_dcolor = alpha * T;
          float dL_dalpha = 0.0f;
          for (int ch = 0; ch < 3; ch++) {
            const float c = colors[3 * (size_t)g + ch];
            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
            last_color[ch] = c;
            dL_dalpha += (c - accum_rec[ch]) * dpx[ch];
            {
              const double add = dchannel_dcolor * dpx[ch];
#pragma omp atomic
              acc[9 * (size_t)g + ch] += add;
            }
          }
          dL_dalpha *= T;
          last_alpha = alpha;
          float bg_dot = 0;
          for (int i = 0; i < 3; i++) bg_dot += bg[i] * dpx[i];
          dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
          const float dL_dG = co[3] * dL_dalpha;
          const float gdx = G * dx, gdy = G * dy;
          const float dG_ddelx = -gdx * co[0] - gdy * co[1];
          const float dG_ddely = -gdy * co[2] - gdx * co[1];
          const double add6[6] = {dL_dG * dG_ddelx * ddelx_dx, dL_dG * dG_ddely * ddely_dy, -0.5f * gdx * dx * dL_dG,
                                  -0.5f * gdx * dy * dL_dG, -0.5f * gdy * dy * dL_dG, G * dL_dalpha};
          for (int k = 0; k < 6; k++) {
#pragma omp atomic
            acc[9 * (size_t)g + 3 + k] += add6[k];
          }
        }
      }
  }
  for (int g = 0; g < P; g++) {
    for (int ch = 0; ch < 3; ch++) dL_dcolor[3 * (size_t)g + ch] += (float)acc[9 * (size_t)g + ch];
    dL_dmean2D[3 * (size_t)g + 0] += (float)acc[9 * (size_t)g + 3];
    dL_dmean2D[3 * (size_t)g + 1] += (float)acc[9 * (size_t)g + 4];
    dL_dconic[4 * (size_t)g + 0] += (float)acc[9 * (size_t)g + 5];
    dL_dconic[4 * (size_t)g + 1] += (float)acc[9 * (size_t)g + 6];
    dL_dconic[4 * (size_t)g + 3] += (float)acc[9 * (size_t)g + 7];
    dL_dopacity[g] += (float)acc[9 * (size_t)g + 8];
  }
  free(acc);
}

/* RAST/auxiliary.h:106-116 dnormvdv(float3) */
static inline void dnormvdv3(const float* v, const float* dv, float* o) {
  const float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  o[0] = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
  o[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
  o[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

/* ---------------------------------------------------------------------------------------------
 * [core] RAST/backward.cu:144-274 computeCov2DCUDA followed by :346-396 preprocessCUDA (backward),
 * including computeColorFromSH bwd (:20-139) and computeCov3D bwd (:278-341).
 * cov3D = the [P][6] covariance used in forward (precomputed or the one written by preprocess).
 * dL_dmean3D is ASSIGNED for visible Gaussians (backward.cu:273) then accumulated into.
 * shs==NULL skips the SH branch; scales==NULL skips the scale/rot branch (backward.cu:390,394).
 */
void orc_preprocess_bwd(int P, int D, int M, const float* means, const int* radii, const float* shs,
                        const uint8_t* clamped, const float* scales, const float* rots, float mod,
                        const float* cov3D, const float* view, const float* proj, const float* campos,
                        int W, int H, float tanx, float tany, const float* dL_dmean2D,
                        const float* dL_dconic, float* dL_dmean3D, const float* dL_dcolor,
                        float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot) {
  const float fy = H / (2.0f * tany), fx = W / (2.0f * tanx);
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; idx++) {
    if (!(radii[idx] > 0)) continue;
    const float* mean = means + 3 * (size_t)idx;
    const float* c3 = cov3D + 6 * (size_t)idx;
    /* ---- computeCov2DCUDA ---- */
    {
      const float dcx = dL_dconic[4 * (size_t)idx], dcy = dL_dconic[4 * (size_t)idx + 1],
                  dcz = dL_dconic[4 * (size_t)idx + 3];
      float t[3], T0[3], T1[3], xm, ym, a, b, c;
      cov2d_T(mean, fx, fy, tanx, tany, view, t, T0, T1, &xm, &ym);
      cov2d_from_T(T0, T1, c3, &a, &b, &c);
      a += 0.3f; c += 0.3f;
      const float denom = a * c - b * b;
      float dL_da = 0, dL_db = 0, dL_dc = 0;
      const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
      float* dcov = dL_dcov3D + 6 * (size_t)idx;
      if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * dcx + 2 * b * c * dcy + (denom - a * c) * dcz);
        dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * c) * dcx);
        dL_db = denom2inv * 2 * (b * c * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
        /* T[0][k] = T0[k], T[1][k] = T1[k] (GLM [col][row]) */
        dcov[0] = (T0[0] * T0[0] * dL_da + T0[0] * T1[0] * dL_db + T1[0] * T1[0] * dL_dc);
        dcov[3] = (T0[1] * T0[1] * dL_da + T0[1] * T1[1] * dL_db + T1[1] * T1[1] * dL_dc);
        dcov[5] = (T0[2] * T0[2] * dL_da + T0[2] * T1[2] * dL_db + T1[2] * T1[2] * dL_dc);
        dcov[1] = 2 * T0[0] * T0[1] * dL_da + (T0[0] * T1[1] + T0[1] * T1[0]) * dL_db + 2 * T1[0] * T1[1] * dL_dc;
        dcov[2] = 2 * T0[0] * T0[2] * dL_da + (T0[0] * T1[2] + T0[2] * T1[0]) * dL_db + 2 * T1[0] * T1[2] * dL_dc;
        dcov[4] = 2 * T0[2] * T0[1] * dL_da + (T0[1] * T1[2] + T0[2] * T1[1]) * dL_db + 2 * T1[1] * T1[2] * dL_dc;
      } else {
        for (int i = 0; i < 6; i++) dcov[i] = 0;
      }
      /* Vrk[col][row], symmetric */
      const float V[9] = {c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]};
#define VR(cc, rr) V[3 * (cc) + (rr)]
      const float dT00 = 2 * (T0[0] * VR(0, 0) + T0[1] * VR(0, 1) + T0[2] * VR(0, 2)) * dL_da +
                         (T1[0] * VR(0, 0) + T1[1] * VR(0, 1) + T1[2] * VR(0, 2)) * dL_db;
      const float dT01 = 2 * (T0[0] * VR(1, 0) + T0[1] * VR(1, 1) + T0[2] * VR(1, 2)) * dL_da +
                         (T1[0] * VR(1, 0) + T1[1] * VR(1, 1) + T1[2] * VR(1, 2)) * dL_db;
      const float dT02 = 2 * (T0[0] * VR(2, 0) + T0[1] * VR(2, 1) + T0[2] * VR(2, 2)) * dL_da +
                         (T1[0] * VR(2, 0) + T1[1] * VR(2, 1) + T1[2] * VR(2, 2)) * dL_db;
      const float dT10 = 2 * (T1[0] * VR(0, 0) + T1[1] * VR(0, 1) + T1[2] * VR(0, 2)) * dL_dc +
                         (T0[0] * VR(0, 0) + T0[1] * VR(0, 1) + T0[2] * VR(0, 2)) * dL_db;
      const float dT11 = 2 * (T1[0] * VR(1, 0) + T1[1] * VR(1, 1) + T1[2] * VR(1, 2)) * dL_dc +
                         (T0[0] * VR(1, 0) + T0[1] * VR(1, 1) + T0[2] * VR(1, 2)) * dL_db;
      const float dT12 = 2 * (T1[0] * VR(2, 0) + T1[1] * VR(2, 1) + T1[2] * VR(2, 2)) * dL_dc +
                         (T0[0] * VR(2, 0) + T0[1] * VR(2, 1) + T0[2] * VR(2, 2)) * dL_db;
#undef VR
      /* W[col][row]: W[0]=(v0,v4,v8) W[1]=(v1,v5,v9) W[2]=(v2,v6,v10) */
      const float* v = view;
      const float dJ00 = v[0] * dT00 + v[4] * dT01 + v[8] * dT02;
      const float dJ02 = v[2] * dT00 + v[6] * dT01 + v[10] * dT02;
      const float dJ11 = v[1] * dT10 + v[5] * dT11 + v[9] * dT12;
      const float dJ12 = v[2] * dT10 + v[6] * dT11 + v[10] * dT12;
      const float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
      const float dtx = xm * -fx * tz2 * dJ02;
      const float dty = ym * -fy * tz2 * dJ12;
      const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * t[0]) * tz3 * dJ02 + (2 * fy * t[1]) * tz3 * dJ12;
      /* transformVec4x3Transpose, RAST/auxiliary.h:88-96 */
      float* dm = dL_dmean3D + 3 * (size_t)idx;
      dm[0] = v[0] * dtx + v[1] * dty + v[2] * dtz;
      dm[1] = v[4] * dtx + v[5] * dty + v[6] * dtz;
      dm[2] = v[8] * dtx + v[9] * dty + v[10] * dtz;
    }
    /* ---- preprocessCUDA (backward) ---- */
    {
      float hom[4];
      xform4x4(mean, proj, hom);
      const float m_w = 1.0f / (hom[3] + 0.0000001f);
      const float mul1 = (proj[0] * mean[0] + proj[4] * mean[1] + proj[8] * mean[2] + proj[12]) * m_w * m_w;
      const float mul2 = (proj[1] * mean[0] + proj[5] * mean[1] + proj[9] * mean[2] + proj[13]) * m_w * m_w;
      const float gx_ = dL_dmean2D[3 * (size_t)idx], gy_ = dL_dmean2D[3 * (size_t)idx + 1];
      float* dm = dL_dmean3D + 3 * (size_t)idx;
      dm[0] += (proj[0] * m_w - proj[3] * mul1) * gx_ + (proj[1] * m_w - proj[3] * mul2) * gy_;
      dm[1] += (proj[4] * m_w - proj[7] * mul1) * gx_ + (proj[5] * m_w - proj[7] * mul2) * gy_;
      dm[2] += (proj[8] * m_w - proj[11] * mul1) * gx_ + (proj[9] * m_w - proj[11] * mul2) * gy_;
    }
    if (shs) { /* computeColorFromSH backward, RAST/backward.cu:20-139 */
      const float dir_orig[3] = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
      const float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
      const float x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
      const float* sh = shs + (size_t)idx * M * 3;
      float* dsh = dL_dsh + (size_t)idx * M * 3;
      float dRGB[3], dx[3] = {0, 0, 0}, dy[3] = {0, 0, 0}, dz[3] = {0, 0, 0};
      for (int ch = 0; ch < 3; ch++)
        dRGB[ch] = dL_dcolor[3 * (size_t)idx + ch] * (clamped[3 * (size_t)idx + ch] ? 0.f : 1.f);
#define S(i, ch) sh[3 * (i) + (ch)]
#define DSH(i, w) for (int ch = 0; ch < 3; ch++) dsh[3 * (i) + ch] = (w) * dRGB[ch]
      DSH(0, SH_C0);
      if (D > 0) {
        DSH(1, -SH_C1 * y); DSH(2, SH_C1 * z); DSH(3, -SH_C1 * x);
        for (int ch = 0; ch < 3; ch++) { dx[ch] = -SH_C1 * S(3, ch); dy[ch] = -SH_C1 * S(1, ch); dz[ch] = SH_C1 * S(2, ch); }
        if (D > 1) {
          const float xx = x * x, yy = y * y, zz = z * z, xy_ = x * y, yz = y * z, xz = x * z;
          DSH(4, SH_C2[0] * xy_); DSH(5, SH_C2[1] * yz); DSH(6, SH_C2[2] * (2.f * zz - xx - yy));
          DSH(7, SH_C2[3] * xz); DSH(8, SH_C2[4] * (xx - yy));
          for (int ch = 0; ch < 3; ch++) {
            dx[ch] += SH_C2[0] * y * S(4, ch) + SH_C2[2] * 2.f * -x * S(6, ch) + SH_C2[3] * z * S(7, ch) + SH_C2[4] * 2.f * x * S(8, ch);
            dy[ch] += SH_C2[0] * x * S(4, ch) + SH_C2[1] * z * S(5, ch) + SH_C2[2] * 2.f * -y * S(6, ch) + SH_C2[4] * 2.f * -y * S(8, ch);
            dz[ch] += SH_C2[1] * y * S(5, ch) + SH_C2[2] * 2.f * 2.f * z * S(6, ch) + SH_C2[3] * x * S(7, ch);
          }
          if (D > 2) {
            DSH(9, SH_C3[0] * y * (3.f * xx - yy)); DSH(10, SH_C3[1] * xy_ * z);
            DSH(11, SH_C3[2] * y * (4.f * zz - xx - yy)); DSH(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
            DSH(13, SH_C3[4] * x * (4.f * zz - xx - yy)); DSH(14, SH_C3[5] * z * (xx - yy));
            DSH(15, SH_C3[6] * x * (xx - 3.f * yy));
            for (int ch = 0; ch < 3; ch++) {
              dx[ch] += (SH_C3[0] * S(9, ch) * 3.f * 2.f * xy_ + SH_C3[1] * S(10, ch) * yz + SH_C3[2] * S(11, ch) * -2.f * xy_ +
                         SH_C3[3] * S(12, ch) * -3.f * 2.f * xz + SH_C3[4] * S(13, ch) * (-3.f * xx + 4.f * zz - yy) +
                         SH_C3[5] * S(14, ch) * 2.f * xz + SH_C3[6] * S(15, ch) * 3.f * (xx - yy));
              dy[ch] += (SH_C3[0] * S(9, ch) * 3.f * (xx - yy) + SH_C3[1] * S(10, ch) * xz +
                         SH_C3[2] * S(11, ch) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * S(12, ch) * -3.f * 2.f * yz +
                         SH_C3[4] * S(13, ch) * -2.f * xy_ + SH_C3[5] * S(14, ch) * -2.f * yz + SH_C3[6] * S(15, ch) * -3.f * 2.f * xy_);
              dz[ch] += (SH_C3[1] * S(10, ch) * xy_ + SH_C3[2] * S(11, ch) * 4.f * 2.f * yz +
                         SH_C3[3] * S(12, ch) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * S(13, ch) * 4.f * 2.f * xz +
                         SH_C3[5] * S(14, ch) * (xx - yy));
            }
          }
        }
      }
#undef S
#undef DSH
      const float ddir[3] = {dx[0] * dRGB[0] + dx[1] * dRGB[1] + dx[2] * dRGB[2],
                             dy[0] * dRGB[0] + dy[1] * dRGB[1] + dy[2] * dRGB[2],
                             dz[0] * dRGB[0] + dz[1] * dRGB[1] + dz[2] * dRGB[2]};
      float dmean[3];
      dnormvdv3(dir_orig, ddir, dmean);
      float* dm = dL_dmean3D + 3 * (size_t)idx;
      dm[0] += dmean[0]; dm[1] += dmean[1]; dm[2] += dmean[2];
    }
    if (scales) { /* computeCov3D backward, RAST/backward.cu:278-341 */
      const float* q = rots + 4 * (size_t)idx;
      const float r = q[0], x = q[1], y = q[2], z = q[3];
      float Rg[9], Mc[9];
      quat_cols(q, Rg);
      const float s[3] = {mod * scales[3 * (size_t)idx], mod * scales[3 * (size_t)idx + 1], mod * scales[3 * (size_t)idx + 2]};
      for (int c = 0; c < 3; c++)
        for (int k = 0; k < 3; k++) Mc[3 * c + k] = s[k] * Rg[3 * c + k];
      const float* d = dL_dcov3D + 6 * (size_t)idx;
      /* dL_dSigma GLM cols (symmetric) */
      const float dS[9] = {d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2], 0.5f * d[4], d[5]};
      /* dL_dM = 2.0f * M * dL_dSigma : (2M)[c][r] then product: out[j][i] = sum_k (2M)[k][i] * dS[j][k] */
      float dM[9];
      for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++)
          dM[3 * j + i] = (2.0f * Mc[0 + i]) * dS[3 * j + 0] + (2.0f * Mc[3 + i]) * dS[3 * j + 1] + (2.0f * Mc[6 + i]) * dS[3 * j + 2];
      /* Rt = transpose(R): Rt[c][r] = Rg[r][c];  dMt[c][r] = dM[r][c] */
      float Rt[9], dMt[9];
      for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) { Rt[3 * c + rr] = Rg[3 * rr + c]; dMt[3 * c + rr] = dM[3 * rr + c]; }
      float* dsc = dL_dscale + 3 * (size_t)idx;
      for (int c = 0; c < 3; c++)
        dsc[c] = Rt[3 * c + 0] * dMt[3 * c + 0] + Rt[3 * c + 1] * dMt[3 * c + 1] + Rt[3 * c + 2] * dMt[3 * c + 2];
      for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) dMt[3 * c + rr] *= s[c];
#define DM(c, rr) dMt[3 * (c) + (rr)]
      float* dq = dL_drot + 4 * (size_t)idx;
      dq[0] = 2 * z * (DM(0, 1) - DM(1, 0)) + 2 * y * (DM(2, 0) - DM(0, 2)) + 2 * x * (DM(1, 2) - DM(2, 1));
      dq[1] = 2 * y * (DM(1, 0) + DM(0, 1)) + 2 * z * (DM(2, 0) + DM(0, 2)) + 2 * r * (DM(1, 2) - DM(2, 1)) - 4 * x * (DM(2, 2) + DM(1, 1));
      dq[2] = 2 * x * (DM(1, 0) + DM(0, 1)) + 2 * r * (DM(2, 0) - DM(0, 2)) + 2 * z * (DM(1, 2) + DM(2, 1)) - 4 * y * (DM(2, 2) + DM(0, 0));
      dq[3] = 2 * r * (DM(0, 1) - DM(1, 0)) + 2 * x * (DM(2, 0) + DM(0, 2)) + 2 * y * (DM(1, 2) + DM(2, 1)) - 4 * z * (DM(1, 1) + DM(0, 0));
#undef DM
    }
  }
}

/* ---------------------------------------------------------------------------------------------
 * Whole forward, OpenMP, used as the CPU baseline ("port") and by tests for the composed path.
 * Per-tile bucketing + per-tile sort on the unique composite (depth_bits<<32 | gaussian idx)
 * produces exactly the list of the stable radix sort in orc_bin.  Returns R; buffers are
 * internal.  out_final_T / out_ncontrib may be NULL.
 */
static int cmp_u64(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return (x > y) - (x < y);
}
int64_t orc_forward(int P, int D, int M, const float* bg, int W, int H, const float* means,
                    const float* shs, const float* colors_precomp, const float* opac, const float* scales,
                    float mod, const float* rots, const float* cov3D_precomp, const float* view,
                    const float* proj, const float* campos, float tanx, float tany, float* out_color,
                    int* radii, float* out_final_T, uint32_t* out_ncontrib) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, NT = gx * gy;
  float* xy = (float*)calloc((size_t)P * 2, 4);
  float* depths = (float*)calloc(P, 4);
  float* rgb = (float*)calloc((size_t)P * 3, 4);
  float* conic = (float*)calloc((size_t)P * 4, 4);
  uint32_t* tiles = (uint32_t*)calloc(P, 4);
  uint8_t* clamped = (uint8_t*)calloc((size_t)P * 3, 1);
  orc_preprocess(P, D, M, means, scales, mod, rots, opac, shs, cov3D_precomp, colors_precomp, view, proj,
                 campos, W, H, tanx, tany, radii, xy, depths, NULL, rgb, conic, tiles, clamped);
  /* count per tile: every thread takes a contiguous slice of the Gaussians and counts into its own row; the rows then become
   * each thread's first slot per tile, so the fill below needs no atomics.  Slots inside a tile are in (thread, id) order here;
   * the per-tile sort on the full (depth bits, id) key makes the list independent of it. */
  int64_t* tcount = (int64_t*)calloc((size_t)NT + 1, sizeof(int64_t));
  int nth = 1;
#ifdef _OPENMP
  nth = omp_get_max_threads();
#endif
  int64_t* local = (int64_t*)calloc((size_t)nth * NT, sizeof(int64_t));
#pragma omp parallel num_threads(nth)
  {
    int th = 0;
#ifdef _OPENMP
    th = omp_get_thread_num();
#endif
    const int i0 = (int)((int64_t)P * th / nth), i1 = (int)((int64_t)P * (th + 1) / nth);
    int64_t* row = local + (size_t)th * NT;
    for (int i = i0; i < i1; i++)
      if (radii[i] > 0) {
        int x0, y0, x1, y1;
        get_rect(xy[2 * (size_t)i], xy[2 * (size_t)i + 1], radii[i], gx, gy, &x0, &y0, &x1, &y1);
        for (int y = y0; y < y1; y++)
          for (int x = x0; x < x1; x++) row[y * gx + x]++;
      }
  }
  for (int t = 0; t < NT; t++) {               /* exclusive offsets: tile-major, thread-minor */
    int64_t run = tcount[t];
    for (int th = 0; th < nth; th++) { const int64_t c = local[(size_t)th * NT + t]; local[(size_t)th * NT + t] = run; run += c; }
    tcount[t + 1] = run;
  }
  const int64_t R = tcount[NT];
  uint64_t* comp = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(R ? R : 1));
  int64_t* cur = local;                         /* (freed below under the name the rest of the function uses) */
#pragma omp parallel num_threads(nth)
  {
    int th = 0;
#ifdef _OPENMP
    th = omp_get_thread_num();
#endif
    const int i0 = (int)((int64_t)P * th / nth), i1 = (int)((int64_t)P * (th + 1) / nth);
    int64_t* row = local + (size_t)th * NT;
    for (int i = i0; i < i1; i++)
      if (radii[i] > 0) {
        int x0, y0, x1, y1;
        get_rect(xy[2 * (size_t)i], xy[2 * (size_t)i + 1], radii[i], gx, gy, &x0, &y0, &x1, &y1);
        uint32_t db;
        memcpy(&db, depths + i, 4);
        for (int y = y0; y < y1; y++)
          for (int x = x0; x < x1; x++) comp[row[y * gx + x]++] = ((uint64_t)db << 32) | (uint32_t)i;
      }
  }
  uint32_t* plist = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(R ? R : 1));
  uint32_t* ranges = (uint32_t*)calloc((size_t)NT * 2, 4);
#pragma omp parallel for schedule(dynamic, 8)
  for (int t = 0; t < NT; t++) {
    const int64_t a = tcount[t], b = tcount[t + 1];
    if (b > a) {
      qsort(comp + a, (size_t)(b - a), sizeof(uint64_t), cmp_u64);
      for (int64_t e = a; e < b; e++) plist[e] = (uint32_t)(comp[e] & 0xffffffffu);
      ranges[2 * t] = (uint32_t)a; ranges[2 * t + 1] = (uint32_t)b;
    }
  }
  float* fT = out_final_T ? out_final_T : (float*)malloc(sizeof(float) * (size_t)W * H);
  uint32_t* nc = out_ncontrib ? out_ncontrib : (uint32_t*)malloc(sizeof(uint32_t) * (size_t)W * H);
  orc_render_fwd(W, H, ranges, plist, xy, colors_precomp ? colors_precomp : rgb, conic, bg, out_color, fT, nc);
  if (!out_final_T) free(fT);
  if (!out_ncontrib) free(nc);
  free(xy); free(depths); free(rgb); free(conic); free(tiles); free(clamped);
  free(tcount); free(comp); free(cur); free(plist); free(ranges);
  return R;
}

/* ---------------------------------------------------------------------------------------------
 * scene/simple_knn/cuda_headers/simple_knn.cu:147-221: the reference result is the exact mean of the
 * three smallest squared distances to OTHER points (i != idx; duplicates give 0).  The Morton
 * ordering / box pruning there is an acceleration that does not change the result, so the oracle
 * is brute force; the insertion network is updateKBest<3> (:131-145).
 */
void orc_knn_mean_dist2(int P, const float* pts, float* out) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) {
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    const float* r = pts + 3 * (size_t)i;
    for (int j = 0; j < P; j++) {
      if (j == i) continue;
      const float* q = pts + 3 * (size_t)j;
      const float dx = q[0] - r[0], dy = q[1] - r[1], dz = q[2] - r[2];
      float dist = dx * dx + dy * dy + dz * dz;
      for (int k = 0; k < 3; k++)
        if (best[k] > dist) { float t = best[k]; best[k] = dist; dist = t; }
    }
    out[i] = (best[0] + best[1] + best[2]) / 3.0f;
  }
}

/* ---------------------------------------------------------------------------------------------
 * edittool/general_utils.py:73-88 get_barycentric_coordinate (float64 there; float64 here).
 */
void orc_bary_weights(int N, const double* g, const double* p1, const double* p2, const double* p3, double* w) {
  for (int i = 0; i < N; i++) {
    double e1[3], e2[3], e3[3];
    for (int k = 0; k < 3; k++) { e1[k] = g[3 * i + k] - p1[3 * i + k]; e2[k] = g[3 * i + k] - p2[3 * i + k]; e3[k] = g[3 * i + k] - p3[3 * i + k]; }
#define CRN(a, b) sqrt(pow(a[1] * b[2] - a[2] * b[1], 2) + pow(a[2] * b[0] - a[0] * b[2], 2) + pow(a[0] * b[1] - a[1] * b[0], 2))
    const double s1 = CRN(e2, e3), s2 = CRN(e1, e3), s3 = CRN(e1, e2);
#undef CRN
    const double s = s1 + s2 + s3;
    w[3 * i] = s1 / s; w[3 * i + 1] = s2 / s; w[3 * i + 2] = s3 / s;
  }
}

/* ---------------------------------------------------------------------------------------------
 * edittool/__init__.py:103-131 SingleObjectDeform.deform_gaussian, tensor-in form.
 *   tri [N][3] int32 vertex ids of the bound face (gaussian_triangles), w [N][3] weights,
 *   dV [Vm][3] = V1 - V0, Rv/Sv [Vm][3][3] row-major per-vertex rotation / shear (pyACAP GetRS
 *   output reshaped, :112-113), cov [N][3][3], pos [N][3].
 * Outputs: pos' [N][3], cov' [N][3][3], rot [N][3][3] (= blended R transposed, :122).
 * R and S are blended linearly, no re-orthonormalisation.  The reference computes this in fp32: numpy / igl /
 * pyACAP hand over float64 arrays, but jt.array() narrows float64 to float32 (Jittor's default), so the blend and
 * the matrix products run in fp32 in an order Jittor's reduce / bmm kernels choose.  No fp32 evaluation order can
 * be pinned from the reference source, so the oracle evaluates the same algebra in double (the exact-arithmetic
 * answer) and rounds the outputs to float; the HIP kernel works in fp32 and is compared with a relative tolerance
 * of 1e-5, which any fp32 evaluation order of these 3-term sums satisfies.
 */
void orc_deform(int N, const int* tri, const float* w, const float* dV, const float* Rv, const float* Sv,
                const float* cov, const float* pos, float* pos_out, float* cov_out, float* rot_out) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; i++) {
    const int* t = tri + 3 * (size_t)i;
    const float* wi = w + 3 * (size_t)i;
    double d[3], Rb[9], Sb[9];
    for (int k = 0; k < 3; k++)
      d[k] = ((double)wi[0] * dV[3 * (size_t)t[0] + k] + (double)wi[1] * dV[3 * (size_t)t[1] + k]) + (double)wi[2] * dV[3 * (size_t)t[2] + k];
    for (int k = 0; k < 9; k++) {
      Rb[k] = ((double)wi[0] * Rv[9 * (size_t)t[0] + k] + (double)wi[1] * Rv[9 * (size_t)t[1] + k]) + (double)wi[2] * Rv[9 * (size_t)t[2] + k];
      Sb[k] = ((double)wi[0] * Sv[9 * (size_t)t[0] + k] + (double)wi[1] * Sv[9 * (size_t)t[1] + k]) + (double)wi[2] * Sv[9 * (size_t)t[2] + k];
    }
    double Rt[9], RS[9], A[9];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) Rt[3 * a + b] = Rb[3 * b + a];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++)
        RS[3 * a + b] = (Rt[3 * a] * Sb[b] + Rt[3 * a + 1] * Sb[3 + b]) + Rt[3 * a + 2] * Sb[6 + b];
    const float* C = cov + 9 * (size_t)i;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++)
        A[3 * a + b] = (RS[3 * a] * C[b] + RS[3 * a + 1] * C[3 + b]) + RS[3 * a + 2] * C[6 + b];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) /* (A * RS^T)[a][b] = sum_k A[a][k] RS[b][k] */
        cov_out[9 * (size_t)i + 3 * a + b] = (float)((A[3 * a] * RS[3 * b] + A[3 * a + 1] * RS[3 * b + 1]) + A[3 * a + 2] * RS[3 * b + 2]);
    for (int k = 0; k < 9; k++) rot_out[9 * (size_t)i + k] = (float)Rt[k];
    for (int k = 0; k < 3; k++) pos_out[3 * (size_t)i + k] = (float)((double)pos[3 * (size_t)i + k] + d[k]);
  }
}

/* ---------------------------------------------------------------------------------------------
 * edittool/__init__.py:442-448 (ObjectVisualTool.render_gaussian colour path):
 *   dir = normalize(x' - campos); dir_rot = deform_rot^T . dir; rgb = max(SH_3(dir_rot) + 0.5, 0)
 * deform_rot [N][3][3] row-major as produced by orc_deform; shs [N][16][3].
 */
void orc_sh_colors_rotated(int N, int deg, int M, const float* pos, const float* campos, const float* rot,
                           const float* shs, float* rgb) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; i++) {
    float d[3] = {pos[3 * (size_t)i] - campos[0], pos[3 * (size_t)i + 1] - campos[1], pos[3 * (size_t)i + 2] - campos[2]};
    const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    d[0] /= len; d[1] /= len; d[2] /= len;
    const float* R = rot + 9 * (size_t)i;
    /* (R^T d)[a] = sum_k R[k][a] d[k] */
    float dr[3];
    for (int a = 0; a < 3; a++) dr[a] = (R[a] * d[0] + R[3 + a] * d[1]) + R[6 + a] * d[2];
    float col[3];
    sh_eval(deg, shs + (size_t)i * M * 3, dr, col);
    for (int ch = 0; ch < 3; ch++) rgb[3 * (size_t)i + ch] = fmaxf(col[ch] + 0.5f, 0.0f);
  }
}
