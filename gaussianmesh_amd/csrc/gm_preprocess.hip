// gm_preprocess.hip -- per-Gaussian kernels: forward preprocess, markVisible, backward preprocess.
//
// Replaces (reference, RAST = gaussian_renderer/diff_gaussian_rasterizater/cuda_rasterizer):
//   RAST/forward.cu:155-256   preprocessCUDA (fwd)  + computeCov3D :118-152, computeCov2D :74-113,
//                             computeColorFromSH :20-71, in_frustum RAST/auxiliary.h:138-163
//   RAST/rasterizer_impl.cu:54-66 checkFrustum
//   RAST/backward.cu:144-274  computeCov2DCUDA, :346-396 preprocessCUDA (bwd), :20-139 SH bwd, :278-341 cov3D bwd
//
// ARITHMETIC CONTRACT: this file is compiled with FMA contraction OFF and evaluates
// every expression in the association order of the reference's C source (GLM column-major products
// expanded by hand), with correctly rounded division / sqrt (hipcc default) and ndc2Pix in double.
// The CPU oracle follows the same contract, so radii, tile rectangles, depth keys and everything
// derived from them (instance lists, ranges) are bit-identical between the two.
//
// One thread per Gaussian, 256-thread workgroups.  The kernel is HBM-streaming bound (236 B in,
// ~90 B out per Gaussian at SH degree 3); the camera matrices are wave-uniform and read through
// the scalar cache.
#include "gm_common.h"
#pragma clang fp contract(off)
#include "gm_sh.h"
#include "gm_cull.h"
#include "gm_stage.h"
#include "gm_pre_body.h"
#include <cstdlib>

namespace gm {

struct PreArgs {
  int P, D, M, W, H, gx, gy;
  const float *means, *scales, *rots, *opac, *shs, *cov3D_pre, *colors_pre, *view, *proj, *campos;
  float mod, tanx, tany, fx, fy;
  float4* splat; int* radii_int; int* radii_out; uint32_t* tiles; uint4* bin; int tile_cull; int prefiltered; uint32_t* counters; uint32_t* slots; uint32_t* coarse; float* cov3D; uint8_t* clamped; uint32_t* depth_key;
};

// STAGE_SH + DMA (the instantiated fast path, SH input with M == 16): one-wave workgroups of 64 Gaussians whose SH rows
// (192 B each) arrive in LDS by LDS-DMA as a linear image (row stride 12 granules), as in gm_deform.hip's
// deform_shade_kernel: many small workgroups in different phases and no staging registers.  Without STAGE_SH (precomputed
// colours, other M, unaligned rows) each thread walks its own row.
template <bool STAGE_SH, int TH = 256, bool DMA = false>
__global__ __launch_bounds__(TH) void preprocess_fwd_kernel(const PreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds_pre[];
  constexpr int LROW = DMA ? 12 : 13;
  const int idx = blockIdx.x * TH + threadIdx.x;
  // DMA path: every per-Gaussian input is requested BEFORE the SH rows are waited for - one memory round per wave instead of
  // rows -> mean -> rotation / scale -> opacity one after the other (the kernel is bound by latency, not by bytes)
  float in_p[3] = {0.f, 0.f, 0.f}, in_s[3] = {0.f, 0.f, 0.f}, in_op = 0.f;
  float4 in_q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (DMA && idx < a.P) {
#pragma unroll
    for (int k = 0; k < 3; k++) in_p[k] = a.means[3 * (size_t)idx + k];
    in_op = a.opac[idx];
    if (!a.cov3D_pre) {
      in_q = reinterpret_cast<const float4*>(a.rots)[idx];
#pragma unroll
      for (int k = 0; k < 3; k++) in_s[k] = a.scales[3 * (size_t)idx + k];
    }
  }
  if (STAGE_SH) {
    const size_t row0 = (size_t)blockIdx.x * TH;
    const int nrows = min(TH, a.P - (int)row0);
    if (DMA) {
      float4* l4 = reinterpret_cast<float4*>(lds_pre);
      if (nrows == TH) {
        const char* gsh = reinterpret_cast<const char*>(a.shs + row0 * 48) + threadIdx.x * 16;
        if (a.D >= 3) {
#pragma unroll
          for (int q = 0; q < 12; q++) dma16(gsh + q * (TH * 16), reinterpret_cast<char*>(l4) + q * (TH * 16));
        } else {
          // below the full degree only the leading (D+1)^2 coefficients of a row are read (forward.cu:20-71 touches no others): the
          // lanes whose 16-byte granule lies behind them sit the copy out - at degree 0 (the first 1000 training iterations,
          // train_mesh_gaussian.py:70-71) one granule of twelve; the rest of the LDS row is never looked at
          const int nq = (3 * (a.D + 1) * (a.D + 1) + 3) >> 2;
#pragma unroll
          for (int q = 0; q < 12; q++)
            if ((int)((q * TH + threadIdx.x) % 12u) < nq) dma16(gsh + q * (TH * 16), reinterpret_cast<char*>(l4) + q * (TH * 16));
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
      } else if ((int)threadIdx.x < nrows) {
#pragma unroll
        for (int c = 0; c < 12; c++) l4[threadIdx.x * 12 + c] = reinterpret_cast<const float4*>(a.shs)[(row0 + threadIdx.x) * 12 + c];
      }
    } else {
      stage_rows16<12, 13, TH>(a.shs, row0, nrows, reinterpret_cast<float4*>(lds_pre));
    }
    __syncthreads();
  }
  if (idx == 0) a.counters[GM_CNT_POLICY] = (uint32_t)a.tile_cull;   // emission policy of THIS forward (checked by duplicate_kernel / render_bwd_kernel)
  int radius_i = 0;
  uint32_t tiles = 0, dkey = 0xFFFFFFFFu;
  uint4 bin = make_uint4(0u, 0u, 0u, 0u);
  if (idx < a.P) do {
    const V3 p = DMA ? V3{in_p[0], in_p[1], in_p[2]} : V3{a.means[3 * (size_t)idx], a.means[3 * (size_t)idx + 1], a.means[3 * (size_t)idx + 2]};
    const PreCam cam = {a.W, a.H, a.gx, a.gy, a.tile_cull, a.view, a.proj, a.tanx, a.tany, a.fx, a.fy};
    {                                               // in_frustum first (auxiliary.h:153): culled Gaussians read nothing else (non-DMA path)
      const float vz = a.view[2] * p.x + a.view[6] * p.y + a.view[10] * p.z + a.view[14];
      if (vz <= 0.2f) {
        // the reference prints "Point is filtered although prefiltered is set" and traps the kernel (auxiliary.h:155-159);
        // here the forward completes, the violation is flagged in the status words and the host side turns it into an error
        if (a.prefiltered) a.counters[GM_CNT_PREFILTER] = 1u;
        break;
      }
    }
    float c3[6];
    if (a.cov3D_pre) {
#pragma unroll
      for (int k = 0; k < 6; k++) c3[k] = a.cov3D_pre[6 * (size_t)idx + k];
    } else {                                        // computeCov3D, forward.cu:118-152
      const float4 q = DMA ? in_q : reinterpret_cast<const float4*>(a.rots)[idx];
      float Rg[9], Mc[9];
      quat_cols(q.x, q.y, q.z, q.w, Rg);
      const float s[3] = {a.mod * (DMA ? in_s[0] : a.scales[3 * (size_t)idx]), a.mod * (DMA ? in_s[1] : a.scales[3 * (size_t)idx + 1]),
                          a.mod * (DMA ? in_s[2] : a.scales[3 * (size_t)idx + 2])};
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 3; k++) Mc[3 * c + k] = s[k] * Rg[3 * c + k];
#define SIG(u, w) (Mc[3 * u + 0] * Mc[3 * w + 0] + Mc[3 * u + 1] * Mc[3 * w + 1] + Mc[3 * u + 2] * Mc[3 * w + 2])
      c3[0] = SIG(0, 0); c3[1] = SIG(0, 1); c3[2] = SIG(0, 2); c3[3] = SIG(1, 1); c3[4] = SIG(1, 2); c3[5] = SIG(2, 2);
#undef SIG
#pragma unroll
      for (int k = 0; k < 6; k++) a.cov3D[6 * (size_t)idx + k] = c3[k];
    }
    PreGeom pg;
    if (!pre_project(cam, p, c3, pg)) break;
    const float pix = pg.pix, piy = pg.piy;
    float col[3];
    uint8_t clampbits = 0;
    if (a.colors_pre) {
      col[0] = a.colors_pre[3 * (size_t)idx]; col[1] = a.colors_pre[3 * (size_t)idx + 1]; col[2] = a.colors_pre[3 * (size_t)idx + 2];
    } else {                                        // computeColorFromSH, forward.cu:20-71
      float dx = p.x - a.campos[0], dy = p.y - a.campos[1], dz = p.z - a.campos[2];
      const float len = sqrtf(dx * dx + dy * dy + dz * dz);
      dx = dx / len; dy = dy / len; dz = dz / len;
      float sh[48];
      const int ncoef = (a.D + 1) * (a.D + 1);
      if (STAGE_SH) {
        const float4* row = reinterpret_cast<const float4*>(lds_pre) + threadIdx.x * LROW;
#pragma unroll
        for (int c = 0; c < 12; c++) { const float4 v = row[c]; sh[4 * c] = v.x; sh[4 * c + 1] = v.y; sh[4 * c + 2] = v.z; sh[4 * c + 3] = v.w; }
      } else {
        load_sh(a.shs, idx, a.M, ncoef, sh);
      }
#pragma unroll
      for (int ch = 0; ch < 3; ch++) {
        float r = sh_channel(a.D, [&](int i) { return sh[3 * i + ch]; }, dx, dy, dz);
        r += 0.5f;
        if (r < 0) clampbits |= (uint8_t)(1u << ch);
        col[ch] = fmaxf(r, 0.0f);
      }
    }
    a.clamped[idx] = clampbits;
    const float opac = DMA ? in_op : a.opac[idx];
    splat_store(a.splat, (size_t)idx, pix, piy, pg.conx, pg.cony, pg.conz, opac, col[0], col[1], col[2], pg.depth);
    radius_i = (int)pg.radius;
    pre_emit(cam, pg, opac, tiles, bin);
    dkey = __float_as_uint(pg.depth);
  } while (0);
  if (idx < a.P) {
    if (a.radii_out) a.radii_out[idx] = radius_i; else a.radii_int[idx] = radius_i;     // (the internal copy: for callers that pass no radii array)
    a.tiles[idx] = tiles;
    a.bin[idx] = bin;
    a.depth_key[idx] = dkey;
  }
  slot_accumulate(a.slots, a.coarse, tiles, dkey);
}

int launch_preprocess(const RasterArgs& r, GeomState& g, int* radii) {
  StageScope sc(ST_PREPROCESS, r.stream);
  PreArgs a;
  a.P = r.P; a.D = r.D; a.M = r.M; a.W = r.W; a.H = r.H;
  a.gx = (r.W + GM_TILE - 1) / GM_TILE; a.gy = (r.H + GM_TILE - 1) / GM_TILE;
  a.means = r.means3D; a.scales = r.scales; a.rots = r.rotations; a.opac = r.opacities; a.shs = r.shs;
  a.cov3D_pre = r.cov3D_precomp; a.colors_pre = r.colors_precomp; a.view = r.viewmatrix; a.proj = r.projmatrix;
  a.campos = r.cam_pos; a.mod = r.scale_modifier; a.tanx = r.tan_fovx; a.tany = r.tan_fovy;
  a.fy = r.H / (2.0f * r.tan_fovy); a.fx = r.W / (2.0f * r.tan_fovx);   // rasterizer_impl.cu:359-360
  a.splat = g.splat; a.radii_int = g.radii; a.radii_out = radii; a.tiles = g.tiles_touched; a.bin = g.bin; a.tile_cull = r.tile_cull; a.prefiltered = r.prefiltered; a.counters = g.counters; a.slots = g.slots; a.coarse = g.coarse; a.cov3D = g.cov3D;
  a.clamped = g.clamped; a.depth_key = g.depth_key;
  if (r.P > 0) {
    if (a.shs && a.M == 16 && aligned16(a.shs)) {
      hipLaunchKernelGGL((preprocess_fwd_kernel<true, 64, true>), dim3((r.P + 63) / 64), dim3(64), sizeof(float4) * 64 * 12, r.stream, a);
    }
    else if (a.shs && (a.M == 1 || a.M == 4 || a.M == 12))
      // a dense [P,1,3] / [P,4,3] / [P,12,3] tensor of the ACTIVE coefficients (the trainer's phases below the full SH degree,
      // train.Trainer(dense_dc)): nothing to stage - 12 / 48 / 144 contiguous bytes per Gaussian, 16-byte loads - but every
      // per-Gaussian input requested up front, as on the row path
      hipLaunchKernelGGL((preprocess_fwd_kernel<false, 64, true>), dim3((r.P + 63) / 64), dim3(64), 0, r.stream, a);
    else
      hipLaunchKernelGGL(preprocess_fwd_kernel<false>, dim3((r.P + 255) / 256), dim3(256), 0, r.stream, a);
  }
  GM_LAUNCH_CHECK(r.debug, r.stream);
  return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mark_visible_kernel(int P, const float* __restrict__ means,
                                                           const float* __restrict__ view, uint8_t* __restrict__ present) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const V3 p = {means[3 * (size_t)idx], means[3 * (size_t)idx + 1], means[3 * (size_t)idx + 2]};
  const V3 pv = xform4x3(p, view);
  present[idx] = (pv.z <= 0.2f) ? 0 : 1;
}

int launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s) {
  if (P > 0) hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
  GM_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// backward: computeCov2DCUDA + preprocessCUDA(bwd) fused, one thread per Gaussian.  Every gradient
// output row is written by its thread (zeros for culled Gaussians) so no memset is needed for them.
struct PreBwdArgs {
  int P, D, M, W, H;
  const float *means, *shs, *scales, *rots, *cov3D, *view, *proj, *campos;
  const int* radii; const uint8_t* clamped; const float4* splat;
  float mod, tanx, tany, fx, fy;
  const float* grad_acc;                                   // [P][12] packed accumulators written by render_bwd_kernel
  float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor;  // API outputs unpacked from grad_acc
  float *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale, *dL_drot;
  ShAdamArgs adam;                                         // adam.m != null: the SH rows' Adam step happens here (gm_backward_sh_step), dL_dsh is not written
  const uint32_t* counters;                                //   (GM_CNT_REFUSED: a refused forward's backward must not step)
  int cov2d_f32;                                           // verification aid (gm_debug_backward_cov2d_float32): conic -> cov2D in float32, as backward.cu:196-215 states it
};

template <bool STAGE_SH>
__device__ __forceinline__ void preprocess_bwd_body(const PreBwdArgs& a, const int idx, float4* lrow);

// STAGE_SH: SH rows come in through LDS and the 192-byte dL/dSH rows leave through the same LDS rows as coalesced
// 16-byte stores (gm_stage.h); these two streams are 384 of the ~560 bytes this kernel moves per Gaussian.
template <bool STAGE_SH, int TH = 256, bool DMA = false>
__global__ __launch_bounds__(TH) void preprocess_bwd_kernel(const PreBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds_pre[];
  constexpr int LROW = DMA ? 12 : 13;          // DMA: the LDS image is the linear image of the rows (in and out)
  float4* l4 = reinterpret_cast<float4*>(lds_pre);
  float4* lrow = l4 + threadIdx.x * LROW;
  const size_t row0 = (size_t)blockIdx.x * TH;
  const int nrows = min(TH, a.P - (int)row0);
  if (STAGE_SH) {
    if (DMA) {
      if (nrows == TH) {
        const char* gsh = reinterpret_cast<const char*>(a.shs + row0 * 48) + threadIdx.x * 16;
        if (a.D >= 3) {
#pragma unroll
          for (int q = 0; q < 12; q++) dma16(gsh + q * (TH * 16), reinterpret_cast<char*>(l4) + q * (TH * 16));
        } else {                                     // only the granules that hold the active degree's coefficients (see preprocess_fwd_kernel)
          const int nq = (3 * (a.D + 1) * (a.D + 1) + 3) >> 2;
#pragma unroll
          for (int q = 0; q < 12; q++)
            if ((int)((q * TH + threadIdx.x) % 12u) < nq) dma16(gsh + q * (TH * 16), reinterpret_cast<char*>(l4) + q * (TH * 16));
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
      } else if ((int)threadIdx.x < nrows) {
#pragma unroll
        for (int c = 0; c < 12; c++) lrow[c] = reinterpret_cast<const float4*>(a.shs)[(row0 + threadIdx.x) * 12 + c];
      }
    } else {
      stage_rows16<12, 13, TH>(a.shs, row0, nrows, l4);
    }
    __syncthreads();
  }
  const int idx = blockIdx.x * TH + threadIdx.x;
  if (idx < a.P) preprocess_bwd_body<STAGE_SH>(a, idx, lrow);
  if (STAGE_SH) {
    __syncthreads();
    if (DMA && a.adam.m) {
      // Adam step of the block's SH rows, fused (gm_backward_sh_step): the gradient rows sit in LDS as the linear image of the block;
      // granule i of that image <-> granule i of the parameter / moment rows, so the loop is the coalesced copy-out below with the
      // update in between.  The parameter granule is read again from global memory (the row was fetched by this block's DMA a few
      // microseconds ago: an L2 hit), m and v come from HBM, all three go back: 192 B of dL/dSH written and read and 192 B of parameter
      // read per Gaussian less than the backward + adam_kernel pair.  Rows behind adam.rows (a frozen cloud sharing the operand) and
      // the backward of a REFUSED forward (zero gradients: the iteration is repeated) are left alone.
      const long long lim = ((long long)a.adam.rows - (long long)row0) * 12;
      if (a.counters[GM_CNT_REFUSED] == 0u) {
        float4* gp = reinterpret_cast<float4*>(a.adam.p) + row0 * 12;
        float4* gm_ = reinterpret_cast<float4*>(a.adam.m) + row0 * 12;
        float4* gv = reinterpret_cast<float4*>(a.adam.v) + row0 * 12;
        // below the full degree the granules behind the active coefficients have g = m = v = 0 (they never had a gradient: the degree only
        // rises): the rule leaves them as they are, so they are not touched at all (as gm_adam_step_active)
        const int nq = a.D >= 3 ? 12 : (3 * (a.D + 1) * (a.D + 1) + 3) >> 2;
        for (int i = threadIdx.x; i < nrows * 12 && (long long)i < lim; i += TH) {
          const int c = i % 12;                                    // granule of its row: elements 4 c .. 4 c + 3 of the 48
          if (c >= nq) continue;
          const float4 g4 = l4[i];
          float4 p4 = gp[i], m4 = gm_[i], v4 = gv[i];
          float* pp = &p4.x; float* mm = &m4.x; float* vv = &v4.x; const float* gg = &g4.x;
#pragma unroll
          for (int j = 0; j < 4; j++)
            adam_update(pp[j], mm[j], vv[j], gg[j], a.adam.b1, a.adam.b2, a.adam.c1, a.adam.c2, a.adam.eps, (4 * c + j >= 3) ? a.adam.step_hi : a.adam.step_lo);
          gp[i] = p4; gm_[i] = m4; gv[i] = v4;
        }
      }
    } else if (DMA) {
      float4* dst = reinterpret_cast<float4*>(a.dL_dsh) + row0 * 12;
      for (int i = threadIdx.x; i < nrows * 12; i += TH) dst[i] = l4[i];
    } else {
      unstage_rows16<12, 13, TH>(a.dL_dsh, row0, nrows, l4);
    }
  }
}

template <bool STAGE_SH>
__device__ __forceinline__ void preprocess_bwd_body(const PreBwdArgs& a, const int idx, float4* lrow) {
  const int ncoef = (a.D + 1) * (a.D + 1);
  float dmean[3] = {0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool visible = a.radii[idx] > 0;
  // unpack the blend-stage accumulators (all zero for Gaussians no tile ever touched)
  // grad_acc = dL/dcolor rgb + the moments (1, dx, dy, dx^2, dx dy, dy^2) of h = G dL/dG over all pixels; combine them
  // with this Gaussian's conic / opacity into the reference's accumulators (backward.cu:538-554)
  const float4 acc0 = reinterpret_cast<const float4*>(a.grad_acc)[3 * (size_t)idx];
  const float4 acc1 = reinterpret_cast<const float4*>(a.grad_acc)[3 * (size_t)idx + 1];
  const float m2yy = a.grad_acc[12 * (size_t)idx + 8];
  const float dcol[3] = {acc0.x, acc0.y, acc0.z};
  float g2dx = 0.f, g2dy = 0.f, gop = 0.f;
  if (visible) {
    const float4 s0 = splat_row(a.splat, (size_t)idx, 0), s1 = splat_row(a.splat, (size_t)idx, 1);
    const float cx = s0.z, cy = s0.w, cz = s1.x, op = s1.y;
    const float m0 = acc0.w, m1x = acc1.x, m1y = acc1.y;
    g2dx = -(0.5f * a.W) * (cx * m1x + cy * m1y);
    g2dy = -(0.5f * a.H) * (cz * m1y + cy * m1x);
    gop = (m0 != 0.f) ? m0 / op : 0.f;
  }
  const float gcx = -0.5f * acc1.z, gcy = -0.5f * acc1.w, gcw = -0.5f * m2yy;
  // dL/dcolour, dL/dconic and (below) dL/dcov3D are consumed right here when the inputs are SH rows and scale / rotation: a caller
  // that does not want them passes NULL (52 of ~600 bytes per Gaussian not written)
  if (a.dL_dcolor) { a.dL_dcolor[3 * (size_t)idx] = acc0.x; a.dL_dcolor[3 * (size_t)idx + 1] = acc0.y; a.dL_dcolor[3 * (size_t)idx + 2] = acc0.z; }
  a.dL_dmean2D[3 * (size_t)idx] = g2dx; a.dL_dmean2D[3 * (size_t)idx + 1] = g2dy; a.dL_dmean2D[3 * (size_t)idx + 2] = 0.f;
  if (a.dL_dconic) reinterpret_cast<float4*>(a.dL_dconic)[idx] = make_float4(gcx, gcy, 0.f, gcw);
  a.dL_dopacity[idx] = gop;
  if (visible) {
    const V3 mean = {a.means[3 * (size_t)idx], a.means[3 * (size_t)idx + 1], a.means[3 * (size_t)idx + 2]};
    float c3[6];
#pragma unroll
    for (int k = 0; k < 6; k++) c3[k] = a.cov3D[6 * (size_t)idx + k];
    {  // ---- computeCov2DCUDA, backward.cu:144-274
      const float dcx = gcx, dcy = gcy, dcz = gcw;
      V3 t; float T0[3], T1[3], xm, ym;
      cov2d_T(mean, a.fx, a.fy, a.tanx, a.tany, a.view, t, T0, T1, xm, ym);
      // conic -> cov2D (backward.cu:196-215) in BINARY64.  The three derivatives are quadratic forms in (a, b, c) whose terms cancel by
      // det / (a c): for a 100:1 needle seen at an angle a c / det ~ 500 and the float32 statement of the reference loses the second
      // to third digit of dL/dscale, dL/drot and dL/dmean (tools/needle_stages.py: the float64 blend-stage gradients through the
      // float32 formula are 4e-3 ... 1.2e-2 of the tensor's size away from float64 autograd; with THIS block in binary64 - the
      // covariance entries from the float32 T and cov3D, the determinant, the three derivatives - 5e-5 ... 6e-4).  ~60 double
      // operations per Gaussian in a kernel that waits for memory.  On well-conditioned splats the result is the float32 one to 1e-6.
      double dL_da_d = 0, dL_db_d = 0, dL_dc_d = 0;
      {
        const double V0[3] = {c3[0], c3[1], c3[2]}, V1[3] = {c3[1], c3[3], c3[4]}, V2[3] = {c3[2], c3[4], c3[5]};
        double A0[3], A1[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          A0[k] = (double)T0[0] * V0[k] + (double)T0[1] * V1[k] + (double)T0[2] * V2[k];
          A1[k] = (double)T1[0] * V0[k] + (double)T1[1] * V1[k] + (double)T1[2] * V2[k];
        }
        const double ca = A0[0] * T0[0] + A0[1] * T0[1] + A0[2] * T0[2] + 0.3;
        const double cb = A1[0] * T0[0] + A1[1] * T0[1] + A1[2] * T0[2];
        const double cc = A1[0] * T1[0] + A1[1] * T1[1] + A1[2] * T1[2] + 0.3;
        const double denom = ca * cc - cb * cb;
        const float d2f = (float)(denom * denom) + 0.0000001f;                  // (the reciprocal itself is well-conditioned: float32)
        const double denom2inv = (double)(1.0f / d2f);
        if (denom2inv != 0) {
          const double dcx_d = dcx, dcy_d = dcy, dcz_d = dcz;
          dL_da_d = denom2inv * (-cc * cc * dcx_d + 2 * cb * cc * dcy_d - cb * cb * dcz_d);          // (denom - a c = -b^2)
          dL_dc_d = denom2inv * (-ca * ca * dcz_d + 2 * ca * cb * dcy_d - cb * cb * dcx_d);
          dL_db_d = denom2inv * 2 * (cb * cc * dcx_d - (ca * cc + cb * cb) * dcy_d + ca * cb * dcz_d);   // (denom + 2 b^2 = a c + b^2)
        }
      }
      float dL_da = (float)dL_da_d, dL_db = (float)dL_db_d, dL_dc = (float)dL_dc_d;
      if (a.cov2d_f32) {
        // The reference's own arithmetic for this block (tests only; wave-uniform): cov2D from T and cov3D, the determinant and the
        // three derivatives in float32 exactly as backward.cu:196-215 writes them (and as oracle/gm_oracle.c restates them).  With it
        // the needle scenes are within 1e-3 of the C oracle again: the binary64 block above is the ONLY deviation from the reference
        // formula on the backward path (tests/test_gpu_fuzz_parity.py).
        float ca, cb, cc;
        cov2d_from_T(T0, T1, c3, ca, cb, cc);
        ca += 0.3f; cc += 0.3f;
        const float denom = ca * cc - cb * cb;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        dL_da = 0.f; dL_db = 0.f; dL_dc = 0.f;
        if (denom2inv != 0) {
          dL_da = denom2inv * (-cc * cc * dcx + 2 * cb * cc * dcy + (denom - ca * cc) * dcz);
          dL_dc = denom2inv * (-ca * ca * dcz + 2 * ca * cb * dcy + (denom - ca * cc) * dcx);
          dL_db = denom2inv * 2 * (cb * cc * dcx - (denom + 2 * cb * cb) * dcy + ca * cb * dcz);
        }
      }
      {
        dcov[0] = (T0[0] * T0[0] * dL_da + T0[0] * T1[0] * dL_db + T1[0] * T1[0] * dL_dc);
        dcov[3] = (T0[1] * T0[1] * dL_da + T0[1] * T1[1] * dL_db + T1[1] * T1[1] * dL_dc);
        dcov[5] = (T0[2] * T0[2] * dL_da + T0[2] * T1[2] * dL_db + T1[2] * T1[2] * dL_dc);
        dcov[1] = 2 * T0[0] * T0[1] * dL_da + (T0[0] * T1[1] + T0[1] * T1[0]) * dL_db + 2 * T1[0] * T1[1] * dL_dc;
        dcov[2] = 2 * T0[0] * T0[2] * dL_da + (T0[0] * T1[2] + T0[2] * T1[0]) * dL_db + 2 * T1[0] * T1[2] * dL_dc;
        dcov[4] = 2 * T0[2] * T0[1] * dL_da + (T0[1] * T1[2] + T0[2] * T1[1]) * dL_db + 2 * T1[1] * T1[2] * dL_dc;
      }
      const float V[9] = {c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]};
#define VR(cc_, rr_) V[3 * (cc_) + (rr_)]
      const float dT00 = 2 * (T0[0] * VR(0, 0) + T0[1] * VR(0, 1) + T0[2] * VR(0, 2)) * dL_da + (T1[0] * VR(0, 0) + T1[1] * VR(0, 1) + T1[2] * VR(0, 2)) * dL_db;
      const float dT01 = 2 * (T0[0] * VR(1, 0) + T0[1] * VR(1, 1) + T0[2] * VR(1, 2)) * dL_da + (T1[0] * VR(1, 0) + T1[1] * VR(1, 1) + T1[2] * VR(1, 2)) * dL_db;
      const float dT02 = 2 * (T0[0] * VR(2, 0) + T0[1] * VR(2, 1) + T0[2] * VR(2, 2)) * dL_da + (T1[0] * VR(2, 0) + T1[1] * VR(2, 1) + T1[2] * VR(2, 2)) * dL_db;
      const float dT10 = 2 * (T1[0] * VR(0, 0) + T1[1] * VR(0, 1) + T1[2] * VR(0, 2)) * dL_dc + (T0[0] * VR(0, 0) + T0[1] * VR(0, 1) + T0[2] * VR(0, 2)) * dL_db;
      const float dT11 = 2 * (T1[0] * VR(1, 0) + T1[1] * VR(1, 1) + T1[2] * VR(1, 2)) * dL_dc + (T0[0] * VR(1, 0) + T0[1] * VR(1, 1) + T0[2] * VR(1, 2)) * dL_db;
      const float dT12 = 2 * (T1[0] * VR(2, 0) + T1[1] * VR(2, 1) + T1[2] * VR(2, 2)) * dL_dc + (T0[0] * VR(2, 0) + T0[1] * VR(2, 1) + T0[2] * VR(2, 2)) * dL_db;
#undef VR
      const float* v = a.view;
      const float dJ00 = v[0] * dT00 + v[4] * dT01 + v[8] * dT02;
      const float dJ02 = v[2] * dT00 + v[6] * dT01 + v[10] * dT02;
      const float dJ11 = v[1] * dT10 + v[5] * dT11 + v[9] * dT12;
      const float dJ12 = v[2] * dT10 + v[6] * dT11 + v[10] * dT12;
      const float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
      const float dtx = xm * -a.fx * tz2 * dJ02;
      const float dty = ym * -a.fy * tz2 * dJ12;
      const float dtz = -a.fx * tz2 * dJ00 - a.fy * tz2 * dJ11 + (2 * a.fx * t.x) * tz3 * dJ02 + (2 * a.fy * t.y) * tz3 * dJ12;
      dmean[0] = v[0] * dtx + v[1] * dty + v[2] * dtz;     // transformVec4x3Transpose
      dmean[1] = v[4] * dtx + v[5] * dty + v[6] * dtz;
      dmean[2] = v[8] * dtx + v[9] * dty + v[10] * dtz;
    }
    {  // ---- projection part, backward.cu:370-387
      const float* pr = a.proj;
      const float hw = pr[3] * mean.x + pr[7] * mean.y + pr[11] * mean.z + pr[15];
      const float m_w = 1.0f / (hw + 0.0000001f);
      const float mul1 = (pr[0] * mean.x + pr[4] * mean.y + pr[8] * mean.z + pr[12]) * m_w * m_w;
      const float mul2 = (pr[1] * mean.x + pr[5] * mean.y + pr[9] * mean.z + pr[13]) * m_w * m_w;
      const float gxx = g2dx, gyy = g2dy;
      dmean[0] += (pr[0] * m_w - pr[3] * mul1) * gxx + (pr[1] * m_w - pr[3] * mul2) * gyy;
      dmean[1] += (pr[4] * m_w - pr[7] * mul1) * gxx + (pr[5] * m_w - pr[7] * mul2) * gyy;
      dmean[2] += (pr[8] * m_w - pr[11] * mul1) * gxx + (pr[9] * m_w - pr[11] * mul2) * gyy;
    }
    if (a.shs) {  // ---- computeColorFromSH backward, backward.cu:20-139
      const float ox = mean.x - a.campos[0], oy = mean.y - a.campos[1], oz = mean.z - a.campos[2];
      const float len = sqrtf(ox * ox + oy * oy + oz * oz);
      const float x = ox / len, y = oy / len, z = oz / len;
      float sh[48];
      if (STAGE_SH) {
#pragma unroll
        for (int c = 0; c < 12; c++) { const float4 v = lrow[c]; sh[4 * c] = v.x; sh[4 * c + 1] = v.y; sh[4 * c + 2] = v.z; sh[4 * c + 3] = v.w; }
      } else {
        load_sh(a.shs, idx, a.M, ncoef, sh);
      }
      const uint8_t cl = a.clamped[idx];
      float dRGB[3], wgt[16];
#pragma unroll
      for (int ch = 0; ch < 3; ch++) dRGB[ch] = dcol[ch] * (((cl >> ch) & 1) ? 0.f : 1.f);
      float ddx[3] = {0, 0, 0}, ddy[3] = {0, 0, 0}, ddz[3] = {0, 0, 0};
#define S(i, ch) sh[3 * (i) + (ch)]
      wgt[0] = SH_C0;
      if (a.D > 0) {
        wgt[1] = -SH_C1 * y; wgt[2] = SH_C1 * z; wgt[3] = -SH_C1 * x;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) { ddx[ch] = -SH_C1 * S(3, ch); ddy[ch] = -SH_C1 * S(1, ch); ddz[ch] = SH_C1 * S(2, ch); }
        if (a.D > 1) {
          const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          wgt[4] = SH_C2[0] * xy; wgt[5] = SH_C2[1] * yz; wgt[6] = SH_C2[2] * (2.f * zz - xx - yy);
          wgt[7] = SH_C2[3] * xz; wgt[8] = SH_C2[4] * (xx - yy);
#pragma unroll
          for (int ch = 0; ch < 3; ch++) {
            ddx[ch] += SH_C2[0] * y * S(4, ch) + SH_C2[2] * 2.f * -x * S(6, ch) + SH_C2[3] * z * S(7, ch) + SH_C2[4] * 2.f * x * S(8, ch);
            ddy[ch] += SH_C2[0] * x * S(4, ch) + SH_C2[1] * z * S(5, ch) + SH_C2[2] * 2.f * -y * S(6, ch) + SH_C2[4] * 2.f * -y * S(8, ch);
            ddz[ch] += SH_C2[1] * y * S(5, ch) + SH_C2[2] * 2.f * 2.f * z * S(6, ch) + SH_C2[3] * x * S(7, ch);
          }
          if (a.D > 2) {
            wgt[9] = SH_C3[0] * y * (3.f * xx - yy); wgt[10] = SH_C3[1] * xy * z;
            wgt[11] = SH_C3[2] * y * (4.f * zz - xx - yy); wgt[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
            wgt[13] = SH_C3[4] * x * (4.f * zz - xx - yy); wgt[14] = SH_C3[5] * z * (xx - yy);
            wgt[15] = SH_C3[6] * x * (xx - 3.f * yy);
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
              ddx[ch] += (SH_C3[0] * S(9, ch) * 3.f * 2.f * xy + SH_C3[1] * S(10, ch) * yz + SH_C3[2] * S(11, ch) * -2.f * xy +
                          SH_C3[3] * S(12, ch) * -3.f * 2.f * xz + SH_C3[4] * S(13, ch) * (-3.f * xx + 4.f * zz - yy) +
                          SH_C3[5] * S(14, ch) * 2.f * xz + SH_C3[6] * S(15, ch) * 3.f * (xx - yy));
              ddy[ch] += (SH_C3[0] * S(9, ch) * 3.f * (xx - yy) + SH_C3[1] * S(10, ch) * xz +
                          SH_C3[2] * S(11, ch) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * S(12, ch) * -3.f * 2.f * yz +
                          SH_C3[4] * S(13, ch) * -2.f * xy + SH_C3[5] * S(14, ch) * -2.f * yz + SH_C3[6] * S(15, ch) * -3.f * 2.f * xy);
              ddz[ch] += (SH_C3[1] * S(10, ch) * xy + SH_C3[2] * S(11, ch) * 4.f * 2.f * yz +
                          SH_C3[3] * S(12, ch) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * S(13, ch) * 4.f * 2.f * xz +
                          SH_C3[5] * S(14, ch) * (xx - yy));
            }
          }
        }
      }
#undef S
      if (STAGE_SH) {                              // M == 16: the row goes back through LDS
        float o[48];
#pragma unroll
        for (int i = 0; i < 16; i++)
#pragma unroll
          for (int ch = 0; ch < 3; ch++) o[3 * i + ch] = (i < ncoef) ? wgt[i] * dRGB[ch] : 0.f;
#pragma unroll
        for (int c = 0; c < 12; c++) lrow[c] = make_float4(o[4 * c], o[4 * c + 1], o[4 * c + 2], o[4 * c + 3]);
      } else if (((a.M * 3) & 3) == 0 && a.M <= 16 && (reinterpret_cast<uintptr_t>(a.dL_dsh) & 15) == 0) {        // rows of whole 16-byte granules (the dense [P,4,3] / [P,12,3] leaves): 16-byte stores
        float4* dsh4 = reinterpret_cast<float4*>(a.dL_dsh + (size_t)idx * a.M * 3);
        float o[48];
#pragma unroll
        for (int i = 0; i < 16; i++)
#pragma unroll
          for (int ch = 0; ch < 3; ch++) o[3 * i + ch] = (i < ncoef) ? wgt[i] * dRGB[ch] : 0.f;
#pragma unroll
        for (int c = 0; c < 12; c++)
          if (4 * c < a.M * 3) dsh4[c] = make_float4(o[4 * c], o[4 * c + 1], o[4 * c + 2], o[4 * c + 3]);
      } else {
        float* dsh = a.dL_dsh + (size_t)idx * a.M * 3;
#pragma unroll
        for (int i = 0; i < 16; i++) {
          if (i < a.M) {
            const bool on = i < ncoef;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) dsh[3 * i + ch] = on ? wgt[i] * dRGB[ch] : 0.f;
          }
        }
        for (int i = 16; i < a.M; i++)
          for (int ch = 0; ch < 3; ch++) dsh[3 * i + ch] = 0.f;
      }
      const float ddir[3] = {ddx[0] * dRGB[0] + ddx[1] * dRGB[1] + ddx[2] * dRGB[2],
                             ddy[0] * dRGB[0] + ddy[1] * dRGB[1] + ddy[2] * dRGB[2],
                             ddz[0] * dRGB[0] + ddz[1] * dRGB[1] + ddz[2] * dRGB[2]};
      // dnormvdv, auxiliary.h:106-116
      const float sum2 = ox * ox + oy * oy + oz * oz;
      const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
      dmean[0] += ((+sum2 - ox * ox) * ddir[0] - oy * ox * ddir[1] - oz * ox * ddir[2]) * invsum32;
      dmean[1] += (-ox * oy * ddir[0] + (sum2 - oy * oy) * ddir[1] - oz * oy * ddir[2]) * invsum32;
      dmean[2] += (-ox * oz * ddir[0] - oy * oz * ddir[1] + (sum2 - oz * oz) * ddir[2]) * invsum32;
    }
  } else if (a.shs) {
    if (STAGE_SH) {
#pragma unroll
      for (int c = 0; c < 12; c++) lrow[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else if (((a.M * 3) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.dL_dsh) & 15) == 0) {
      float4* dsh4 = reinterpret_cast<float4*>(a.dL_dsh + (size_t)idx * a.M * 3);
      for (int c = 0; 4 * c < a.M * 3; c++) dsh4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      float* dsh = a.dL_dsh + (size_t)idx * a.M * 3;
      for (int i = 0; i < a.M * 3; i++) dsh[i] = 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) a.dL_dmean3D[3 * (size_t)idx + k] = dmean[k];
#pragma unroll
  for (int k = 0; k < 6; k++) if (a.dL_dcov3D) a.dL_dcov3D[6 * (size_t)idx + k] = dcov[k];
  if (a.scales) {  // ---- computeCov3D backward, backward.cu:278-341
    float dsc[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 0};
    if (visible) {
      const float4 q = reinterpret_cast<const float4*>(a.rots)[idx];
      const float r = q.x, x = q.y, y = q.z, z = q.w;
      float Rg[9], Mc[9];
      quat_cols(r, x, y, z, Rg);
      const float s[3] = {a.mod * a.scales[3 * (size_t)idx], a.mod * a.scales[3 * (size_t)idx + 1], a.mod * a.scales[3 * (size_t)idx + 2]};
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 3; k++) Mc[3 * c + k] = s[k] * Rg[3 * c + k];
      const float* d = dcov;
      const float dS[9] = {d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2], 0.5f * d[4], d[5]};
      float dM[9];
#pragma unroll
      for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++)
          dM[3 * j + i] = (2.0f * Mc[0 + i]) * dS[3 * j + 0] + (2.0f * Mc[3 + i]) * dS[3 * j + 1] + (2.0f * Mc[6 + i]) * dS[3 * j + 2];
      float Rt[9], dMt[9];
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int rr = 0; rr < 3; rr++) { Rt[3 * c + rr] = Rg[3 * rr + c]; dMt[3 * c + rr] = dM[3 * rr + c]; }
#pragma unroll
      for (int c = 0; c < 3; c++) dsc[c] = Rt[3 * c + 0] * dMt[3 * c + 0] + Rt[3 * c + 1] * dMt[3 * c + 1] + Rt[3 * c + 2] * dMt[3 * c + 2];
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int rr = 0; rr < 3; rr++) dMt[3 * c + rr] *= s[c];
#define DM(c_, r_) dMt[3 * (c_) + (r_)]
      dq[0] = 2 * z * (DM(0, 1) - DM(1, 0)) + 2 * y * (DM(2, 0) - DM(0, 2)) + 2 * x * (DM(1, 2) - DM(2, 1));
      dq[1] = 2 * y * (DM(1, 0) + DM(0, 1)) + 2 * z * (DM(2, 0) + DM(0, 2)) + 2 * r * (DM(1, 2) - DM(2, 1)) - 4 * x * (DM(2, 2) + DM(1, 1));
      dq[2] = 2 * x * (DM(1, 0) + DM(0, 1)) + 2 * r * (DM(2, 0) - DM(0, 2)) + 2 * z * (DM(1, 2) + DM(2, 1)) - 4 * y * (DM(2, 2) + DM(0, 0));
      dq[3] = 2 * r * (DM(0, 1) - DM(1, 0)) + 2 * x * (DM(2, 0) + DM(0, 2)) + 2 * y * (DM(1, 2) + DM(2, 1)) - 4 * z * (DM(1, 1) + DM(0, 0));
#undef DM
    }
#pragma unroll
    for (int k = 0; k < 3; k++) a.dL_dscale[3 * (size_t)idx + k] = dsc[k];
    reinterpret_cast<float4*>(a.dL_drot)[idx] = make_float4(dq[0], dq[1], dq[2], dq[3]);
  }
}

static bool g_bwd_cov2d_f32 = false;                     // verification aid (tests only), see PreBwdArgs::cov2d_f32
extern "C" void gm_debug_backward_cov2d_float32(int on) { g_bwd_cov2d_f32 = on != 0; }

int launch_preprocess_bwd(const RasterArgs& r, GeomState& g, const int* radii, float* dL_dmean2D, float* dL_dconic,
                          float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                          float* dL_dscale, float* dL_drot, const ShAdamArgs* sh_adam) {
  StageScope sc(ST_PREPROCESS_BWD, r.stream);
  PreBwdArgs a;
  a.P = r.P; a.D = r.D; a.M = r.M; a.W = r.W; a.H = r.H;
  a.means = r.means3D; a.shs = r.shs; a.scales = r.scales; a.rots = r.rotations;
  a.cov3D = r.cov3D_precomp ? r.cov3D_precomp : g.cov3D;
  a.view = r.viewmatrix; a.proj = r.projmatrix; a.campos = r.cam_pos;
  a.radii = radii ? radii : g.radii; a.clamped = g.clamped; a.splat = g.splat;
  a.mod = r.scale_modifier; a.tanx = r.tan_fovx; a.tany = r.tan_fovy;
  a.fy = r.H / (2.0f * r.tan_fovy); a.fx = r.W / (2.0f * r.tan_fovx);
  a.grad_acc = g.grad_acc;
  a.cov2d_f32 = g_bwd_cov2d_f32 ? 1 : 0;
  a.adam = ShAdamArgs{};
  a.counters = g.counters;
  if (sh_adam) {
    if (!(a.shs && a.M == 16 && aligned16(a.shs) && aligned16(sh_adam->m) && aligned16(sh_adam->v)) || sh_adam->p != a.shs) {
      set_error("gm_backward_sh_step: needs the [P,16,3] shs operand itself as the parameter, 16-byte aligned rows and moments"); return 1;
    }
    a.adam = *sh_adam;
  }
  a.dL_dmean2D = dL_dmean2D; a.dL_dconic = dL_dconic; a.dL_dopacity = dL_dopacity; a.dL_dcolor = dL_dcolor;
  a.dL_dmean3D = dL_dmean3D; a.dL_dcov3D = dL_dcov3D; a.dL_dsh = dL_dsh; a.dL_dscale = dL_dscale; a.dL_drot = dL_drot;
  if (r.P > 0) {
    if (a.shs && a.M == 16 && aligned16(a.shs) && (sh_adam || aligned16(a.dL_dsh))) {
      hipLaunchKernelGGL((preprocess_bwd_kernel<true, 64, true>), dim3((r.P + 63) / 64), dim3(64), sizeof(float4) * 64 * 12, r.stream, a);
    }
    else
      hipLaunchKernelGGL(preprocess_bwd_kernel<false>, dim3((r.P + 255) / 256), dim3(256), 0, r.stream, a);
  }
  GM_LAUNCH_CHECK(r.debug, r.stream);
  return 0;
}

}  // namespace gm
