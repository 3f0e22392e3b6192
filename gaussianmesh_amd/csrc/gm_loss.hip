// gm_loss.hip -- photometric loss of the training loop: L1 and SSIM (11x11 Gaussian window, sigma 1.5, zero padding)
// with the SSIM gradient, as two single-halo tile kernels.
//
// Replaces (reference): utils/loss_utils.py:17-18 (l1_loss), :23-81 (gaussian / create_window / ssim / _ssim: five
// depthwise 11x11 convolutions per call + elementwise passes, then Jittor autograd through them), as used by
// train_mesh_gaussian.py:92-94:  loss = (1 - l) * L1 + l * (1 - ssim(image, gt)).
//
// The 11x11 window is the outer product of a normalised 1-D Gaussian (loss_utils.py:23-32), so the convolutions are
// done separably (11 + 11 taps) on a 32x32 pixel tile with a 5-pixel halo staged in LDS.
//   forward : x = image, y = gt.  mu1, mu2, E[xx], E[yy], E[xy] -> ssim map S; per-workgroup partial sums of S and of
//             |x - y|; optionally the three partial derivatives dS/dmu1, dS/dE[xx], dS/dE[xy] per pixel.
//   backward: dL/dx = conv(g dS/dmu1) + 2 x conv(g dS/dE[xx]) + y conv(g dS/dE[xy]) + g_l1 sign(x - y)
//             (the window is symmetric, so the adjoint of the correlation is the same correlation).
// HBM traffic per pixel-channel: forward 8 B in + 12 B out, backward 20 B in + 4 B out.
#include "gm_common.h"

namespace gm {

#define LS_TILE 32
#define LS_HALO 5
#define LS_SPAN (LS_TILE + 2 * LS_HALO)     // 42
#define LS_THREADS 256

struct LossWindow { float w[11]; };
typedef float lv2f __attribute__((ext_vector_type(2)));

static LossWindow make_window() {
  // loss_utils.py:23-25: exp(-(x - 5)^2 / (2 sigma^2)) in Python doubles -> float32 array -> divided by its float32 sum
  LossWindow lw;
  float g[11], sum = 0.f;
  for (int i = 0; i < 11; i++) { g[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += g[i]; }
  for (int i = 0; i < 11; i++) lw.w[i] = g[i] / sum;
  return lw;
}

__device__ __forceinline__ float block_sum(float v, float* red /*[4]*/) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

template <bool WRITE_MAPS>
__global__ __launch_bounds__(LS_THREADS) void ssim_fwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                                                              int H, int W, LossWindow win, float* __restrict__ d_mu1,
                                                              float* __restrict__ d_e11, float* __restrict__ d_e12,
                                                              float* __restrict__ partial) {
  // Quantities travel in PAIRS - (x, y), (xx, yy), then xy alone - so that the 11-tap sums run on the packed-f32 pipe (v_pk_fma_f32:
  // two of them per issue slot): three instructions per tap and output instead of five, and one 8-byte LDS access per pair.  Every sum
  // keeps its own order of additions: results are bit-identical to the one-quantity-at-a-time form.
  // LDS: the staged inputs and the horizontal sums share their memory (27.1 KiB per workgroup, five workgroups per CU instead of the
  // three that 41 KiB allowed): the horizontal pass keeps its results in registers until every thread has read its inputs.
  struct Horiz { float2 hb01[LS_SPAN][LS_TILE + 1], hb23[LS_SPAN][LS_TILE + 1]; float hb4[LS_SPAN][LS_TILE + 1]; };
  __shared__ __attribute__((aligned(16))) char lds_raw[sizeof(Horiz)];
  static_assert(sizeof(float2) * LS_SPAN * (LS_SPAN + 1) <= sizeof(Horiz), "the staged inputs fit the horizontal sums' memory");
  float2 (*sxy)[LS_SPAN + 1] = reinterpret_cast<float2 (*)[LS_SPAN + 1]>(lds_raw);          // staged (image, target) with the halo
  Horiz& hz = *reinterpret_cast<Horiz*>(lds_raw);                  // horizontal sums of (x, y), (xx, yy) and xy
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int ox = blockIdx.x * LS_TILE, oy = blockIdx.y * LS_TILE;
  const size_t plane = (size_t)blockIdx.z * H * W;
  for (int i = tid; i < LS_SPAN * LS_SPAN; i += LS_THREADS) {
    const int r = i / LS_SPAN, c = i - r * LS_SPAN;
    const int gx = ox + c - LS_HALO, gy = oy + r - LS_HALO;
    const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;          // zero padding (conv2d padding = 5)
    const size_t p = plane + (size_t)(in ? gy : 0) * W + (in ? gx : 0);
    sxy[r][c] = make_float2(in ? img1[p] : 0.f, in ? img2[p] : 0.f);
  }
  __syncthreads();
  // horizontal taps: one work item = 4 adjacent output columns of one row (14 staged values feed 4 x 11 taps);
  // consecutive lanes take consecutive rows (row stride 43 pairs: conflict-free)
  constexpr int HITEMS = LS_SPAN * (LS_TILE / 4), HPASS = (HITEMS + LS_THREADS - 1) / LS_THREADS;     // 336 items, 2 passes
  lv2f h01[HPASS][4], h23[HPASS][4];
  float h4[HPASS][4];
#pragma unroll
  for (int ps = 0; ps < HPASS; ps++) {
    const int i = tid + ps * LS_THREADS;
    if (i < HITEMS) {
      const int r = i % LS_SPAN, c0 = (i / LS_SPAN) * 4;
      lv2f p0[14], p1[14];
      float xy[14];
#pragma unroll
      for (int k = 0; k < 14; k++) {
        const float2 v = sxy[r][c0 + k];
        p0[k] = lv2f{v.x, v.y};
        p1[k] = p0[k] * p0[k];
        xy[k] = v.x * v.y;
      }
#pragma unroll
      for (int o = 0; o < 4; o++) {
        lv2f a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
        float a4 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
          const float w = win.w[k];
          const lv2f ww = {w, w};
          a01 = ww * p0[o + k] + a01; a23 = ww * p1[o + k] + a23; a4 += w * xy[o + k];
        }
        h01[ps][o] = a01; h23[ps][o] = a23; h4[ps][o] = a4;
      }
    }
  }
  // the thread's own four pixels (L1 term) before the staged inputs are overwritten
  float2 own[4];
#pragma unroll
  for (int j = 0; j < 4; j++) own[j] = sxy[(tid >> 5) * 4 + j + LS_HALO][(tid & 31) + LS_HALO];
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < HPASS; ps++) {
    const int i = tid + ps * LS_THREADS;
    if (i < HITEMS) {
      const int r = i % LS_SPAN, c0 = (i / LS_SPAN) * 4;
#pragma unroll
      for (int o = 0; o < 4; o++) {
        hz.hb01[r][c0 + o] = make_float2(h01[ps][o].x, h01[ps][o].y); hz.hb23[r][c0 + o] = make_float2(h23[ps][o].x, h23[ps][o].y);
        hz.hb4[r][c0 + o] = h4[ps][o];
      }
    }
  }
  __syncthreads();
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  const int c = tid & 31;
  float s_sum = 0.f, l1_sum = 0.f;
  // vertical taps: a thread owns 4 adjacent rows of one column (14 values per quantity feed 4 x 11 taps)
  float vq[5][4];
  {
    lv2f c01[14], c23[14];
    float c4[14];
#pragma unroll
    for (int k = 0; k < 14; k++) {
      const float2 u = hz.hb01[(tid >> 5) * 4 + k][c], v = hz.hb23[(tid >> 5) * 4 + k][c];
      c01[k] = lv2f{u.x, u.y}; c23[k] = lv2f{v.x, v.y}; c4[k] = hz.hb4[(tid >> 5) * 4 + k][c];
    }
#pragma unroll
    for (int o = 0; o < 4; o++) {
      lv2f a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
      float a4 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) {
        const float w = win.w[k];
        const lv2f ww = {w, w};
        a01 = ww * c01[o + k] + a01; a23 = ww * c23[o + k] + a23; a4 += w * c4[o + k];
      }
      vq[0][o] = a01.x; vq[1][o] = a01.y; vq[2][o] = a23.x; vq[3][o] = a23.y; vq[4][o] = a4;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int r = (tid >> 5) * 4 + j;
    const float mu1 = vq[0][j], mu2 = vq[1][j], e11 = vq[2][j], e22 = vq[3][j], e12 = vq[4][j];
    const int gx = ox + c, gy = oy + r;
    if (gx < W && gy < H) {
      const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
      const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
      const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = s1 + s2 + C2;
      const float inv_b1 = 1.0f / B1, inv_b2 = 1.0f / B2;
      const float S = (A1 * A2) * (inv_b1 * inv_b2);
      s_sum += S;
      l1_sum += fabsf(own[j].x - own[j].y);
      if (WRITE_MAPS) {
        const size_t p = plane + (size_t)gy * W + gx;
        // S as a function of (mu1, E[xx], E[xy]) with sigma1^2 = E[xx] - mu1^2, sigma12 = E[xy] - mu1 mu2
        const float dS_ds1 = -S * inv_b2;                       // = dS/dE[xx]
        const float dS_ds12 = 2.f * A1 * (inv_b1 * inv_b2);     // = dS/dE[xy]
        d_mu1[p] = 2.f * mu2 * A2 * (inv_b1 * inv_b2) - 2.f * mu1 * S * inv_b1 - 2.f * mu1 * dS_ds1 - mu2 * dS_ds12;
        d_e11[p] = dS_ds1;
        d_e12[p] = dS_ds12;
      }
    }
  }
  const float ts = block_sum(s_sum, red);
  const float tl = block_sum(l1_sum, red);
  if (tid == 0) {
    const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partial[2 * b] = ts;
    partial[2 * b + 1] = tl;
  }
}

__global__ __launch_bounds__(LS_THREADS) void ssim_bwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                                                              const float* __restrict__ d_mu1, const float* __restrict__ d_e11,
                                                              const float* __restrict__ d_e12, int H, int W, LossWindow win,
                                                              const float* __restrict__ g_ssim /*[planes]*/,
                                                              const float* __restrict__ g_l1 /*[1] or null*/,
                                                              float* __restrict__ dL_dimg1) {
  // (dS/dmu1, dS/dE[xx]) travel as a pair, dS/dE[xy] alone: packed-f32 sums as in ssim_fwd_kernel
  __shared__ float2 sm01[LS_SPAN][LS_SPAN + 1];
  __shared__ float sm2[LS_SPAN][LS_SPAN + 1];
  __shared__ float2 hb01[LS_SPAN][LS_TILE + 1];
  __shared__ float hb2[LS_SPAN][LS_TILE + 1];
  const int tid = threadIdx.x;
  const int ox = blockIdx.x * LS_TILE, oy = blockIdx.y * LS_TILE;
  const size_t plane = (size_t)blockIdx.z * H * W;
  for (int i = tid; i < LS_SPAN * LS_SPAN; i += LS_THREADS) {
    const int r = i / LS_SPAN, c = i - r * LS_SPAN;
    const int gx = ox + c - LS_HALO, gy = oy + r - LS_HALO;
    const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;          // no ssim-map pixel outside the image
    const size_t p = plane + (size_t)(in ? gy : 0) * W + (in ? gx : 0);
    sm01[r][c] = make_float2(in ? d_mu1[p] : 0.f, in ? d_e11[p] : 0.f);
    sm2[r][c] = in ? d_e12[p] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < LS_SPAN * (LS_TILE / 4); i += LS_THREADS) {      // see ssim_fwd_kernel
    const int r = i % LS_SPAN, c0 = (i / LS_SPAN) * 4;
    lv2f v01[14];
    float v2[14];
#pragma unroll
    for (int k = 0; k < 14; k++) { const float2 u = sm01[r][c0 + k]; v01[k] = lv2f{u.x, u.y}; v2[k] = sm2[r][c0 + k]; }
#pragma unroll
    for (int o = 0; o < 4; o++) {
      lv2f a01 = {0.f, 0.f};
      float a2 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) {
        const float w = win.w[k];
        const lv2f ww = {w, w};
        a01 = ww * v01[o + k] + a01; a2 += w * v2[o + k];
      }
      hb01[r][c0 + o] = make_float2(a01.x, a01.y); hb2[r][c0 + o] = a2;
    }
  }
  __syncthreads();
  const float gs = g_ssim[blockIdx.z];
  const float gl = g_l1 ? g_l1[0] : 0.f;
  const int c = tid & 31;
  float vq[3][4];
  {
    lv2f c01[14];
    float c2[14];
#pragma unroll
    for (int k = 0; k < 14; k++) { const float2 u = hb01[(tid >> 5) * 4 + k][c]; c01[k] = lv2f{u.x, u.y}; c2[k] = hb2[(tid >> 5) * 4 + k][c]; }
#pragma unroll
    for (int o = 0; o < 4; o++) {
      lv2f a01 = {0.f, 0.f};
      float a2 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) {
        const float w = win.w[k];
        const lv2f ww = {w, w};
        a01 = ww * c01[o + k] + a01; a2 += w * c2[o + k];
      }
      vq[0][o] = a01.x; vq[1][o] = a01.y; vq[2][o] = a2;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int r = (tid >> 5) * 4 + j;
    const float A = vq[0][j], B = vq[1][j], Cc = vq[2][j];
    const int gx = ox + c, gy = oy + r;
    if (gx < W && gy < H) {
      const size_t p = plane + (size_t)gy * W + gx;
      const float x = img1[p], y = img2[p], d = x - y;
      const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      dL_dimg1[p] = gs * (A + 2.f * x * B + y * Cc) + gl * sgn;
    }
  }
}

// value = offset + c_ssim * sum partial[.][0] + c_l1 * sum partial[.][1], summed in double by one workgroup: the fused loss's scalar
// without the four small launches (reduce, dot, add, cast) the tensor library spends on it
__global__ __launch_bounds__(1024) void loss_combine_kernel(const float* __restrict__ partial, long long n, double c_ssim, double c_l1,
                                                            double offset, float* __restrict__ out) {
  __shared__ double red[2][16];
  double a = 0.0, b = 0.0;
  for (long long i = threadIdx.x; i < n; i += 1024) { a += (double)partial[2 * i]; b += (double)partial[2 * i + 1]; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int w = 0; w < 16; w++) { sa += red[0][w]; sb += red[1][w]; }
    out[0] = (float)(offset + c_ssim * sa + c_l1 * sb);
  }
}

int launch_loss_combine(const float* partial, long long n, double c_ssim, double c_l1, double offset, float* out, hipStream_t s) {
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(1024), 0, s, partial, n, c_ssim, c_l1, offset, out);
  GM_HIP(hipGetLastError());
  return 0;
}

int launch_ssim_fwd(const float* img1, const float* img2, int planes, int H, int W, float* d_mu1, float* d_e11, float* d_e12,
                    float* partial, hipStream_t s) {
  StageScope sc(ST_LOSS, s);
  const dim3 grid((W + LS_TILE - 1) / LS_TILE, (H + LS_TILE - 1) / LS_TILE, planes);
  const LossWindow win = make_window();
  if (d_mu1)
    hipLaunchKernelGGL(ssim_fwd_kernel<true>, grid, dim3(LS_THREADS), 0, s, img1, img2, H, W, win, d_mu1, d_e11, d_e12, partial);
  else
    hipLaunchKernelGGL(ssim_fwd_kernel<false>, grid, dim3(LS_THREADS), 0, s, img1, img2, H, W, win, d_mu1, d_e11, d_e12, partial);
  GM_HIP(hipGetLastError());
  return 0;
}

int launch_ssim_bwd(const float* img1, const float* img2, const float* d_mu1, const float* d_e11, const float* d_e12, int planes,
                    int H, int W, const float* g_ssim, const float* g_l1, float* dL_dimg1, hipStream_t s) {
  StageScope sc(ST_LOSS_BWD, s);
  const dim3 grid((W + LS_TILE - 1) / LS_TILE, (H + LS_TILE - 1) / LS_TILE, planes);
  hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(LS_THREADS), 0, s, img1, img2, d_mu1, d_e11, d_e12, H, W, make_window(), g_ssim, g_l1,
                     dL_dimg1);
  GM_HIP(hipGetLastError());
  return 0;
}

}  // namespace gm
