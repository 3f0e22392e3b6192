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
  __shared__ float sx[LS_SPAN][LS_SPAN + 1], sy[LS_SPAN][LS_SPAN + 1];
  __shared__ float hb[5][LS_SPAN][LS_TILE + 1];
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int ox = blockIdx.x * LS_TILE, oy = blockIdx.y * LS_TILE;
  const size_t plane = (size_t)blockIdx.z * H * W;
  for (int i = tid; i < LS_SPAN * LS_SPAN; i += LS_THREADS) {
    const int r = i / LS_SPAN, c = i - r * LS_SPAN;
    const int gx = ox + c - LS_HALO, gy = oy + r - LS_HALO;
    const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;          // zero padding (conv2d padding = 5)
    const size_t p = plane + (size_t)(in ? gy : 0) * W + (in ? gx : 0);
    sx[r][c] = in ? img1[p] : 0.f;
    sy[r][c] = in ? img2[p] : 0.f;
  }
  __syncthreads();
  // horizontal taps: one work item = 4 adjacent output columns of one row (14 staged values feed 4 x 11 taps);
  // consecutive lanes take consecutive rows (row stride 43 words: conflict-free)
  for (int i = tid; i < LS_SPAN * (LS_TILE / 4); i += LS_THREADS) {
    const int r = i % LS_SPAN, c0 = (i / LS_SPAN) * 4;
    float xv[14], yv[14], xx[14], yy[14], xy[14];
#pragma unroll
    for (int k = 0; k < 14; k++) {
      xv[k] = sx[r][c0 + k]; yv[k] = sy[r][c0 + k];
      xx[k] = xv[k] * xv[k]; yy[k] = yv[k] * yv[k]; xy[k] = xv[k] * yv[k];
    }
#pragma unroll
    for (int o = 0; o < 4; o++) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) {
        const float w = win.w[k];
        a0 += w * xv[o + k]; a1 += w * yv[o + k]; a2 += w * xx[o + k]; a3 += w * yy[o + k]; a4 += w * xy[o + k];
      }
      hb[0][r][c0 + o] = a0; hb[1][r][c0 + o] = a1; hb[2][r][c0 + o] = a2; hb[3][r][c0 + o] = a3; hb[4][r][c0 + o] = a4;
    }
  }
  __syncthreads();
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
  const int c = tid & 31;
  float s_sum = 0.f, l1_sum = 0.f;
  // vertical taps: a thread owns 4 adjacent rows of one column (14 values per quantity feed 4 x 11 taps)
  float vq[5][4];
#pragma unroll
  for (int q = 0; q < 5; q++) {
    float col[14];
#pragma unroll
    for (int k = 0; k < 14; k++) col[k] = hb[q][(tid >> 5) * 4 + k][c];
#pragma unroll
    for (int o = 0; o < 4; o++) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) acc += win.w[k] * col[o + k];
      vq[q][o] = acc;
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
      l1_sum += fabsf(sx[r + LS_HALO][c + LS_HALO] - sy[r + LS_HALO][c + LS_HALO]);
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
  __shared__ float sm[3][LS_SPAN][LS_SPAN + 1];
  __shared__ float hb[3][LS_SPAN][LS_TILE + 1];
  const int tid = threadIdx.x;
  const int ox = blockIdx.x * LS_TILE, oy = blockIdx.y * LS_TILE;
  const size_t plane = (size_t)blockIdx.z * H * W;
  for (int i = tid; i < LS_SPAN * LS_SPAN; i += LS_THREADS) {
    const int r = i / LS_SPAN, c = i - r * LS_SPAN;
    const int gx = ox + c - LS_HALO, gy = oy + r - LS_HALO;
    const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;          // no ssim-map pixel outside the image
    const size_t p = plane + (size_t)(in ? gy : 0) * W + (in ? gx : 0);
    sm[0][r][c] = in ? d_mu1[p] : 0.f;
    sm[1][r][c] = in ? d_e11[p] : 0.f;
    sm[2][r][c] = in ? d_e12[p] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < 3 * LS_SPAN * (LS_TILE / 4); i += LS_THREADS) {      // see ssim_fwd_kernel
    const int r = i % LS_SPAN, rest = i / LS_SPAN, c0 = (rest & 7) * 4, q = rest >> 3;
    float v[14];
#pragma unroll
    for (int k = 0; k < 14; k++) v[k] = sm[q][r][c0 + k];
#pragma unroll
    for (int o = 0; o < 4; o++) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) acc += win.w[k] * v[o + k];
      hb[q][r][c0 + o] = acc;
    }
  }
  __syncthreads();
  const float gs = g_ssim[blockIdx.z];
  const float gl = g_l1 ? g_l1[0] : 0.f;
  const int c = tid & 31;
  float vq[3][4];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    float col[14];
#pragma unroll
    for (int k = 0; k < 14; k++) col[k] = hb[q][(tid >> 5) * 4 + k][c];
#pragma unroll
    for (int o = 0; o < 4; o++) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) acc += win.w[k] * col[o + k];
      vq[q][o] = acc;
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
