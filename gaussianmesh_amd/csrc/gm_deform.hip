// gm_deform.hip -- mesh-driven deformation of bound Gaussians and the edit tool's view-dependent colour.
//
// Replaces the Jittor tensor algebra of the reference (about ten separate elementwise/gather/matmul ops that
// materialise [N,3,3,3] gathers):
//   edittool/__init__.py:116-131  SingleObjectDeform.deform_gaussian
//   edittool/__init__.py:442-448  ObjectVisualTool.render_gaussian colour path (dir rotation + eval_sh + clamp)
//   edittool/general_utils.py:26-37 strip_symmetric (fused as the optional cov6 output)
// One thread per Gaussian; the per-vertex tables (dV, R, S: 84 B x Vm, ~0.6 MB for a 7.5k-vertex proxy mesh)
// stay L2-resident, the per-Gaussian streams (72 B in, 84-108 B out) are the HBM traffic.
#include "gm_common.h"
#pragma clang fp contract(off)
#include "gm_sh.h"

namespace gm {

__global__ __launch_bounds__(256) void deform_kernel(int N, const int* __restrict__ tri, const float* __restrict__ w,
                                                     const float* __restrict__ dV, const float* __restrict__ Rv,
                                                     const float* __restrict__ Sv, const float* __restrict__ cov,
                                                     const float* __restrict__ pos, float* __restrict__ pos_out,
                                                     float* __restrict__ cov_out, float* __restrict__ rot_out,
                                                     float* __restrict__ cov6_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int t0 = tri[3 * (size_t)i], t1 = tri[3 * (size_t)i + 1], t2 = tri[3 * (size_t)i + 2];
  const float w0 = w[3 * (size_t)i], w1 = w[3 * (size_t)i + 1], w2 = w[3 * (size_t)i + 2];
  float d[3], Rb[9], Sb[9];
#pragma unroll
  for (int k = 0; k < 3; k++) d[k] = (w0 * dV[3 * (size_t)t0 + k] + w1 * dV[3 * (size_t)t1 + k]) + w2 * dV[3 * (size_t)t2 + k];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    Rb[k] = (w0 * Rv[9 * (size_t)t0 + k] + w1 * Rv[9 * (size_t)t1 + k]) + w2 * Rv[9 * (size_t)t2 + k];
    Sb[k] = (w0 * Sv[9 * (size_t)t0 + k] + w1 * Sv[9 * (size_t)t1 + k]) + w2 * Sv[9 * (size_t)t2 + k];
  }
  float Rt[9], RS[9], A[9], C[9], O[9];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) Rt[3 * a + b] = Rb[3 * b + a];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) RS[3 * a + b] = (Rt[3 * a] * Sb[b] + Rt[3 * a + 1] * Sb[3 + b]) + Rt[3 * a + 2] * Sb[6 + b];
#pragma unroll
  for (int k = 0; k < 9; k++) C[k] = cov[9 * (size_t)i + k];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) A[3 * a + b] = (RS[3 * a] * C[b] + RS[3 * a + 1] * C[3 + b]) + RS[3 * a + 2] * C[6 + b];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) O[3 * a + b] = (A[3 * a] * RS[3 * b] + A[3 * a + 1] * RS[3 * b + 1]) + A[3 * a + 2] * RS[3 * b + 2];
#pragma unroll
  for (int k = 0; k < 9; k++) { cov_out[9 * (size_t)i + k] = O[k]; rot_out[9 * (size_t)i + k] = Rt[k]; }
#pragma unroll
  for (int k = 0; k < 3; k++) pos_out[3 * (size_t)i + k] = pos[3 * (size_t)i + k] + d[k];
  if (cov6_out) {
    float* o = cov6_out + 6 * (size_t)i;
    o[0] = O[0]; o[1] = O[1]; o[2] = O[2]; o[3] = O[4]; o[4] = O[5]; o[5] = O[8];
  }
}

int launch_deform(int N, const int* tri, const float* w, const float* dV, const float* Rv, const float* Sv,
                  const float* cov, const float* pos, float* pos_out, float* cov_out, float* rot_out, float* cov6_out,
                  hipStream_t s) {
  StageScope sc(ST_DEFORM, s);
  if (N > 0)
    hipLaunchKernelGGL(deform_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, tri, w, dV, Rv, Sv, cov, pos, pos_out, cov_out,
                       rot_out, cov6_out);
  GM_HIP(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void sh_colors_kernel(int N, int deg, int M, const float* __restrict__ pos,
                                                        const float* __restrict__ campos, const float* __restrict__ rot,
                                                        const float* __restrict__ shs, float* __restrict__ rgb) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float dx = pos[3 * (size_t)i] - campos[0], dy = pos[3 * (size_t)i + 1] - campos[1], dz = pos[3 * (size_t)i + 2] - campos[2];
  const float len = sqrtf(dx * dx + dy * dy + dz * dz);
  dx = dx / len; dy = dy / len; dz = dz / len;
  float x = dx, y = dy, z = dz;
  if (rot) {   // dir_rot = rot^T dir
    const float* R = rot + 9 * (size_t)i;
    x = (R[0] * dx + R[3] * dy) + R[6] * dz;
    y = (R[1] * dx + R[4] * dy) + R[7] * dz;
    z = (R[2] * dx + R[5] * dy) + R[8] * dz;
  }
  float sh[48];
  load_sh(shs, i, M, (deg + 1) * (deg + 1), sh);
#pragma unroll
  for (int ch = 0; ch < 3; ch++) {
    const float r = sh_channel(deg, [&](int k) { return sh[3 * k + ch]; }, x, y, z);
    rgb[3 * (size_t)i + ch] = fmaxf(r + 0.5f, 0.0f);
  }
}

int launch_sh_colors(int N, int deg, int M, const float* pos, const float* campos, const float* rot,
                     const float* shs, float* rgb, hipStream_t s) {
  StageScope sc(ST_SH_COLORS, s);
  if (N > 0) hipLaunchKernelGGL(sh_colors_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, deg, M, pos, campos, rot, shs, rgb);
  GM_HIP(hipGetLastError());
  return 0;
}

}  // namespace gm
