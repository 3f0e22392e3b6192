// gm_train.hip -- the per-Gaussian host-framework work around the rasterizer in one training iteration, fused:
//
//   mesh_activate_fwd / _bwd : MeshBasedGaussianModel's parameter -> rasterizer-input map and its adjoint
//       get_xyz      = softmax(bc) . (v1,v2,v3) + alpha r (sigmoid(d) - 0.5) n      scene/mesh_based_gaussian_model.py:138-152
//       get_scaling  = exp(_scaling)            get_rotation = normalize(_rotation)                   :122-128, 33-43
//       get_opacity  = sigmoid(_opacity)                                                              :172-174
//     In the reference these are ~15 Jittor elementwise ops forward and ~25 in the autograd pass, each a pass over P;
//     here one kernel each way (60 B in, 44 B out per Gaussian forward).
//   adam_kernel : jittor.nn.Adam's update for all parameter groups of the model in ONE launch (table of tensors,
//     per-group learning rate; the SH tensor [P,16,3] takes two rates - coefficient 0 is the reference's "f_dc" group,
//     the rest "f_rest" - so no concatenation / split of the 192-byte SH rows is needed per iteration):
//       m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v) + eps)
//     (scene/mesh_based_gaussian_model.py:242-263 training_setup; jittor/optim.py Adam.step).
//     `active` (gm_adam_step_active): while the model's active SH degree D is below 3 (train_mesh_gaussian.py:70-71: one degree per
//     1000 iterations, starting at 0) the coefficients >= (D+1)^2 of every row have never had a non-zero gradient: g = m = v = 0,
//     and the rule above leaves p, m and v exactly as they are.  The kernel then does not touch them at all - at D = 0 that is
//     45 of the 60 parameters of a Gaussian, 28 bytes each.
#include "gm_common.h"

namespace gm {

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + __expf(-x)); }

// sqrt(|AB x AC|) of the bound face: the reference's circumradius() (utils/loss_utils.py:86-101)
__device__ __forceinline__ float face_radius(const ActArgs& a, size_t i3) {
  const float ax = a.v2[i3] - a.v1[i3], ay = a.v2[i3 + 1] - a.v1[i3 + 1], az = a.v2[i3 + 2] - a.v1[i3 + 2];
  const float bx = a.v3[i3] - a.v1[i3], by = a.v3[i3 + 1] - a.v1[i3 + 1], bz = a.v3[i3 + 2] - a.v1[i3 + 2];
  const float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
  return sqrtf(sqrtf(cx * cx + cy * cy + cz * cz));
}

// mr_partial (may be null): per-workgroup sums of the mesh-restrict term max(0, max_c scales - mr_weight * face_radius)
// (utils/loss_utils.py:103-108); the host side adds the partials.
__global__ __launch_bounds__(256) void mesh_activate_fwd_kernel(const ActArgs a, float* __restrict__ xyz, float* __restrict__ scales,
                                                                 float4* __restrict__ rots, float* __restrict__ opac,
                                                                 float mr_weight, float* __restrict__ mr_partial) {
  __shared__ float red[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float term = 0.f;
  if (i < a.N) {
  const size_t i3 = 3 * (size_t)i;
  const float b0 = a.bc[i3], b1 = a.bc[i3 + 1], b2 = a.bc[i3 + 2];
  const float mx = fmaxf(b0, fmaxf(b1, b2));
  const float e0 = __expf(b0 - mx), e1 = __expf(b1 - mx), e2 = __expf(b2 - mx);
  const float inv = 1.0f / (e0 + e1 + e2);
  const float w0 = e0 * inv, w1 = e1 * inv, w2 = e2 * inv;
  const float k = a.alpha * a.r[i] * (sigmoidf(a.dist[i]) - 0.5f);
#pragma unroll
  for (int c = 0; c < 3; c++)
    xyz[i3 + c] = (w0 * a.v1[i3 + c] + w1 * a.v2[i3 + c] + w2 * a.v3[i3 + c]) + k * a.normal[i3 + c];
  float smax = -3.4e38f;
#pragma unroll
  for (int c = 0; c < 3; c++) { const float sc = __expf(a.scaling[i3 + c]); scales[i3 + c] = sc; smax = fmaxf(smax, sc); }
  const float4 q = reinterpret_cast<const float4*>(a.rotation)[i];
  const float qn = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);   // F.normalize eps
  rots[i] = make_float4(q.x * qn, q.y * qn, q.z * qn, q.w * qn);
  opac[i] = sigmoidf(a.opacity[i]);
  if (mr_partial) term = fmaxf(smax - mr_weight * face_radius(a, i3), 0.f);
  }
  if (mr_partial) {                              // wave-uniform
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) term += __shfl_xor(term, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = term;
    __syncthreads();
    if (threadIdx.x == 0) mr_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

__global__ __launch_bounds__(256) void mesh_activate_bwd_kernel(const ActArgs a, const float* __restrict__ d_xyz,
                                                                 const float* __restrict__ d_scales, const float4* __restrict__ d_rots,
                                                                 const float* __restrict__ d_opac, float* __restrict__ d_bc,
                                                                 float* __restrict__ d_dist, float* __restrict__ d_scaling,
                                                                 float4* __restrict__ d_rotation, float* __restrict__ d_opacity,
                                                                 float mr_weight, const float* __restrict__ d_mr /*[1] or null*/) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.N) return;
  const size_t i3 = 3 * (size_t)i;
  const float gx = d_xyz ? d_xyz[i3] : 0.f, gy = d_xyz ? d_xyz[i3 + 1] : 0.f, gz = d_xyz ? d_xyz[i3 + 2] : 0.f;
  {  // softmax-barycentric position and normal offset
    const float b0 = a.bc[i3], b1 = a.bc[i3 + 1], b2 = a.bc[i3 + 2];
    const float mx = fmaxf(b0, fmaxf(b1, b2));
    const float e0 = __expf(b0 - mx), e1 = __expf(b1 - mx), e2 = __expf(b2 - mx);
    const float inv = 1.0f / (e0 + e1 + e2);
    const float w0 = e0 * inv, w1 = e1 * inv, w2 = e2 * inv;
    const float dw0 = gx * a.v1[i3] + gy * a.v1[i3 + 1] + gz * a.v1[i3 + 2];
    const float dw1 = gx * a.v2[i3] + gy * a.v2[i3 + 1] + gz * a.v2[i3 + 2];
    const float dw2 = gx * a.v3[i3] + gy * a.v3[i3 + 1] + gz * a.v3[i3 + 2];
    const float dot = w0 * dw0 + w1 * dw1 + w2 * dw2;
    d_bc[i3] = w0 * (dw0 - dot); d_bc[i3 + 1] = w1 * (dw1 - dot); d_bc[i3 + 2] = w2 * (dw2 - dot);
    const float sd = sigmoidf(a.dist[i]);
    const float gn = gx * a.normal[i3] + gy * a.normal[i3 + 1] + gz * a.normal[i3 + 2];
    d_dist[i] = gn * a.alpha * a.r[i] * sd * (1.0f - sd);
  }
  {
    float sc[3], ds[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { sc[c] = __expf(a.scaling[i3 + c]); ds[c] = d_scales ? d_scales[i3 + c] : 0.f; }
    if (d_mr) {                                  // d/dscales of max(0, max_c scales - w R): one-hot on the first largest axis
      const int am = (sc[0] >= sc[1] && sc[0] >= sc[2]) ? 0 : (sc[1] >= sc[2] ? 1 : 2);
      if (sc[am] - mr_weight * face_radius(a, i3) > 0.f) ds[am] += d_mr[0];
    }
#pragma unroll
    for (int c = 0; c < 3; c++) d_scaling[i3 + c] = ds[c] * sc[c];
  }
  {
    const float4 q = reinterpret_cast<const float4*>(a.rotation)[i];
    const float nrm = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f), qn = 1.0f / nrm;
    const float4 y = make_float4(q.x * qn, q.y * qn, q.z * qn, q.w * qn);
    const float4 g = d_rots ? d_rots[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float yd = y.x * g.x + y.y * g.y + y.z * g.z + y.w * g.w;
    d_rotation[i] = make_float4((g.x - y.x * yd) * qn, (g.y - y.y * yd) * qn, (g.z - y.z * yd) * qn, (g.w - y.w * yd) * qn);
  }
  {
    const float o = sigmoidf(a.opacity[i]);
    d_opacity[i] = d_opac ? d_opac[i] * o * (1.0f - o) : 0.f;
  }
}

int launch_mesh_activate_fwd(const ActArgs& a, float* xyz, float* scales, float* rots, float* opac, float mr_weight, float* mr_partial,
                             hipStream_t s) {
  if (a.N <= 0) return 0;
  hipLaunchKernelGGL(mesh_activate_fwd_kernel, dim3((a.N + 255) / 256), dim3(256), 0, s, a, xyz, scales, reinterpret_cast<float4*>(rots), opac,
                     mr_weight, mr_partial);
  GM_HIP(hipGetLastError());
  return 0;
}

int launch_mesh_activate_bwd(const ActArgs& a, const float* d_xyz, const float* d_scales, const float* d_rots, const float* d_opac,
                             float* d_bc, float* d_dist, float* d_scaling, float* d_rotation, float* d_opacity, float mr_weight,
                             const float* d_mr, hipStream_t s) {
  if (a.N <= 0) return 0;
  hipLaunchKernelGGL(mesh_activate_bwd_kernel, dim3((a.N + 255) / 256), dim3(256), 0, s, a, d_xyz, d_scales,
                     reinterpret_cast<const float4*>(d_rots), d_opac, d_bc, d_dist, d_scaling, reinterpret_cast<float4*>(d_rotation), d_opacity,
                     mr_weight, d_mr);
  GM_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(const AdamTable tab) {
  const AdamTensor t = tab.t[blockIdx.y];
  const float b1 = tab.b1, b2 = tab.b2, eps = tab.eps, c1 = tab.omb1, c2 = tab.omb2;
  const unsigned long long n4 = t.n >> 2;
  if (t.active) {
    // only the first ga granules of every period: thread <-> (period index, live granule)
    const unsigned gp = t.period >> 2, ga = (t.active + 3u) >> 2;
    const unsigned long long rows = n4 / gp, live = rows * ga;
    for (unsigned long long w = (unsigned long long)blockIdx.x * 256 + threadIdx.x; w < live; w += (unsigned long long)gridDim.x * 256) {
      const unsigned long long row = w / ga;
      const unsigned k = (unsigned)(w - row * ga);
      const unsigned long long q = row * gp + k;
      const float4 g = reinterpret_cast<const float4*>(t.g)[q];
      float4 p = reinterpret_cast<float4*>(t.p)[q], m = reinterpret_cast<float4*>(t.m)[q], v = reinterpret_cast<float4*>(t.v)[q];
      float* pp = &p.x; float* mm = &m.x; float* vv = &v.x; const float* gg = &g.x;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const float st = (4u * k + c >= t.split) ? t.step_hi : t.step_lo;
        adam_update(pp[c], mm[c], vv[c], gg[c], b1, b2, c1, c2, eps, st);
      }
      reinterpret_cast<float4*>(t.p)[q] = p; reinterpret_cast<float4*>(t.m)[q] = m; reinterpret_cast<float4*>(t.v)[q] = v;
    }
    return;                                        // (API: n is a multiple of the period when `active` is given)
  }
  for (unsigned long long q = (unsigned long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += (unsigned long long)gridDim.x * 256) {
    const float4 g = reinterpret_cast<const float4*>(t.g)[q];
    float4 p = reinterpret_cast<float4*>(t.p)[q], m = reinterpret_cast<float4*>(t.m)[q], v = reinterpret_cast<float4*>(t.v)[q];
    float* pp = &p.x; float* mm = &m.x; float* vv = &v.x; const float* gg = &g.x;
    const unsigned in_period = t.period ? (unsigned)(q % (t.period >> 2)) * 4u : 0u;     // period is a multiple of 4 (checked by the API)
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const float st = (t.period && in_period + c >= t.split) ? t.step_hi : t.step_lo;
      adam_update(pp[c], mm[c], vv[c], gg[c], b1, b2, c1, c2, eps, st);
    }
    reinterpret_cast<float4*>(t.p)[q] = p; reinterpret_cast<float4*>(t.m)[q] = m; reinterpret_cast<float4*>(t.v)[q] = v;
  }
  // tail (n not a multiple of 4): first workgroup
  if (blockIdx.x == 0 && threadIdx.x < (t.n & 3)) {
    const unsigned long long e = (n4 << 2) + threadIdx.x;
    const float st = (t.period && (unsigned)(e % t.period) >= t.split) ? t.step_hi : t.step_lo;
    const float g = t.g[e];
    float m = t.m[e], v = t.v[e], pe = t.p[e];
    adam_update(pe, m, v, g, b1, b2, c1, c2, eps, st);
    t.m[e] = m; t.v[e] = v; t.p[e] = pe;
  }
}

int launch_adam(const AdamTable& tab, hipStream_t s) {
  if (tab.count <= 0) return 0;
  unsigned long long mx = 0;
  for (int i = 0; i < tab.count; i++) mx = tab.t[i].n > mx ? tab.t[i].n : mx;
  unsigned long long nb = (mx / 4 + 255) / 256;
  if (nb < 1) nb = 1;
  if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb, (unsigned)tab.count), dim3(256), 0, s, tab);
  GM_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Densification statistics of one training iteration (train_mesh_gaussian.py:119-126) in one pass:
//   max_radii2D[vis] = max(max_radii2D[vis], radii[vis])                                            (:123)
//   bc_gradient_accum[vis] += |viewspace_grad[vis, :2]| ;  denom[vis] += 1       (mesh_based_gaussian_model.py:587-589)
// with vis = radii > 0 (render()'s visibility_filter); the reference spends three masked gather / scatter ops on each.
__global__ __launch_bounds__(256) void densify_stats_kernel(int N, const int* __restrict__ radii, const float* __restrict__ grad2d,
                                                            float* __restrict__ max_radii2D, float* __restrict__ grad_accum,
                                                            float* __restrict__ denom) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int r = radii[i];
  if (r <= 0) return;
  max_radii2D[i] = fmaxf(max_radii2D[i], (float)r);
  const float gx = grad2d[3 * (size_t)i], gy = grad2d[3 * (size_t)i + 1];
  grad_accum[i] += sqrtf(gx * gx + gy * gy);
  denom[i] += 1.0f;
}

int launch_densify_stats(int N, const int* radii, const float* grad2d, float* max_radii2D, float* grad_accum, float* denom, hipStream_t s) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(densify_stats_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, radii, grad2d, max_radii2D, grad_accum, denom);
  GM_HIP(hipGetLastError());
  return 0;
}

}  // namespace gm
