// gm_mesh.hip -- per-vertex rotation / stretch of a deformed proxy mesh: the (R, S) pair SingleObjectDeform.deform_gaussian
// feeds into the Gaussian deformation.
//
// Replaces pyACAP.GetRS(rest_vertices, deformed_vertices, ...) at edittool/__init__.py:102, 109.  pyACAP (ACAP/pyACAPv1.zip,
// no version pin) is a binary that is NOT in the reference tree (.MISSING_LARGE_BLOBS) and has no test vectors: "parity
// unpinned".  Restated from its published algorithm (Gao, Lai, Yang, Rosin, Xia, "Sparse Data Driven Mesh Deformation":
// the ACAP feature's first step) and anchored on the call site:
//   per vertex i the affine map of its one-ring, T_i = argmin sum_{j in N(i)} c_ij |(p'_i - p'_j) - T (p_i - p_j)|^2 with
//   cotangent weights c_ij, i.e. T_i = (sum c e' e^T)(sum c e e^T)^-1, then the polar decomposition T_i = Q_i S_i
//   (Q proper rotation, S symmetric).
// Conditioning: negative cotangents are clamped to a small positive weight, and both sums get a tiny (1e-9 of the trace)
// term along the vertex normal - n n^T on the rest side, sqrt(area' / area) n' n^T on the deformed side - so that a planar
// one-ring still yields a full-rank map (its normal direction then follows the deformed normal, scaled like a length).
// For any affine deformation of a non-planar neighbourhood T_i is that affine map exactly: identity -> (I, I), rigid motion
// -> (rotation, I), uniform scale s -> (I, s I).
// Outputs, per vertex, row-major 3x3: S, and R = Q^T - the transpose is pyACAP's row-vector convention as the call site uses
// it: deform_gaussian takes gaussian_deform_rot = blend(R)^T and transforms covariances by (R^T S) C (R^T S)^T
// (edittool/__init__.py:118-129), which is T C T^T exactly when R^T S = T.
//
// One thread per vertex over a CSR vertex -> face adjacency built once per mesh on the host (deterministic, no atomics);
// double precision (7.5 k vertices: the kernel is a few microseconds either way).
#include "gm_common.h"

namespace gm {

__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

// eigen-decomposition of a symmetric 3x3 (cyclic Jacobi): A -> diagonal in place, V columns = eigenvectors
__device__ __forceinline__ void jacobi3(double A[3][3], double V[3][3]) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) V[i][j] = i == j ? 1.0 : 0.0;
  const double scale = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]) + 1e-300;
  for (int sweep = 0; sweep < 10; sweep++) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    if (off <= 1e-16 * scale) break;
#pragma unroll
    for (int pq = 0; pq < 3; pq++) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      const double apq = A[p][q];
      if (fabs(apq) <= 1e-300) continue;
      const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
      const int r = 3 - p - q;
      const double arp = A[r][p], arq = A[r][q];
      A[p][p] -= t * apq; A[q][q] += t * apq; A[p][q] = A[q][p] = 0.0;
      A[r][p] = A[p][r] = cs * arp - sn * arq;
      A[r][q] = A[q][r] = sn * arp + cs * arq;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const double vp = V[k][p], vq = V[k][q];
        V[k][p] = cs * vp - sn * vq; V[k][q] = sn * vp + cs * vq;
      }
    }
  }
}

#ifndef GM_MESH_THREADS
#define GM_MESH_THREADS 64       // one-wave workgroups: 118 of them for the 7.5 k-vertex proxy mesh instead of 30 four-wave ones (pipelined loop +0.8 %)
#endif
struct MeshFrames { int frames; const float* V1[GM_BATCH_MAX]; float4* packed[GM_BATCH_MAX]; };

__global__ __launch_bounds__(GM_MESH_THREADS) void mesh_rs_kernel(int Vm, const float* __restrict__ V0, const float* __restrict__ V1,
                                                      const int* __restrict__ faces, const int* __restrict__ adj_offsets,
                                                      const int* __restrict__ adj_faces, float* __restrict__ R_out,
                                                      float* __restrict__ S_out, float* __restrict__ state_out,
                                                      float4* __restrict__ packed_out, const MeshFrames mf) {
  if (mf.frames > 1) { V1 = mf.V1[blockIdx.z]; packed_out = mf.packed[blockIdx.z]; }      // frame blockIdx.z of a batch: its deformed mesh, its table
  const int v = blockIdx.x * GM_MESH_THREADS + threadIdx.x;
  if (v >= Vm) return;
  // M0 = sum c e e^T (symmetric), M1 = sum c e' e^T over the one-ring edges; every incident face contributes its two edges
  // at v with half the cotangent of the opposite angle (an interior edge gets both halves from its two faces)
  double M0[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, M1[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double nr[3] = {0, 0, 0}, nd[3] = {0, 0, 0}, wsum = 0.0;          // area-weighted normals (rest / deformed)
  // The one-ring is read in chunks of RING faces with all loads of a level issued together (face ids -> corner ids ->
  // positions: three dependent round trips per chunk instead of three per face; 7.5 k threads cannot hide latency otherwise).
  constexpr int RING = 8;
  const int k_begin = adj_offsets[v], k_end = adj_offsets[v + 1];
  float p0[3], p1[3];
#pragma unroll
  for (int j = 0; j < 3; j++) { p0[j] = V0[3 * (size_t)v + j]; p1[j] = V1[3 * (size_t)v + j]; }
  for (int kb = k_begin; kb < k_end; kb += RING) {
    int fid[RING], ia_[RING], ib_[RING];
#pragma unroll
    for (int r = 0; r < RING; r++) fid[r] = adj_faces[min(kb + r, k_end - 1)];
#pragma unroll
    for (int r = 0; r < RING; r++) {
      const int i0 = faces[3 * fid[r]], i1 = faces[3 * fid[r] + 1], i2 = faces[3 * fid[r] + 2];
      ia_[r] = i0 == v ? i1 : (i1 == v ? i2 : i0);                  // the corner after v, the corner before v (orientation kept)
      ib_[r] = i0 == v ? i2 : (i1 == v ? i0 : i1);
    }
    float qa0[RING][3], qb0[RING][3], qa1[RING][3], qb1[RING][3];
#pragma unroll
    for (int r = 0; r < RING; r++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        qa0[r][j] = V0[3 * (size_t)ia_[r] + j]; qb0[r][j] = V0[3 * (size_t)ib_[r] + j];
        qa1[r][j] = V1[3 * (size_t)ia_[r] + j]; qb1[r][j] = V1[3 * (size_t)ib_[r] + j];
      }
#pragma unroll
    for (int r = 0; r < RING; r++) {
    if (kb + r >= k_end) continue;
    double ea[3], eb[3], da[3], db[3], ab[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      ea[j] = (double)qa0[r][j] - p0[j]; eb[j] = (double)qb0[r][j] - p0[j];
      da[j] = (double)qa1[r][j] - p1[j]; db[j] = (double)qb1[r][j] - p1[j];
      ab[j] = eb[j] - ea[j];
    }
    double n0[3], n1[3];
    cross3(ea, eb, n0); cross3(da, db, n1);
    const double l0 = sqrt(n0[0] * n0[0] + n0[1] * n0[1] + n0[2] * n0[2]);
    if (!(l0 > 1e-30)) continue;                                     // degenerate rest face
    // cot of the angle at a (opposite edge v-b) and at b (opposite edge v-a): cot = (u . w) / |u x w|, |u x w| = l0 for all corners
    const double cot_a = -(ea[0] * ab[0] + ea[1] * ab[1] + ea[2] * ab[2]) / l0;     // angle at a between (v - a) = -ea and (b - a) = ab
    const double cot_b = (eb[0] * ab[0] + eb[1] * ab[1] + eb[2] * ab[2]) / l0;      // angle at b between (v - b) = -eb and (a - b) = -ab
    const double wa = fmax(0.5 * cot_b, 1e-3), wb = fmax(0.5 * cot_a, 1e-3);        // weight of edge v-a / v-b
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        M0[i][j] += wa * ea[i] * ea[j] + wb * eb[i] * eb[j];
        M1[i][j] += wa * da[i] * ea[j] + wb * db[i] * eb[j];
      }
#pragma unroll
    for (int j = 0; j < 3; j++) { nr[j] += n0[j]; nd[j] += n1[j]; }
    wsum += l0;
    }
  }
  double F[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  bool have = false;
  if (wsum > 0.0) {
    const double lr = sqrt(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]), ld = sqrt(nd[0] * nd[0] + nd[1] * nd[1] + nd[2] * nd[2]);
    const double lam = 1e-9 * (M0[0][0] + M0[1][1] + M0[2][2]);
    if (lr > 1e-30 && ld > 1e-30) {
      const double sc = sqrt(ld / lr);                               // lengths scale like the square root of the area ratio
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
          M0[i][j] += lam * (nr[i] / lr) * (nr[j] / lr);
          M1[i][j] += lam * sc * (nd[i] / ld) * (nr[j] / lr);
        }
    }
    // F = M1 M0^-1 (M0 symmetric positive definite after the regularisation): adjugate / determinant
    double A[3][3];
    A[0][0] = M0[1][1] * M0[2][2] - M0[1][2] * M0[2][1]; A[0][1] = M0[0][2] * M0[2][1] - M0[0][1] * M0[2][2]; A[0][2] = M0[0][1] * M0[1][2] - M0[0][2] * M0[1][1];
    A[1][0] = M0[1][2] * M0[2][0] - M0[1][0] * M0[2][2]; A[1][1] = M0[0][0] * M0[2][2] - M0[0][2] * M0[2][0]; A[1][2] = M0[0][2] * M0[1][0] - M0[0][0] * M0[1][2];
    A[2][0] = M0[1][0] * M0[2][1] - M0[1][1] * M0[2][0]; A[2][1] = M0[0][1] * M0[2][0] - M0[0][0] * M0[2][1]; A[2][2] = M0[0][0] * M0[1][1] - M0[0][1] * M0[1][0];
    const double det0 = M0[0][0] * A[0][0] + M0[0][1] * A[1][0] + M0[0][2] * A[2][0];
    if (fabs(det0) > 1e-300) {
      have = true;
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) F[i][j] = (M1[i][0] * A[0][j] + M1[i][1] * A[1][j] + M1[i][2] * A[2][j]) / det0;
    }
  }
  double Q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, S[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  // Common case - a proper, well-conditioned map (det F > 0, no direction collapsed by more than ~30x relative to the
  // others): the orthogonal polar factor by Newton's iteration X <- (X + X^-T) / 2 (quadratic convergence, five or six
  // steps of one cofactor matrix and one division each - a fifth of the Jacobi route's divisions and square roots).
  // Reflections and near-singular maps take the eigen-decomposition route below, which fixes their conventions.
  if (have) {
    const double detF = F[0][0] * (F[1][1] * F[2][2] - F[1][2] * F[2][1]) - F[0][1] * (F[1][0] * F[2][2] - F[1][2] * F[2][0]) +
                        F[0][2] * (F[1][0] * F[2][1] - F[1][1] * F[2][0]);
    double fro = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) fro += F[i][j] * F[i][j];
    if (detF > 0.0 && 27.0 * detF * detF > 1e-6 * fro * fro * fro) {
      double X[3][3];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) X[i][j] = F[i][j];
      bool converged = false;
      for (int it = 0; it < 24 && !converged; it++) {
        double Cf[3][3];                                            // cofactor matrix = det(X) X^-T
        Cf[0][0] = X[1][1] * X[2][2] - X[1][2] * X[2][1]; Cf[0][1] = X[1][2] * X[2][0] - X[1][0] * X[2][2]; Cf[0][2] = X[1][0] * X[2][1] - X[1][1] * X[2][0];
        Cf[1][0] = X[0][2] * X[2][1] - X[0][1] * X[2][2]; Cf[1][1] = X[0][0] * X[2][2] - X[0][2] * X[2][0]; Cf[1][2] = X[0][1] * X[2][0] - X[0][0] * X[2][1];
        Cf[2][0] = X[0][1] * X[1][2] - X[0][2] * X[1][1]; Cf[2][1] = X[0][2] * X[1][0] - X[0][0] * X[1][2]; Cf[2][2] = X[0][0] * X[1][1] - X[0][1] * X[1][0];
        const double d = X[0][0] * Cf[0][0] + X[0][1] * Cf[0][1] + X[0][2] * Cf[0][2];
        const double hinv = 0.5 / d;
        double delta = 0.0, mag = 0.0;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) {
            const double y = 0.5 * X[i][j] + hinv * Cf[i][j];
            delta = fmax(delta, fabs(y - X[i][j])); mag = fmax(mag, fabs(y));
            X[i][j] = y;
          }
        converged = delta <= 1e-14 * mag;
      }
      if (converged) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) Q[i][j] = X[i][j];
        double M[3][3];                                             // Q^T F, symmetric up to rounding
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) M[i][j] = X[0][i] * F[0][j] + X[1][i] * F[1][j] + X[2][i] * F[2][j];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) S[i][j] = 0.5 * (M[i][j] + M[j][i]);
        have = false;                                               // done
      }
    }
  }
  if (have) {
    double C[3][3], E[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) C[i][j] = F[0][i] * F[0][j] + F[1][i] * F[1][j] + F[2][i] * F[2][j];   // F^T F
    jacobi3(C, E);
    double sig[3] = {sqrt(fmax(C[0][0], 0.0)), sqrt(fmax(C[1][1], 0.0)), sqrt(fmax(C[2][2], 0.0))};
    const double det = F[0][0] * (F[1][1] * F[2][2] - F[1][2] * F[2][1]) - F[0][1] * (F[1][0] * F[2][2] - F[1][2] * F[2][0]) +
                       F[0][2] * (F[1][0] * F[2][1] - F[1][1] * F[2][0]);
    if (det < 0.0) {                                               // reflection: the smallest stretch takes the sign
      const int m = (sig[0] <= sig[1] && sig[0] <= sig[2]) ? 0 : (sig[1] <= sig[2] ? 1 : 2);
      sig[m] = -sig[m];
    }
    const double big = fmax(fabs(sig[0]), fmax(fabs(sig[1]), fabs(sig[2])));
    double inv[3];
#pragma unroll
    for (int k = 0; k < 3; k++) inv[k] = fabs(sig[k]) > 1e-12 * big && big > 0.0 ? 1.0 / sig[k] : 0.0;
    double FE[3][3];                                                // F E
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int k = 0; k < 3; k++) FE[i][k] = F[i][0] * E[0][k] + F[i][1] * E[1][k] + F[i][2] * E[2][k];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        S[i][j] = E[i][0] * sig[0] * E[j][0] + E[i][1] * sig[1] * E[j][1] + E[i][2] * sig[2] * E[j][2];
        Q[i][j] = FE[i][0] * inv[0] * E[j][0] + FE[i][1] * inv[1] * E[j][1] + FE[i][2] * inv[2] * E[j][2];
      }
    if (inv[0] == 0.0 || inv[1] == 0.0 || inv[2] == 0.0) {       // a collapsed direction: complete Q on it by a cross product
      // columns of Q E for the non-collapsed directions are orthonormal; rebuild the missing one(s) only in the
      // single-collapse case (a flattened neighbourhood), otherwise fall back to the identity rotation
      int zc = (inv[0] == 0.0) + (inv[1] == 0.0) + (inv[2] == 0.0);
      if (zc == 1) {
        const int m = inv[0] == 0.0 ? 0 : (inv[1] == 0.0 ? 1 : 2), p = (m + 1) % 3, q = (m + 2) % 3;
        double up[3], uq[3], um[3];
#pragma unroll
        for (int i = 0; i < 3; i++) { up[i] = FE[i][p] * inv[p]; uq[i] = FE[i][q] * inv[q]; }
        cross3(up, uq, um);
        double ep[3] = {E[0][p], E[1][p], E[2][p]}, eq[3] = {E[0][q], E[1][q], E[2][q]}, em[3];
        cross3(ep, eq, em);                                        // = +-E[:, m]; using the same handedness keeps det Q = +1
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) Q[i][j] = up[i] * ep[j] + uq[i] * eq[j] + um[i] * em[j];
      } else {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) Q[i][j] = i == j ? 1.0 : 0.0;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (R_out) R_out[9 * (size_t)v + 3 * i + j] = (float)Q[j][i];       // R = Q^T (row-vector convention of the call site)
      if (S_out) S_out[9 * (size_t)v + 3 * i + j] = (float)S[i][j];
      if (state_out) {
        state_out[21 * (size_t)v + 3 + 3 * i + j] = (float)Q[j][i];
        state_out[21 * (size_t)v + 12 + 3 * i + j] = (float)S[i][j];
      }
    }
  if (state_out)
#pragma unroll
    for (int j = 0; j < 3; j++) state_out[21 * (size_t)v + j] = V1[3 * (size_t)v + j];
  if (packed_out) {          // the gather table of deform_shade_kernel<true, .> (what pack_mesh_state_kernel makes of `state`)
    const float r[9] = {(float)Q[0][0], (float)Q[1][0], (float)Q[2][0], (float)Q[0][1], (float)Q[1][1], (float)Q[2][1],
                        (float)Q[0][2], (float)Q[1][2], (float)Q[2][2]};
    const float t[9] = {(float)S[0][0], (float)S[0][1], (float)S[0][2], (float)S[1][0], (float)S[1][1], (float)S[1][2],
                        (float)S[2][0], (float)S[2][1], (float)S[2][2]};
    float4* o = packed_out + 6 * (size_t)v;
    o[0] = make_float4(p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2], 0.f);
    o[1] = make_float4(r[0], r[1], r[2], r[3]);
    o[2] = make_float4(r[4], r[5], r[6], r[7]);
    o[3] = make_float4(r[8], t[0], t[1], t[2]);
    o[4] = make_float4(t[3], t[4], t[5], t[6]);
    o[5] = make_float4(t[7], t[8], 0.f, 0.f);
  }
}

int launch_mesh_rs(int Vm, const float* V0, const float* V1, const int* faces, const int* adj_offsets, const int* adj_faces, float* R,
                   float* S, float* state, float* packed, hipStream_t s) {
  if (Vm <= 0) return 0;
  StageScope sc(ST_MESH_RS, s);          // a stage of its own: bench.py's `roofline` names single kernels
  MeshFrames mf{};
  mf.frames = 1;
  hipLaunchKernelGGL(mesh_rs_kernel, dim3((Vm + GM_MESH_THREADS - 1) / GM_MESH_THREADS), dim3(GM_MESH_THREADS), 0, s, Vm, V0, V1, faces, adj_offsets, adj_faces, R, S, state,
                     reinterpret_cast<float4*>(packed), mf);
  GM_HIP(hipGetLastError());
  return 0;
}

int launch_mesh_rs_batch(int frames, int Vm, const float* V0, const float* const* V1, const int* faces, const int* adj_offsets, const int* adj_faces,
                         float* const* packed, hipStream_t s) {
  if (Vm <= 0 || frames <= 0) return 0;
  StageScope sc(ST_MESH_RS, s);
  MeshFrames mf{};
  mf.frames = frames;
  for (int f = 0; f < frames; f++) { mf.V1[f] = V1[f]; mf.packed[f] = reinterpret_cast<float4*>(packed[f]); }
  hipLaunchKernelGGL(mesh_rs_kernel, dim3((Vm + GM_MESH_THREADS - 1) / GM_MESH_THREADS, 1, (uint32_t)frames), dim3(GM_MESH_THREADS), 0, s, Vm, V0, V1[0], faces, adj_offsets,
                     adj_faces, nullptr, nullptr, nullptr, reinterpret_cast<float4*>(packed[0]), mf);
  GM_HIP(hipGetLastError());
  return 0;
}

}  // namespace gm
