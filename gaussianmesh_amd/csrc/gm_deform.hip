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
#include "gm_stage.h"
#include "gm_pre_body.h"
#include <cstdlib>

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

// ---------------------------------------------------------------------------------------------
// Fused edit-loop kernel: deform + view-dependent colour in one pass (what ObjectVisualTool.render_gaussian needs per
// frame: means3D = x', colors_precomp, cov3D_precomp = strip_symmetric(Sigma'); edittool/__init__.py:421-472), and with PRE
// also the forward preprocess of the Gaussian it just produced.
// Reads 264 B and writes 48 B per Gaussian (+72 B if the caller also wants Sigma' and rot), against 72+108 (deform) +
// 240+12 (colour) for the two separate kernels.
// One-wave workgroups of 64 Gaussians.  The block's SH and covariance rows come into LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass, and the loads are in flight while the wave gathers
// its vertices): per wave the dependent memory rounds are two (ids -> {rows DMA, vertex gathers}).  The LDS image is the
// linear image of the rows (a DMA instruction writes wave-uniform base + lane x 16 B); the 192-byte row stride makes the
// per-thread b128 reads bank-conflicted, which is immaterial next to the memory latency this kernel is bound by.
// Results leave through LDS as coalesced 16-byte stores.
// PACKED: dV points at the per-vertex table written by pack_mesh_state_kernel (6 float4 per vertex:
// {dV,0} {R0..3} {R4..7} {R8,S0,S1,S2} {S3..6} {S7,S8,0,0}); the three vertex gathers of a Gaussian are then 18 16-byte
// loads instead of 63 4-byte ones (Rv / Sv unused).
// PRE (edit-loop fast path): followed, in the same thread, by the forward preprocess of the
// Gaussian it just produced (gm_pre_body.h, the code preprocess_fwd_kernel runs on colors_precomp / cov3D_precomp
// inputs).  The deformed position / covariance / colour never travel through HBM (48 B written + 48 B read per
// Gaussian and one launch less per frame); pos_out / cov6_out / rgb_out are optional copies for callers that want them.
struct FusedPre {
  PreCam cam;
  const float* opac;
  float4* splat; int* radii_int; int* radii_out; uint32_t* tiles; uint4* bin; uint32_t* counters; uint32_t* slots; uint32_t* coarse; uint8_t* clamped; uint32_t* depth_key;
  // DIRECT (gm_common.h, DepthSlab): the frame's snapshot of the stream's depth table and the bucket slabs the records are appended to
  const uint32_t* dmap; uint32_t* slab_cnt; uint2* slab_pairs; uint4* slab_recs; uint32_t slab_cap;
  int cov6;                     // the rest covariances come as [N][6] (a caller whose matrices are bit-symmetric: deform.pack_cov6)
};

// Direct depth placement: the wave's visible Gaussians are appended to their depth buckets' slabs.  One returning atomic per
// DISTINCT bucket of the wave (neighbours in memory are neighbours on the mesh: a wave usually spans two or three buckets), issued
// together; lanes whose bucket has not come up after GM_DIRECT_GROUPS rounds of the grouping loop (a cloud whose ids carry no
// locality) take a position on their own.  An entry beyond the slab's capacity is dropped - the counter still counts it and the
// frame is refused (direct_plan_kernel, gm_bucket.hip).
#ifndef GM_DIRECT_GROUPS
#define GM_DIRECT_GROUPS 6
#endif
struct DirectSlot { uint32_t bucket, base, rank; int leader; };
// first half, as soon as the depth key is known: bucket lookup, grouping, the returning atomics.  Nothing here waits for them - the
// emission count that follows (hundreds of instructions) runs while they are in flight.
__device__ __forceinline__ DirectSlot direct_reserve(const FusedPre& fp, uint32_t dkey) {
  const bool vis = dkey != 0xFFFFFFFFu;
  const int lane = threadIdx.x & 63;
  DirectSlot d;
  d.bucket = 0;
  if (vis) {
    const uint32_t e = fp.dmap[(dkey >> GM_COARSE_SHIFT) & (GM_COARSE_BINS - 1)];
    d.bucket = (e >> 16) + (((dkey & ((1u << GM_COARSE_SHIFT) - 1u)) * (e & 0xFFFFu)) >> GM_COARSE_SHIFT);
    d.bucket = min(d.bucket, (1u << GM_BUCKET_BITS) - 1u);   // (bins behind the table's last bucket: the last bucket - still monotone)
  }
  unsigned long long todo = __ballot(vis);
  uint32_t nsame = 1;
  d.rank = 0; d.leader = lane;
#pragma unroll 1
  for (int it = 0; it < GM_DIRECT_GROUPS && todo; it++) {
    const int l = __ffsll(todo) - 1;
    const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)d.bucket, l);
    const unsigned long long same = __ballot(vis && d.bucket == bl);
    if (vis && d.bucket == bl) { d.rank = lanes_below(same); d.leader = l; nsame = (uint32_t)__popcll(same); }
    todo &= ~same;
  }
  d.base = 0;
  if (vis && d.leader == lane) d.base = atomicAdd(fp.slab_cnt + (size_t)d.bucket * GM_SLAB_CNT_STRIDE, nsame);
  return d;
}
// second half: the leaders' positions reach their groups, the entry goes to its slab
__device__ __forceinline__ void direct_store(const FusedPre& fp, const DirectSlot& d, uint32_t id, uint32_t dkey, const uint4& bin) {
  const uint32_t pos = (uint32_t)__shfl((int)d.base, d.leader) + d.rank;
  if (dkey != 0xFFFFFFFFu && pos < fp.slab_cap) {
    const size_t at = (size_t)d.bucket * fp.slab_cap + pos;
    fp.slab_pairs[at] = make_uint2(dkey, id);
    fp.slab_recs[at] = bin;
  }
}

template <bool PACKED, bool PRE, bool DIRECT = false>
__global__ __launch_bounds__(64) void deform_shade_kernel(int N, int deg, const int* __restrict__ tri, const float* __restrict__ w,
                                                          const float* __restrict__ dV, const float* __restrict__ Rv,
                                                          const float* __restrict__ Sv, const float* __restrict__ cov,
                                                          const float* __restrict__ pos, const float* __restrict__ shs,
                                                          const float* __restrict__ campos, float* __restrict__ pos_out,
                                                          float* __restrict__ cov6_out, float* __restrict__ rgb_out,
                                                          float* __restrict__ cov_out, float* __restrict__ rot_out, const FusedPre fp) {
  constexpr int DS_THREADS = 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float4* l_sh = reinterpret_cast<float4*>(lds);   // [64][12] granules: linear image of the block's SH rows (12 KiB)
  const size_t row0 = (size_t)blockIdx.x * DS_THREADS;
  const int nrows = min(DS_THREADS, N - (int)row0);
  const int t = threadIdx.x;
  const size_t i = row0 + t;
  const bool live = t < nrows;
  float O[9], Rt[9], npos[3], col[3];
  int t0 = 0, t1 = 0, t2 = 0; float w0 = 0.f, w1 = 0.f, w2 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f;
  if (live) {
    t0 = tri[3 * i]; t1 = tri[3 * i + 1]; t2 = tri[3 * i + 2];
    w0 = w[3 * i]; w1 = w[3 * i + 1]; w2 = w[3 * i + 2];
    p0 = pos[3 * i]; p1 = pos[3 * i + 1]; p2 = pos[3 * i + 2];
  }
  // ids are in (first use below waits for them); now start the row DMA, then the vertex gathers behind it
  __builtin_amdgcn_sched_barrier(0);
  if (nrows == DS_THREADS) {
    const char* gsh = reinterpret_cast<const char*>(shs + row0 * 48) + t * 16;
#pragma unroll
    for (int q = 0; q < 12; q++) dma16(gsh + q * 1024, reinterpret_cast<char*>(l_sh) + q * 1024);
  } else if (live) {                               // last, partial block: plain copies of the thread's own rows
#pragma unroll
    for (int c = 0; c < 12; c++) l_sh[t * 12 + c] = reinterpret_cast<const float4*>(shs)[i * 12 + c];
  }
  // the covariance row goes straight to registers (36 B per lane, the wave's 2304 B are contiguous): staging it as well
  // would cost 2.25 KiB of LDS per wave, i.e. two resident waves per CU
  float C[9];
  if (PRE && fp.cov6) {                            // (wave-uniform) rest covariance as its six distinct entries xx xy xz yy yz zz: 24 B per lane
    float c6[6];
#pragma unroll
    for (int k = 0; k < 6; k++) c6[k] = live ? cov[i * 6 + k] : 0.f;
    C[0] = c6[0]; C[1] = c6[1]; C[2] = c6[2]; C[3] = c6[1]; C[4] = c6[3]; C[5] = c6[4]; C[6] = c6[2]; C[7] = c6[4]; C[8] = c6[5];
  } else {
#pragma unroll
    for (int k = 0; k < 9; k++) C[k] = live ? cov[i * 9 + k] : 0.f;
  }
  float d[3], Rb[9], Sb[9];
  if (PACKED) {
    const float4* tab = reinterpret_cast<const float4*>(dV);
    float va[24], vb[24], vc[24];
#pragma unroll
    for (int q = 0; q < 6; q++) {
      const float4 a = tab[6 * (size_t)t0 + q], b = tab[6 * (size_t)t1 + q], c = tab[6 * (size_t)t2 + q];
      va[4 * q] = a.x; va[4 * q + 1] = a.y; va[4 * q + 2] = a.z; va[4 * q + 3] = a.w;
      vb[4 * q] = b.x; vb[4 * q + 1] = b.y; vb[4 * q + 2] = b.z; vb[4 * q + 3] = b.w;
      vc[4 * q] = c.x; vc[4 * q + 1] = c.y; vc[4 * q + 2] = c.z; vc[4 * q + 3] = c.w;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) d[k] = (w0 * va[k] + w1 * vb[k]) + w2 * vc[k];
#pragma unroll
    for (int k = 0; k < 9; k++) {
      Rb[k] = (w0 * va[4 + k] + w1 * vb[4 + k]) + w2 * vc[4 + k];
      Sb[k] = (w0 * va[13 + k] + w1 * vb[13 + k]) + w2 * vc[13 + k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 3; k++) d[k] = (w0 * dV[3 * (size_t)t0 + k] + w1 * dV[3 * (size_t)t1 + k]) + w2 * dV[3 * (size_t)t2 + k];
#pragma unroll
    for (int k = 0; k < 9; k++) {
      Rb[k] = (w0 * Rv[9 * (size_t)t0 + k] + w1 * Rv[9 * (size_t)t1 + k]) + w2 * Rv[9 * (size_t)t2 + k];
      Sb[k] = (w0 * Sv[9 * (size_t)t0 + k] + w1 * Sv[9 * (size_t)t1 + k]) + w2 * Sv[9 * (size_t)t2 + k];
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): the DMA has landed
  __syncthreads();
  {
    float RS[9], A[9];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) Rt[3 * a + b] = Rb[3 * b + a];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) RS[3 * a + b] = (Rt[3 * a] * Sb[b] + Rt[3 * a + 1] * Sb[3 + b]) + Rt[3 * a + 2] * Sb[6 + b];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) A[3 * a + b] = (RS[3 * a] * C[b] + RS[3 * a + 1] * C[3 + b]) + RS[3 * a + 2] * C[6 + b];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) O[3 * a + b] = (A[3 * a] * RS[3 * b] + A[3 * a + 1] * RS[3 * b + 1]) + A[3 * a + 2] * RS[3 * b + 2];
    npos[0] = p0 + d[0]; npos[1] = p1 + d[1]; npos[2] = p2 + d[2];
    float dx = npos[0] - campos[0], dy = npos[1] - campos[1], dz = npos[2] - campos[2];
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    dx = dx / len; dy = dy / len; dz = dz / len;
    // dir_rot = rot^T dir with rot = Rt
    const float x = (Rt[0] * dx + Rt[3] * dy) + Rt[6] * dz;
    const float y = (Rt[1] * dx + Rt[4] * dy) + Rt[7] * dz;
    const float z = (Rt[2] * dx + Rt[5] * dy) + Rt[8] * dz;
    float sh[48];
#pragma unroll
    for (int c = 0; c < 12; c++) {
      const float4 v = l_sh[t * 12 + c];
      sh[4 * c] = v.x; sh[4 * c + 1] = v.y; sh[4 * c + 2] = v.z; sh[4 * c + 3] = v.w;
    }
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
      const float r = sh_channel(deg, [&](int k) { return sh[3 * k + ch]; }, x, y, z);
      col[ch] = fmaxf(r + 0.5f, 0.0f);
    }
  }
  // ---- forward preprocess of the deformed Gaussian (colors_precomp / cov3D_precomp input mode)
  uint32_t tiles = 0, dkey = 0xFFFFFFFFu;
  uint4 dbin = make_uint4(0u, 0u, 0u, 0u);
  PreGeom pg;
  bool vis = false;
  if (PRE && live) {
    const V3 p = {npos[0], npos[1], npos[2]};
    const float c3[6] = {O[0], O[1], O[2], O[4], O[5], O[8]};
    vis = pre_project(fp.cam, p, c3, pg);
    if (vis) dkey = __float_as_uint(pg.depth);
  }
  DirectSlot slot;
  if (PRE && DIRECT) slot = direct_reserve(fp, dkey);          // (whole wave)
  if (PRE && live) {
    int radius_i = 0;
    if (vis) {
      const float opac = fp.opac[i];
      splat_store(fp.splat, i, pg.pix, pg.piy, pg.conx, pg.cony, pg.conz, opac, col[0], col[1], col[2], pg.depth);
      radius_i = (int)pg.radius;
      pre_emit(fp.cam, pg, opac, tiles, dbin);
    }
    // Written once, where it is read: the radius goes to the caller's array (the internal copy exists for callers that pass none);
    // the clamp flags and the per-Gaussian instance count are read by the backward pass and by the emission of a rectangle with
    // 65535 instances or more only - a frame of this path has no backward, and the count rides in the emission record otherwise
    // (9 of 349 bytes per Gaussian in a kernel that runs at the HBM's pace)
    if (fp.radii_out) fp.radii_out[i] = radius_i; else fp.radii_int[i] = radius_i;
    if (tiles >= GM_BIN_COUNT_SAT) fp.tiles[i] = tiles;
    if (!DIRECT) {
      fp.bin[i] = dbin;
      fp.depth_key[i] = dkey;
    }
    if (i == 0) fp.counters[GM_CNT_POLICY] = (uint32_t)fp.cam.tile_cull;
  }
  if (PRE && DIRECT) direct_store(fp, slot, (uint32_t)i, dkey, dbin);
  if (PRE) slot_accumulate(fp.slots, fp.coarse, tiles, dkey);
  if (!pos_out) return;                          // wave-uniform
  __syncthreads();                               // everyone is done reading the staged inputs: reuse LDS for the outputs
  float* o_pos = lds;                            // [256][3]
  float* o_rgb = lds + DS_THREADS * 3;           // [256][3]
  float* o_c6 = lds + DS_THREADS * 6;            // [256][6]  (stride 7 to stay conflict-free)
  float* o_cov = lds + DS_THREADS * 13;          // [256][9]
  float* o_rot = lds + DS_THREADS * 22;          // [256][9]
#pragma unroll
  for (int k = 0; k < 3; k++) { o_pos[t * 3 + k] = npos[k]; o_rgb[t * 3 + k] = col[k]; }
  o_c6[t * 7 + 0] = O[0]; o_c6[t * 7 + 1] = O[1]; o_c6[t * 7 + 2] = O[2]; o_c6[t * 7 + 3] = O[4]; o_c6[t * 7 + 4] = O[5]; o_c6[t * 7 + 5] = O[8];
  if (cov_out) {
#pragma unroll
    for (int k = 0; k < 9; k++) { o_cov[t * 9 + k] = O[k]; o_rot[t * 9 + k] = Rt[k]; }
  }
  __syncthreads();
  unstage_rows<3, 3, DS_THREADS>(pos_out, row0, nrows, o_pos);
  unstage_rows<3, 3, DS_THREADS>(rgb_out, row0, nrows, o_rgb);
  unstage_rows<6, 7, DS_THREADS>(cov6_out, row0, nrows, o_c6);
  if (cov_out) {
    unstage_rows<9, 9, DS_THREADS>(cov_out, row0, nrows, o_cov);
    unstage_rows<9, 9, DS_THREADS>(rot_out, row0, nrows, o_rot);
  }
}

// ---------------------------------------------------------------------------------------------
// K frames of one view stream from ONE pass over the static cloud (gm_forward_deformed_batch_async).  Of the 321 bytes per Gaussian the
// fused pass of a frame moves, 256 do not depend on the frame - face ids 12, weights 12, rest covariance 24 (36), rest position 12, SH row
// 192, opacity 4 - and a render loop with several frames in flight streamed them from HBM once per frame.  Here a wave loads them once
// (the SH rows stay in LDS, the rest in ~20 registers) and loops over the batch's frames: gather the frame's per-vertex table (L2
// resident: 0.7 MB per frame), deform, rotated-direction SH colour, project, emission record -> the frame's own geometry buffer.  Every
// expression is the one deform_shade_kernel<true, true> evaluates, in the same order, contraction off: each frame's geometry buffer comes
// out bit for bit as from the single-frame launch (tests/test_gpu_batch.py).
struct FusedFrame {
  PreCam cam;
  const float4* tab; const float* campos;
  float4* splat; int* radii_int; int* radii_out; uint32_t* tiles; uint4* bin; uint32_t* counters; uint32_t* slots; uint32_t* coarse; uint32_t* depth_key;
};
struct FusedBatch { int frames, cov6; const float* opac; FusedFrame f[GM_BATCH_MAX]; };

__global__ __launch_bounds__(64) void deform_shade_pre_batch_kernel(int N, int deg, const int* __restrict__ tri, const float* __restrict__ w,
                                                                     const float* __restrict__ cov, const float* __restrict__ pos,
                                                                     const float* __restrict__ shs, const FusedBatch fb) {
  constexpr int DS_THREADS = 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float4* l_sh = reinterpret_cast<float4*>(lds);   // [64][12] granules: linear image of the block's SH rows (12 KiB)
  const size_t row0 = (size_t)blockIdx.x * DS_THREADS;
  const int nrows = min(DS_THREADS, N - (int)row0);
  const int t = threadIdx.x;
  const size_t i = row0 + t;
  const bool live = t < nrows;
  int t0 = 0, t1 = 0, t2 = 0; float w0 = 0.f, w1 = 0.f, w2 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, opac = 0.f;
  if (live) {
    t0 = tri[3 * i]; t1 = tri[3 * i + 1]; t2 = tri[3 * i + 2];
    w0 = w[3 * i]; w1 = w[3 * i + 1]; w2 = w[3 * i + 2];
    p0 = pos[3 * i]; p1 = pos[3 * i + 1]; p2 = pos[3 * i + 2];
    opac = fb.opac[i];
  }
  __builtin_amdgcn_sched_barrier(0);
  if (nrows == DS_THREADS) {
    const char* gsh = reinterpret_cast<const char*>(shs + row0 * 48) + t * 16;
#pragma unroll
    for (int q = 0; q < 12; q++) dma16(gsh + q * 1024, reinterpret_cast<char*>(l_sh) + q * 1024);
  } else if (live) {
#pragma unroll
    for (int c = 0; c < 12; c++) l_sh[t * 12 + c] = reinterpret_cast<const float4*>(shs)[i * 12 + c];
  }
  float C[9];
  if (fb.cov6) {
    float c6[6];
#pragma unroll
    for (int k = 0; k < 6; k++) c6[k] = live ? cov[i * 6 + k] : 0.f;
    C[0] = c6[0]; C[1] = c6[1]; C[2] = c6[2]; C[3] = c6[1]; C[4] = c6[3]; C[5] = c6[4]; C[6] = c6[2]; C[7] = c6[4]; C[8] = c6[5];
  } else {
#pragma unroll
    for (int k = 0; k < 9; k++) C[k] = live ? cov[i * 9 + k] : 0.f;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): the DMA has landed
  __syncthreads();
#pragma unroll 1
  for (int f = 0; f < fb.frames; f++) {
    const FusedFrame& F = fb.f[f];
    float d[3], Rb[9], Sb[9];
    {
      const float4* tab = F.tab;
      float va[24], vb[24], vc[24];
#pragma unroll
      for (int q = 0; q < 6; q++) {
        const float4 a = tab[6 * (size_t)t0 + q], b = tab[6 * (size_t)t1 + q], c = tab[6 * (size_t)t2 + q];
        va[4 * q] = a.x; va[4 * q + 1] = a.y; va[4 * q + 2] = a.z; va[4 * q + 3] = a.w;
        vb[4 * q] = b.x; vb[4 * q + 1] = b.y; vb[4 * q + 2] = b.z; vb[4 * q + 3] = b.w;
        vc[4 * q] = c.x; vc[4 * q + 1] = c.y; vc[4 * q + 2] = c.z; vc[4 * q + 3] = c.w;
      }
#pragma unroll
      for (int k = 0; k < 3; k++) d[k] = (w0 * va[k] + w1 * vb[k]) + w2 * vc[k];
#pragma unroll
      for (int k = 0; k < 9; k++) {
        Rb[k] = (w0 * va[4 + k] + w1 * vb[4 + k]) + w2 * vc[4 + k];
        Sb[k] = (w0 * va[13 + k] + w1 * vb[13 + k]) + w2 * vc[13 + k];
      }
    }
    float O[9], Rt[9], npos[3], col[3];
    {
      float RS[9], A[9];
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) Rt[3 * a + b] = Rb[3 * b + a];
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) RS[3 * a + b] = (Rt[3 * a] * Sb[b] + Rt[3 * a + 1] * Sb[3 + b]) + Rt[3 * a + 2] * Sb[6 + b];
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) A[3 * a + b] = (RS[3 * a] * C[b] + RS[3 * a + 1] * C[3 + b]) + RS[3 * a + 2] * C[6 + b];
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) O[3 * a + b] = (A[3 * a] * RS[3 * b] + A[3 * a + 1] * RS[3 * b + 1]) + A[3 * a + 2] * RS[3 * b + 2];
      npos[0] = p0 + d[0]; npos[1] = p1 + d[1]; npos[2] = p2 + d[2];
      const float* campos = F.campos;
      float dx = npos[0] - campos[0], dy = npos[1] - campos[1], dz = npos[2] - campos[2];
      const float len = sqrtf(dx * dx + dy * dy + dz * dz);
      dx = dx / len; dy = dy / len; dz = dz / len;
      const float x = (Rt[0] * dx + Rt[3] * dy) + Rt[6] * dz;
      const float y = (Rt[1] * dx + Rt[4] * dy) + Rt[7] * dz;
      const float z = (Rt[2] * dx + Rt[5] * dy) + Rt[8] * dz;
      float sh[48];
#pragma unroll
      for (int c = 0; c < 12; c++) {
        const float4 v = l_sh[t * 12 + c];
        sh[4 * c] = v.x; sh[4 * c + 1] = v.y; sh[4 * c + 2] = v.z; sh[4 * c + 3] = v.w;
      }
#pragma unroll
      for (int ch = 0; ch < 3; ch++) {
        const float r = sh_channel(deg, [&](int k) { return sh[3 * k + ch]; }, x, y, z);
        col[ch] = fmaxf(r + 0.5f, 0.0f);
      }
    }
    uint32_t tiles = 0, dkey = 0xFFFFFFFFu;
    uint4 dbin = make_uint4(0u, 0u, 0u, 0u);
    PreGeom pg;
    bool vis = false;
    if (live) {
      const V3 p = {npos[0], npos[1], npos[2]};
      const float c3[6] = {O[0], O[1], O[2], O[4], O[5], O[8]};
      vis = pre_project(F.cam, p, c3, pg);
      if (vis) dkey = __float_as_uint(pg.depth);
      int radius_i = 0;
      if (vis) {
        splat_store(F.splat, i, pg.pix, pg.piy, pg.conx, pg.cony, pg.conz, opac, col[0], col[1], col[2], pg.depth);
        radius_i = (int)pg.radius;
        pre_emit(F.cam, pg, opac, tiles, dbin);
      }
      if (F.radii_out) F.radii_out[i] = radius_i; else F.radii_int[i] = radius_i;
      if (tiles >= GM_BIN_COUNT_SAT) F.tiles[i] = tiles;
      F.bin[i] = dbin;
      F.depth_key[i] = dkey;
      if (i == 0) F.counters[GM_CNT_POLICY] = (uint32_t)F.cam.tile_cull;
    }
    slot_accumulate(F.slots, F.coarse, tiles, dkey);
  }
}

int launch_deform_shade_pre_batch(int frames, const BatchFrameArgs* fr, int P, int deg, int W, int H, int tile_cull, const int* tri, const float* w,
                                  const float* cov, const float* pos, const float* shs, const float* opacities, bool cov6, int debug, hipStream_t s) {
  if (P <= 0) return 0;
  if (frames < 1 || frames > GM_BATCH_MAX) { set_error("batched fused pass: 1..%d frames", GM_BATCH_MAX); return 1; }
  if (!aligned16(shs) || !aligned16(cov)) { set_error("gm_forward_deformed_batch: shs / cov must be 16-byte aligned"); return 1; }
  StageScope sc(ST_DEFORM, s);
  FusedBatch fb{};
  fb.frames = frames; fb.cov6 = cov6 ? 1 : 0; fb.opac = opacities;
  for (int k = 0; k < frames; k++) {
    const BatchFrameArgs& a = fr[k];
    if (!aligned16(a.packed)) { set_error("gm_forward_deformed_batch: packed tables must be 16-byte aligned"); return 1; }
    FusedFrame& F = fb.f[k];
    F.cam.W = W; F.cam.H = H; F.cam.gx = (W + GM_TILE - 1) / GM_TILE; F.cam.gy = (H + GM_TILE - 1) / GM_TILE;
    F.cam.tile_cull = tile_cull; F.cam.view = a.viewmatrix; F.cam.proj = a.projmatrix;
    F.cam.tanx = a.tan_fovx; F.cam.tany = a.tan_fovy;
    F.cam.fy = H / (2.0f * a.tan_fovy); F.cam.fx = W / (2.0f * a.tan_fovx);      // rasterizer_impl.cu:359-360
    F.tab = reinterpret_cast<const float4*>(a.packed); F.campos = a.cam_pos;
    const GeomState& g = a.g;
    F.splat = g.splat; F.radii_int = g.radii; F.radii_out = a.radii; F.tiles = g.tiles_touched; F.bin = g.bin; F.counters = g.counters; F.slots = g.slots;
    F.coarse = g.coarse; F.depth_key = g.depth_key;
  }
  hipLaunchKernelGGL(deform_shade_pre_batch_kernel, dim3((P + 63) / 64), dim3(64), sizeof(float) * 64 * 48, s, P, deg, tri, w, cov, pos, shs, fb);
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

// mesh state of one deformation frame, as broadcast to the ranks ([Vm][21] = V1 | R | S), minus the rest pose -> the
// gather table of deform_shade_kernel<.., true>
__global__ __launch_bounds__(256) void pack_mesh_state_kernel(int Vm, const float* __restrict__ state, const float* __restrict__ verts,
                                                               float4* __restrict__ packed) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= Vm) return;
  const float* s = state + 21 * (size_t)v;
  float4* o = packed + 6 * (size_t)v;
  o[0] = make_float4(s[0] - verts[3 * (size_t)v], s[1] - verts[3 * (size_t)v + 1], s[2] - verts[3 * (size_t)v + 2], 0.f);
  o[1] = make_float4(s[3], s[4], s[5], s[6]);
  o[2] = make_float4(s[7], s[8], s[9], s[10]);
  o[3] = make_float4(s[11], s[12], s[13], s[14]);
  o[4] = make_float4(s[15], s[16], s[17], s[18]);
  o[5] = make_float4(s[19], s[20], 0.f, 0.f);
}

int launch_pack_mesh_state(int Vm, const float* state, const float* verts, float* packed, hipStream_t s) {
  if (Vm <= 0) return 0;
  StageScope sc(ST_DEFORM, s);
  hipLaunchKernelGGL(pack_mesh_state_kernel, dim3((Vm + 255) / 256), dim3(256), 0, s, Vm, state, verts, reinterpret_cast<float4*>(packed));
  GM_HIP(hipGetLastError());
  return 0;
}

int launch_deform_shade_packed(int N, int deg, int M, const int* tri, const float* w, const float* packed, const float* cov,
                               const float* pos, const float* shs, const float* campos, float* pos_out, float* cov6_out,
                               float* rgb_out, float* cov_out, float* rot_out, hipStream_t s) {
  if (N <= 0) return 0;
  if (M != 16 || !aligned16(shs) || !aligned16(cov) || !aligned16(pos_out) || !aligned16(cov6_out) || !aligned16(rgb_out) ||
      !aligned16(packed) || (cov_out && (!aligned16(cov_out) || !aligned16(rot_out)))) {
    set_error("gm_deform_shade_packed: needs M == 16 and 16-byte aligned buffers"); return 1;
  }
  StageScope sc(ST_DEFORM, s);
  const size_t lds_bytes = sizeof(float) * 64 * 48;
  hipLaunchKernelGGL((deform_shade_kernel<true, false>), dim3((N + 63) / 64), dim3(64), lds_bytes, s, N, deg, tri, w, packed, nullptr, nullptr,
                     cov, pos, shs, campos, pos_out, cov6_out, rgb_out, cov_out, rot_out, FusedPre{});
  GM_HIP(hipGetLastError());
  return 0;
}

int launch_deform_shade_pre(const RasterArgs& r, GeomState& g, int* radii, int deg, const int* tri, const float* w, const float* packed,
                            const float* cov, const float* pos, const float* shs, float* pos_out, float* cov6_out, float* rgb_out,
                            const DepthSlab* slab, bool cov6) {
  const int N = r.P;
  if (N <= 0) return 0;
  if (!aligned16(shs) || !aligned16(cov) || !aligned16(packed) ||
      (pos_out && (!aligned16(pos_out) || !aligned16(cov6_out) || !aligned16(rgb_out)))) {
    set_error("gm_forward_0_deformed: shs / cov / packed / outputs must be 16-byte aligned"); return 1;
  }
  StageScope sc(ST_DEFORM, r.stream);
  FusedPre fp;
  fp.cam.W = r.W; fp.cam.H = r.H; fp.cam.gx = (r.W + GM_TILE - 1) / GM_TILE; fp.cam.gy = (r.H + GM_TILE - 1) / GM_TILE;
  fp.cam.tile_cull = r.tile_cull; fp.cam.view = r.viewmatrix; fp.cam.proj = r.projmatrix;
  fp.cam.tanx = r.tan_fovx; fp.cam.tany = r.tan_fovy;
  fp.cam.fy = r.H / (2.0f * r.tan_fovy); fp.cam.fx = r.W / (2.0f * r.tan_fovx);   // rasterizer_impl.cu:359-360
  fp.opac = r.opacities;
  fp.splat = g.splat; fp.radii_int = g.radii; fp.radii_out = radii; fp.tiles = g.tiles_touched; fp.bin = g.bin; fp.counters = g.counters; fp.slots = g.slots; fp.coarse = g.coarse;
  fp.clamped = g.clamped; fp.depth_key = g.depth_key;
  fp.cov6 = cov6 ? 1 : 0;
  fp.dmap = g.dmap; fp.slab_cnt = nullptr; fp.slab_pairs = nullptr; fp.slab_recs = nullptr; fp.slab_cap = 0;
  if (slab) { fp.slab_cnt = slab->cnt; fp.slab_pairs = slab->pairs; fp.slab_recs = slab->recs; fp.slab_cap = slab->cap; }
  // (measured: fewer resident workgroups per CU - 8 / 6 / 5 instead of 12, by padding this allocation - make the kernel 10 / 24 / 52 %
  // slower and the four-stream loop 6 / 13 / 21 %: it needs every wave it can get to keep enough bytes in flight)
  const size_t lds_bytes = sizeof(float) * 64 * 48;
  if (slab)
    hipLaunchKernelGGL((deform_shade_kernel<true, true, true>), dim3((N + 63) / 64), dim3(64), lds_bytes, r.stream, N, deg, tri, w, packed, nullptr,
                       nullptr, cov, pos, shs, r.cam_pos, pos_out, cov6_out, rgb_out, nullptr, nullptr, fp);
  else
    hipLaunchKernelGGL((deform_shade_kernel<true, true>), dim3((N + 63) / 64), dim3(64), lds_bytes, r.stream, N, deg, tri, w, packed, nullptr,
                       nullptr, cov, pos, shs, r.cam_pos, pos_out, cov6_out, rgb_out, nullptr, nullptr, fp);
  GM_LAUNCH_CHECK(r.debug, r.stream);
  return 0;
}

int launch_deform_shade(int N, int deg, int M, const int* tri, const float* w, const float* dV, const float* Rv, const float* Sv,
                        const float* cov, const float* pos, const float* shs, const float* campos, float* pos_out,
                        float* cov6_out, float* rgb_out, float* cov_out, float* rot_out, hipStream_t s) {
  if (N <= 0) return 0;
  if (M != 16 || !aligned16(shs) || !aligned16(cov) || !aligned16(pos_out) || !aligned16(cov6_out) || !aligned16(rgb_out) ||
      (cov_out && (!aligned16(cov_out) || !aligned16(rot_out)))) {
    // general layout: run the two unfused kernels (needs the 72 B/Gaussian intermediates from the caller)
    if (!cov_out || !rot_out) { set_error("gm_deform_shade: M != 16 or unaligned buffers need cov_out/rot_out"); return 1; }
    if (int rc = launch_deform(N, tri, w, dV, Rv, Sv, cov, pos, pos_out, cov_out, rot_out, cov6_out, s)) return rc;
    return launch_sh_colors(N, deg, M, pos_out, campos, rot_out, shs, rgb_out, s);
  }
  StageScope sc(ST_DEFORM, s);
  const size_t lds_bytes = sizeof(float) * 64 * 48;
  hipLaunchKernelGGL((deform_shade_kernel<false, false>), dim3((N + 63) / 64), dim3(64), lds_bytes, s, N, deg, tri, w, dV, Rv, Sv, cov, pos,
                     shs, campos, pos_out, cov6_out, rgb_out, cov_out, rot_out, FusedPre{});
  GM_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Covariance -> (scale, quaternion): replaces SceneVisualTool.render_gaussian's per-frame
//   eigh(cov) on the device, det sign fixed on the HOST through numpy, sqrt(eigenvalues), matrix -> quaternion
// (edittool/__init__.py:204-207 and :23-38), i.e. a device->host->device round trip of 36 B per Gaussian per frame.
// One thread per Gaussian: cyclic Jacobi in double (3x3 symmetric, <= 8 sweeps), eigenvalues ascending like eigh,
// eigenvector matrix multiplied by sign(det) as the reference does, quaternion (w,x,y,z) by the branch-safe
// largest-component form (the reference's w = sqrt(1+trace)/2 divides by zero for trace <= -1), normalised.
// Eigenvector signs are not unique, so parity is on the reconstructed covariance R diag(s^2) R^T, not on (s, q).
__global__ __launch_bounds__(256) void cov_to_scale_rot_kernel(int N, const float* __restrict__ cov, float* __restrict__ scales,
                                                               float* __restrict__ rots) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  const float* c = cov + 9 * (size_t)i;
  // symmetrise (the deformed covariance RS C RS^T is symmetric up to rounding)
  A[0][0] = c[0]; A[1][1] = c[4]; A[2][2] = c[8];
  A[0][1] = A[1][0] = 0.5 * ((double)c[1] + c[3]);
  A[0][2] = A[2][0] = 0.5 * ((double)c[2] + c[6]);
  A[1][2] = A[2][1] = 0.5 * ((double)c[5] + c[7]);
  const double scale = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]) + 1e-300;
  for (int sweep = 0; sweep < 8; sweep++) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    if (off <= 1e-15 * scale) break;
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
  // ascending order of eigenvalues (columns of V follow)
  double ev[3] = {A[0][0], A[1][1], A[2][2]};
  int idx[3] = {0, 1, 2};
#define CSWAP(a_, b_) if (ev[idx[a_]] > ev[idx[b_]]) { const int t_ = idx[a_]; idx[a_] = idx[b_]; idx[b_] = t_; }
  CSWAP(0, 1) CSWAP(1, 2) CSWAP(0, 1)
#undef CSWAP
  double U[3][3];
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int j = 0; j < 3; j++) U[k][j] = V[k][idx[j]];
  const double det = U[0][0] * (U[1][1] * U[2][2] - U[1][2] * U[2][1]) - U[0][1] * (U[1][0] * U[2][2] - U[1][2] * U[2][0]) +
                     U[0][2] * (U[1][0] * U[2][1] - U[1][1] * U[2][0]);
  const double sg = det < 0 ? -1.0 : 1.0;
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int j = 0; j < 3; j++) U[k][j] *= sg;
  // rotation matrix -> quaternion, largest-component branch
  const double tr = U[0][0] + U[1][1] + U[2][2];
  double qw, qx, qy, qz;
  if (tr > 0) {
    const double s4 = 2.0 * sqrt(1.0 + tr);
    qw = 0.25 * s4; qx = (U[2][1] - U[1][2]) / s4; qy = (U[0][2] - U[2][0]) / s4; qz = (U[1][0] - U[0][1]) / s4;
  } else if (U[0][0] > U[1][1] && U[0][0] > U[2][2]) {
    const double s4 = 2.0 * sqrt(1.0 + U[0][0] - U[1][1] - U[2][2]);
    qw = (U[2][1] - U[1][2]) / s4; qx = 0.25 * s4; qy = (U[0][1] + U[1][0]) / s4; qz = (U[0][2] + U[2][0]) / s4;
  } else if (U[1][1] > U[2][2]) {
    const double s4 = 2.0 * sqrt(1.0 + U[1][1] - U[0][0] - U[2][2]);
    qw = (U[0][2] - U[2][0]) / s4; qx = (U[0][1] + U[1][0]) / s4; qy = 0.25 * s4; qz = (U[1][2] + U[2][1]) / s4;
  } else {
    const double s4 = 2.0 * sqrt(1.0 + U[2][2] - U[0][0] - U[1][1]);
    qw = (U[1][0] - U[0][1]) / s4; qx = (U[0][2] + U[2][0]) / s4; qy = (U[1][2] + U[2][1]) / s4; qz = 0.25 * s4;
  }
  const double qn = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
  reinterpret_cast<float4*>(rots)[i] = make_float4((float)(qw * qn), (float)(qx * qn), (float)(qy * qn), (float)(qz * qn));
#pragma unroll
  for (int j = 0; j < 3; j++) scales[3 * (size_t)i + j] = (float)sqrt(fmax(ev[idx[j]], 0.0));
}

int launch_cov_to_scale_rot(int N, const float* cov, float* scales, float* rots, hipStream_t s) {
  if (N > 0) hipLaunchKernelGGL(cov_to_scale_rot_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, cov, scales, rots);
  GM_HIP(hipGetLastError());
  return 0;
}

}  // namespace gm
