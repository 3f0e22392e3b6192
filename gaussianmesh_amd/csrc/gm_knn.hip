// gm_knn.hip -- mean squared distance to the 3 nearest neighbours (model initialisation helper).
//
// Replaces SimpleKNN::knn (scene/simple_knn/cuda_headers/simple_knn.cu:185-221; python entry
// scene/simple_knn/__init__.py:15-28 distCUDA2).  Same algorithm family: Morton order, 1024-point boxes,
// exact 3-NN with box-distance pruning; the result (mean of the three smallest squared distances to other
// points, duplicates count as distance 0) does not depend on the acceleration structure.
// Differences by design: no internal allocation (caller workspace), no host readback of the bounding box
// (it stays on the device), our own radix sort (gm_sort.hip).
#include "gm_common.h"
#include <cfloat>
#pragma clang fp contract(off)   // squared distances evaluated exactly as the CPU oracle does

namespace gm {

#define KNN_BOX 1024

struct KnnWs {
  float* bbox_partial;   // [1024][6]
  float* bbox;           // [6] min xyz, max xyz (reduction seeded with 0 like the reference, simple_knn.cu:191-200)
  uint32_t* keys[2];
  uint32_t* idx[2];
  uint32_t* hist;
  uint32_t* digit_total;
  float* boxes;          // [nboxes][6]
  char* end;
  static KnnWs from(void* ws, size_t P) {
    char* p = reinterpret_cast<char*>(ws);
    KnnWs k;
    k.bbox_partial = carve<float>(p, 1024 * 6);
    k.bbox = carve<float>(p, 8);
    k.keys[0] = carve<uint32_t>(p, P); k.keys[1] = carve<uint32_t>(p, P);
    k.idx[0] = carve<uint32_t>(p, P); k.idx[1] = carve<uint32_t>(p, P);
    k.hist = carve<uint32_t>(p, 256 * sort_blocks(P));
    k.digit_total = carve<uint32_t>(p, sort_chunk_counters(P));
    k.boxes = carve<float>(p, 6 * ((P + KNN_BOX - 1) / KNN_BOX));
    k.end = p;
    return k;
  }
};

size_t knn_workspace_bytes(int P) {
  KnnWs k = KnnWs::from(nullptr, (size_t)(P > 0 ? P : 1));
  return (size_t)k.end + 256;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
  return v;
}

// block-wide min/max of 6 values; result valid in thread 0
__device__ __forceinline__ void block_minmax(float* mn, float* mx, float (*sm)[6]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 3; k++) { mn[k] = wave_min(mn[k]); mx[k] = wave_max(mx[k]); }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) { sm[wave][k] = mn[k]; sm[wave][3 + k] = mx[k]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); w++)
#pragma unroll
      for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], sm[w][k]); mx[k] = fmaxf(mx[k], sm[w][3 + k]); }
  }
}

__global__ __launch_bounds__(256) void knn_bbox_partial(int P, const float* __restrict__ pts, float* __restrict__ partial) {
  __shared__ float sm[4][6];
  float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
#pragma unroll
    for (int k = 0; k < 3; k++) { const float v = pts[3 * (size_t)i + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
  }
  block_minmax(mn, mx, sm);
  if (threadIdx.x == 0)
#pragma unroll
    for (int k = 0; k < 3; k++) { partial[6 * blockIdx.x + k] = mn[k]; partial[6 * blockIdx.x + 3 + k] = mx[k]; }
}

__global__ __launch_bounds__(256) void knn_bbox_final(int nb, const float* __restrict__ partial, float* __restrict__ bbox) {
  __shared__ float sm[4][6];
  float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < nb; i += 256)
#pragma unroll
    for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], partial[6 * i + k]); mx[k] = fmaxf(mx[k], partial[6 * i + 3 + k]); }
  block_minmax(mn, mx, sm);
  if (threadIdx.x == 0)
#pragma unroll
    for (int k = 0; k < 3; k++) { bbox[k] = mn[k]; bbox[3 + k] = mx[k]; }
}

// simple_knn.cu:45-61
__device__ __forceinline__ uint32_t prep_morton(uint32_t x) {
  x = (x | (x << 16)) & 0x030000FF;
  x = (x | (x << 8)) & 0x0300F00F;
  x = (x | (x << 4)) & 0x030C30C3;
  x = (x | (x << 2)) & 0x09249249;
  return x;
}

__global__ __launch_bounds__(256) void knn_morton(int P, const float* __restrict__ pts, const float* __restrict__ bbox,
                                                  uint32_t* __restrict__ codes) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const float mnx = bbox[0], mny = bbox[1], mnz = bbox[2], mxx = bbox[3], mxy = bbox[4], mxz = bbox[5];
  const uint32_t x = prep_morton((uint32_t)(((pts[3 * (size_t)i] - mnx) / (mxx - mnx)) * ((1 << 10) - 1)));
  const uint32_t y = prep_morton((uint32_t)(((pts[3 * (size_t)i + 1] - mny) / (mxy - mny)) * ((1 << 10) - 1)));
  const uint32_t z = prep_morton((uint32_t)(((pts[3 * (size_t)i + 2] - mnz) / (mxz - mnz)) * ((1 << 10) - 1)));
  codes[i] = x | (y << 1) | (z << 2);
}

// one workgroup per box of 1024 Morton-consecutive points (simple_knn.cu:78-117)
__global__ __launch_bounds__(256) void knn_box_minmax(int P, const float* __restrict__ pts, const uint32_t* __restrict__ idx,
                                                      float* __restrict__ boxes) {
  __shared__ float sm[4][6];
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
#pragma unroll
  for (int r = 0; r < KNN_BOX / 256; r++) {
    const int i = blockIdx.x * KNN_BOX + r * 256 + threadIdx.x;
    if (i < P) {
      const uint32_t g = idx[i];
#pragma unroll
      for (int k = 0; k < 3; k++) { const float v = pts[3 * (size_t)g + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
    }
  }
  block_minmax(mn, mx, sm);
  if (threadIdx.x == 0)
#pragma unroll
    for (int k = 0; k < 3; k++) { boxes[6 * blockIdx.x + k] = mn[k]; boxes[6 * blockIdx.x + 3 + k] = mx[k]; }
}

__device__ __forceinline__ void update3(float dist, float* best) {   // updateKBest<3>, simple_knn.cu:131-145
#pragma unroll
  for (int j = 0; j < 3; j++)
    if (best[j] > dist) { const float t = best[j]; best[j] = dist; dist = t; }
}
__device__ __forceinline__ float dist2(const float* a, const float* b) {
  const float dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
  return dx * dx + dy * dy + dz * dz;
}

// one thread per Morton-sorted point (simple_knn.cu:147-183)
__global__ __launch_bounds__(256) void knn_box_mean_dist(int P, const float* __restrict__ pts, const uint32_t* __restrict__ idx,
                                                         const float* __restrict__ boxes, float* __restrict__ dists) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const uint32_t self = idx[i];
  const float p[3] = {pts[3 * (size_t)self], pts[3 * (size_t)self + 1], pts[3 * (size_t)self + 2]};
  float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  for (int j = max(0, i - 3); j <= min(P - 1, i + 3); j++) {
    if (j == i) continue;
    const uint32_t g = idx[j];
    const float q[3] = {pts[3 * (size_t)g], pts[3 * (size_t)g + 1], pts[3 * (size_t)g + 2]};
    update3(dist2(p, q), best);
  }
  const float reject = best[2];
  best[0] = best[1] = best[2] = FLT_MAX;
  const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
  for (int b = 0; b < nboxes; b++) {
    const float* bx = boxes + 6 * b;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;      // distBoxPoint, simple_knn.cu:119-129
    if (p[0] < bx[0] || p[0] > bx[3]) d0 = fminf(fabsf(p[0] - bx[0]), fabsf(p[0] - bx[3]));
    if (p[1] < bx[1] || p[1] > bx[4]) d1 = fminf(fabsf(p[1] - bx[1]), fabsf(p[1] - bx[4]));
    if (p[2] < bx[2] || p[2] > bx[5]) d2 = fminf(fabsf(p[2] - bx[2]), fabsf(p[2] - bx[5]));
    const float dist = d0 * d0 + d1 * d1 + d2 * d2;
    if (dist > reject || dist > best[2]) continue;
    const int e = min(P, (b + 1) * KNN_BOX);
    for (int j = b * KNN_BOX; j < e; j++) {
      if (j == i) continue;
      const uint32_t g = idx[j];
      const float q[3] = {pts[3 * (size_t)g], pts[3 * (size_t)g + 1], pts[3 * (size_t)g + 2]};
      update3(dist2(p, q), best);
    }
  }
  dists[self] = (best[0] + best[1] + best[2]) / 3.0f;
}

int launch_knn(int P, const float* points, float* meanDists, void* ws, size_t ws_bytes, hipStream_t s) {
  if (P <= 0) return 0;
  if (ws_bytes < knn_workspace_bytes(P)) { set_error("gm_knn: workspace too small (%zu < %zu)", ws_bytes, knn_workspace_bytes(P)); return 3; }
  KnnWs k = KnnWs::from(ws, (size_t)P);
  const int nb = min(1024, (P + 255) / 256);
  hipLaunchKernelGGL(knn_bbox_partial, dim3(nb), dim3(256), 0, s, P, points, k.bbox_partial);
  hipLaunchKernelGGL(knn_bbox_final, dim3(1), dim3(256), 0, s, nb, k.bbox_partial, k.bbox);
  hipLaunchKernelGGL(knn_morton, dim3((P + 255) / 256), dim3(256), 0, s, P, points, k.bbox, k.keys[0]);
  GM_HIP(hipGetLastError());
  int rc = radix_sort_pairs(k.keys, k.idx, k.hist, k.digit_total, (size_t)P, 30, true, 0, s);
  if (rc) return rc;
  const uint32_t* sorted_idx = k.idx[sort_final_slot(30)];
  const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
  hipLaunchKernelGGL(knn_box_minmax, dim3(nboxes), dim3(256), 0, s, P, points, sorted_idx, k.boxes);
  hipLaunchKernelGGL(knn_box_mean_dist, dim3((P + 255) / 256), dim3(256), 0, s, P, points, sorted_idx, k.boxes, meanDists);
  GM_HIP(hipGetLastError());
  return 0;
}

}  // namespace gm
