// gm_stage.h -- cooperative, fully coalesced movement of AoS rows between HBM and LDS.
//
// The reference boundary is array-of-structures ([P,3], [P,9], [P,16,3] rows of 12 / 36 / 192 bytes).  A thread that
// walks its own 192-byte SH row with 16-byte loads makes every wave-level load instruction touch 64 different cache
// lines, each line being revisited by 8 later instructions after it has left the 32 KiB L1: the L2->L1 traffic is ~8x
// the data.  Instead the workgroup copies its contiguous slab of rows with consecutive lanes reading consecutive 16
// bytes (each line fetched once, 1 KiB per wave instruction) into LDS with an odd row stride, and each thread then
// reads its row from LDS conflict-free (stride 49 / 9 / 3 dwords are all coprime with the 32 banks).
#pragma once
#include <hip/hip_runtime.h>

namespace gm {

// rows [row0, row0+nrows) of ROW floats -> lds[r * LSTRIDE + c].  g must be 16-byte aligned and row0*ROW a multiple of 4.
template <int ROW, int LSTRIDE, int THREADS>
__device__ __forceinline__ void stage_rows(const float* __restrict__ g, size_t row0, int nrows, float* __restrict__ lds) {
  const float* src = g + row0 * ROW;
  const int total = nrows * ROW, total4 = total & ~3;
  for (int i = threadIdx.x * 4; i < total4; i += THREADS * 4) {
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; c++) lds[((i + c) / ROW) * LSTRIDE + ((i + c) % ROW)] = e[c];
  }
  if ((int)threadIdx.x < total - total4) {
    const int e = total4 + threadIdx.x;
    lds[(e / ROW) * LSTRIDE + (e % ROW)] = src[e];
  }
}

// lds[r * LSTRIDE + c] -> rows [row0, row0+nrows) of ROW floats, coalesced 16-byte stores
template <int ROW, int LSTRIDE, int THREADS>
__device__ __forceinline__ void unstage_rows(float* __restrict__ g, size_t row0, int nrows, const float* __restrict__ lds) {
  float* dst = g + row0 * ROW;
  const int total = nrows * ROW, total4 = total & ~3;
  for (int i = threadIdx.x * 4; i < total4; i += THREADS * 4) {
    float e[4];
#pragma unroll
    for (int c = 0; c < 4; c++) e[c] = lds[((i + c) / ROW) * LSTRIDE + ((i + c) % ROW)];
    *reinterpret_cast<float4*>(dst + i) = make_float4(e[0], e[1], e[2], e[3]);
  }
  if ((int)threadIdx.x < total - total4) {
    const int e = total4 + threadIdx.x;
    dst[e] = lds[(e / ROW) * LSTRIDE + (e % ROW)];
  }
}

// 16-byte-granule variants for rows that are a whole number of float4 (the 192-byte SH rows: G = 12 granules).
// Row r, granule c lives at LDS granule r * LG + c with LG odd (13), so that a ds_read_b128 / ds_write_b128 of
// "granule c of my row" by 16 consecutive lanes touches 16 different 16-byte bank columns: conflict-free, one LDS
// instruction per granule instead of four scalar ones, and one divide per granule in the copy loop instead of four.
template <int G, int LG, int THREADS>
__device__ __forceinline__ void stage_rows16(const float* __restrict__ g, size_t row0, int nrows, float4* __restrict__ lds) {
  const float4* src = reinterpret_cast<const float4*>(g) + row0 * G;
  const int total = nrows * G;
  for (int i = threadIdx.x; i < total; i += THREADS) {
    const int r = i / G, c = i - r * G;
    lds[r * LG + c] = src[i];
  }
}
template <int G, int LG, int THREADS>
__device__ __forceinline__ void unstage_rows16(float* __restrict__ g, size_t row0, int nrows, const float4* __restrict__ lds) {
  float4* dst = reinterpret_cast<float4*>(g) + row0 * G;
  const int total = nrows * G;
  for (int i = threadIdx.x; i < total; i += THREADS) {
    const int r = i / G, c = i - r * G;
    dst[i] = lds[r * LG + c];
  }
}
// plain linear copy of nfloats (multiple-of-4 part as float4) - rows with a stride coprime to the bank count
// (9 or 3 dwords) can be read straight out of the linear image
template <int THREADS>
__device__ __forceinline__ void stage_linear(const float* __restrict__ src, int nfloats, float* __restrict__ lds) {
  const int n4 = nfloats >> 2;
  for (int i = threadIdx.x; i < n4; i += THREADS) reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(src)[i];
  const int t = (n4 << 2) + threadIdx.x;
  if (t < nfloats) lds[t] = src[t];
}

// LDS-DMA (global_load_lds_dwordx4): lane l of the wave copies 16 bytes from its own global address g to
// lds_base + 16 l (lds_base is wave-uniform); no staging registers, completion is counted by vmcnt.
__device__ __forceinline__ void dma16(const void* g, void* lds_base) {
  __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)g, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace gm
