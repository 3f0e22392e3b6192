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

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace gm
