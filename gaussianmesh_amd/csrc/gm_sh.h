// gm_sh.h -- real spherical-harmonics helpers shared by the per-Gaussian kernels.
// Constants: reference cuda_rasterizer/auxiliary.h:21-38; polynomial: cuda_rasterizer/forward.cu:30-62
// (identical to utils/sh_utils.py:57-112 eval_sh, pinned by tests/golden/sh_eval.npz).
// Include AFTER `#pragma clang fp contract(off)`: the evaluation order is part of the arithmetic contract.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gm {

__device__ static const float SH_C0 = 0.28209479177387814f;
__device__ static const float SH_C1 = 0.4886025119029199f;
__device__ static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                           -1.0925484305920792f, 0.5462742152960396f};
__device__ static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                           0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                           -0.5900435899266435f};

// SH polynomial for one colour channel; s = the 16 coefficients of that channel.  forward.cu:30-62
template <typename F>
__device__ __forceinline__ float sh_channel(int deg, F S, float x, float y, float z) {
  float r = SH_C0 * S(0);
  if (deg > 0) {
    r = r - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      r = r + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) + SH_C2[2] * (2.0f * zz - xx - yy) * S(6) +
          SH_C2[3] * xz * S(7) + SH_C2[4] * (xx - yy) * S(8);
      if (deg > 2) {
        r = r + SH_C3[0] * y * (3.0f * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10) +
            SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11) + SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12) +
            SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) + SH_C3[5] * z * (xx - yy) * S(14) +
            SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
      }
    }
  }
  return r;
}

// loads the first ncoef SH coefficients (x3 channels) of Gaussian idx into sh[48] with 16-byte loads
__device__ __forceinline__ void load_sh(const float* __restrict__ shs, size_t idx, int M, int ncoef, float* sh) {
  const float* base = shs + idx * (size_t)M * 3;
  const int nf = ncoef * 3;
  if ((((size_t)M * 3) & 3) == 0) {            // rows are 16-byte aligned (M = 16 -> 192 B rows)
    const float4* b4 = reinterpret_cast<const float4*>(base);
#pragma unroll
    for (int k = 0; k < 12; k++) {
      if (4 * k < nf) {
        const float4 q = b4[k];
        sh[4 * k] = q.x; sh[4 * k + 1] = q.y; sh[4 * k + 2] = q.z; sh[4 * k + 3] = q.w;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 48; k++)
      if (k < nf) sh[k] = base[k];
  }
}

}  // namespace gm
