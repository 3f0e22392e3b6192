// Layout probe for v_mfma_f32_16x16x4_f32 on gfx950: D = A (16x4) * B (4x16), one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a_in, const float* b_in, float* d_out) {
  const int l = threadIdx.x;
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a_in[l], b_in[l], c, 0, 0, 0);
  for (int r = 0; r < 4; r++) d_out[l * 4 + r] = c[r];
}
int main() {
  float ha[64], hb[64], hd[256];
  // hypothesis: lane l holds A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16]; D[i = 4 * (l / 16) + r][j = l % 16] in vgpr r
  float A[16][4], B[4][16];
  for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i][k] = (float)(1 + i * 7 + k * 3);
  for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k][j] = (float)(2 + k * 5 + j);
  for (int l = 0; l < 64; l++) { ha[l] = A[l % 16][l / 16]; hb[l] = B[l / 16][l % 16]; }
  float *da, *db, *dd;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
  hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
    const int i = 4 * (l / 16) + r, j = l % 16;
    float want = 0; for (int k = 0; k < 4; k++) want += A[i][k] * B[k][j];
    if (hd[l * 4 + r] != want) bad++;
  }
  printf("mfma_f32_16x16x4f32 layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
  return bad != 0;
}
