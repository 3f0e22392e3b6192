// Layout probe for v_mfma_f32_32x32x2_f32 on gfx950: D (32x32) = A (32x2) * B (2x32) + C, one wave, as the forward blend uses it
// (gm_render.hip: rows = (pixel half, survivor), columns = the 32 pixels of a half, k = two monomials of the exponent).
// Hypothesis: lane l holds A[i = l % 32][k = l / 32] and B[k = l / 32][j = l % 32]; register r (0..15) of lane l holds
// D[i = 8 * (r / 4) + 4 * (l / 32) + r % 4][j = l % 32].  Also checked: an output element depends only on its own row of A and
// column of B (two identical rows give bit-identical results with arbitrary floats), and three chained k = 2 steps agree with the
// float32 sum of the six products to a few ulp.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void probe(const float* a_in, const float* b_in, float* d_out, int steps) {
  const int l = threadIdx.x;
  v16f c = {0};
  for (int s = 0; s < steps; s++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a_in[s * 64 + l], b_in[s * 64 + l], c, 0, 0, 0);
  for (int r = 0; r < 16; r++) d_out[l * 16 + r] = c[r];
}
int main() {
  const int steps = 3, K = 2 * steps;
  static float A[32][6], B[6][32], ha[3 * 64], hb[3 * 64], hd[64 * 16];
  srand(1);
  for (int i = 0; i < 32; i++) for (int k = 0; k < K; k++) A[i][k] = (float)rand() / RAND_MAX * 20.f - 10.f;
  for (int k = 0; k < K; k++) for (int j = 0; j < 32; j++) B[k][j] = (float)rand() / RAND_MAX * 8.f - 4.f;
  for (int k = 0; k < K; k++) { A[5][k] = A[22][k]; }                       // two identical rows
  for (int s = 0; s < steps; s++) for (int l = 0; l < 64; l++) { ha[s * 64 + l] = A[l % 32][2 * s + l / 32]; hb[s * 64 + l] = B[2 * s + l / 32][l % 32]; }
  float *da, *db, *dd;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, sizeof(hd));
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd, steps);
  hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
  static float D[32][32];
  int bad = 0; double worst = 0;
  for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) {
    const int i = 8 * (r / 4) + 4 * (l / 32) + r % 4, j = l % 32;
    double want = 0, mag = 0; for (int k = 0; k < K; k++) { want += (double)A[i][k] * B[k][j]; mag += fabs((double)A[i][k] * B[k][j]); }
    D[i][j] = hd[l * 16 + r];
    const double e = fabs(hd[l * 16 + r] - want) / mag;
    if (e > 1e-6) bad++;
    if (e > worst) worst = e;
  }
  int rowdiff = 0;
  for (int j = 0; j < 32; j++) if (D[5][j] != D[22][j]) rowdiff++;
  printf("mfma_f32_32x32x2f32 layout hypothesis: %s (%d mismatches), worst error / sum|terms| = %.3g (2^-24 = 6e-8), identical rows bit-identical: %s\n",
         bad ? "WRONG" : "confirmed", bad, worst, rowdiff ? "NO" : "yes");
  return bad != 0 || rowdiff != 0;
}
