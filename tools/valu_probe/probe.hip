// Issue-rate probe for gfx950 (MI355X): how many shader cycles does ONE wave64 instruction of a given class occupy its SIMD for?
// DESIGN.md's "vector-issue ceiling" of the blend kernels needs this constant; rounds 2-4 charged 4 cycles per wave64 vector
// instruction, the hardware guide (MI355X_MICROARCH.md, "Wave scheduling", per-instruction table) says 2.  This program measures it.
//
// Method: 256-thread workgroups (four waves: one per SIMD of a CU), W workgroups per CU on all 256 CUs, W = 1, 2, 4, 8, i.e. W waves
// per SIMD.  Every wave runs ITERS iterations of a block of 64 instructions of one class on INDEPENDENT registers (16 accumulators in
// rotation; "dep": one accumulator, a dependent chain).  Reported per class and W:
//   wave  = s_memtime ticks the wave spent per instruction (what ONE wave sees: issue + dependency latency at W = 1),
//   simd  = launch wall time x shader clock / (wave-instructions per SIMD) = cycles of SIMD time per wave-instruction at that
//           occupancy; its floor over W is the issue cost of the class.
// The shader clock is taken from s_memtime against wall_clock64 (100 MHz) over the same interval.
// Build: hipcc --offload-arch=gfx950 -O2 -o probe probe.hip ; run on the GPU box; output committed as profiles/r05_valu_probe.txt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 1; } } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

enum Cls { FMA, FMA_DEP, MUL, ADD, PK_FMA, PK_MUL, EXP, LOG, RCP, SQRT, CNDMASK, CMP, CMP_CNDMASK, MED3, READLANE, MBCNT, DS_B128_BCAST, DS_B128_LANE, DS_B32_LANE,
           DS_B64_LANE, FWD_BODY, BWD_BODY, NCLS };
static const char* cls_name[NCLS] = {"v_fma_f32 (16 independent)", "v_fma_f32 (dependent chain)", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32",
                                     "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_sqrt_f32", "v_cndmask_b32 (vcc)", "v_cmp_ge_f32 (-> vcc)",
                                     "v_cmp_ge_f32 + v_cndmask_b32 pair", "v_med3_f32", "v_readlane_b32 (-> sgpr)", "v_mbcnt_lo/hi pair",
                                     "ds_read_b128, one address per wave (broadcast)", "ds_read_b128, lane-linear", "ds_read_b32, lane-linear",
                                     "ds_read_b64, lane-linear",
                                     "forward-blend body, cycles per SURVIVOR (15 VALU as compiled here)", "backward-blend phase-1 body, cycles per ENTRY (28 VALU+3 ds_read+1 ds_write)"};
// instructions per block of the class (for the per-instruction figures)
static const int cls_insts[NCLS] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 16, 8};

template <int C>
__global__ __launch_bounds__(256) void probe_kernel(int iters, unsigned long long* __restrict__ out, float seed) {
  __shared__ float4 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_float4(seed, seed + 1.f, seed + 2.f, seed + 3.f);
  __syncthreads();
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float a8 = seed + 8, a9 = seed + 9, a10 = seed + 10, a11 = seed + 11, a12 = seed + 12, a13 = seed + 13, a14 = seed + 14, a15 = seed + 15;
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p0 = {seed, seed}, p1 = {seed + 1, seed}, p2 = {seed + 2, seed}, p3 = {seed + 3, seed}, p4 = {seed + 4, seed}, p5 = {seed + 5, seed}, p6 = {seed + 6, seed}, p7 = {seed + 7, seed};
  float4 q0 = lds[0], q1 = q0, q2 = q0, q3 = q0;
  float b = 0.999f, c = 1e-6f;
  const unsigned lane_addr128 = (threadIdx.x & 63) * 16, lane_addr32 = (threadIdx.x & 63) * 4, lane_addr64 = (threadIdx.x & 63) * 8, zero_addr = 0;
  int s0 = 0;
  const unsigned long long w0 = wall_clock64();
  const unsigned long long t0 = __builtin_readcyclecounter();     // s_memtime
  for (int it = 0; it < iters; it++) {
    if (C == FMA) {
      REP4(asm volatile("v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
                        "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
                        "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
                        "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11),
                          "+v"(a12), "+v"(a13), "+v"(a14), "+v"(a15) : "v"(b), "v"(c));)
    } else if (C == FMA_DEP) {
      REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n" : "+v"(a0) : "v"(b), "v"(c));)
    } else if (C == MUL || C == ADD || C == EXP || C == LOG || C == RCP || C == SQRT || C == MED3) {
#define ONE16(OP, TAIL) \
      REP4(asm volatile(OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" \
                        OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n" \
                        OP " %8, %8" TAIL "\n" OP " %9, %9" TAIL "\n" OP " %10, %10" TAIL "\n" OP " %11, %11" TAIL "\n" \
                        OP " %12, %12" TAIL "\n" OP " %13, %13" TAIL "\n" OP " %14, %14" TAIL "\n" OP " %15, %15" TAIL "\n" \
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11), \
                          "+v"(a12), "+v"(a13), "+v"(a14), "+v"(a15) : "v"(b), "v"(c));)
      if (C == MUL) { ONE16("v_mul_f32", ", %16") }
      else if (C == ADD) { ONE16("v_add_f32", ", %17") }
      else if (C == EXP) { ONE16("v_exp_f32", "") }
      else if (C == LOG) { ONE16("v_log_f32", "") }
      else if (C == RCP) { ONE16("v_rcp_f32", "") }
      else if (C == SQRT) { ONE16("v_sqrt_f32", "") }
      else { ONE16("v_med3_f32", ", %16, %17") }
    } else if (C == PK_FMA || C == PK_MUL) {
#define PK8(OP, TAIL) \
      REP4(REP4(asm volatile(OP " %0, %0, %8" TAIL "\n" OP " %1, %1, %8" TAIL "\n" OP " %2, %2, %8" TAIL "\n" OP " %3, %3, %8" TAIL "\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(p7), "v"(p6));))
      if (C == PK_FMA) { PK8("v_pk_fma_f32", ", %9") } else { PK8("v_pk_mul_f32", "") }
    } else if (C == CNDMASK) {
      asm volatile("v_cmp_ge_f32 vcc, %0, %1" :: "v"(a0), "v"(b) : "vcc");
      REP4(asm volatile("v_cndmask_b32 %0, %0, %16, vcc\n v_cndmask_b32 %1, %1, %16, vcc\n v_cndmask_b32 %2, %2, %16, vcc\n v_cndmask_b32 %3, %3, %16, vcc\n"
                        "v_cndmask_b32 %4, %4, %16, vcc\n v_cndmask_b32 %5, %5, %16, vcc\n v_cndmask_b32 %6, %6, %16, vcc\n v_cndmask_b32 %7, %7, %16, vcc\n"
                        "v_cndmask_b32 %8, %8, %16, vcc\n v_cndmask_b32 %9, %9, %16, vcc\n v_cndmask_b32 %10, %10, %16, vcc\n v_cndmask_b32 %11, %11, %16, vcc\n"
                        "v_cndmask_b32 %12, %12, %16, vcc\n v_cndmask_b32 %13, %13, %16, vcc\n v_cndmask_b32 %14, %14, %16, vcc\n v_cndmask_b32 %15, %15, %16, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11),
                          "+v"(a12), "+v"(a13), "+v"(a14), "+v"(a15) : "v"(b), "v"(c) : "vcc");)
    } else if (C == CMP) {
      REP16(asm volatile("v_cmp_ge_f32 vcc, %0, %4\n v_cmp_ge_f32 vcc, %1, %4\n v_cmp_ge_f32 vcc, %2, %4\n v_cmp_ge_f32 vcc, %3, %4\n" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b) : "vcc");)
    } else if (C == CMP_CNDMASK) {
      REP4(REP4(asm volatile("v_cmp_ge_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_ge_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");
                asm volatile("v_cmp_ge_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_ge_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc");))
      // (32 pairs = 64 instructions: the REP4(REP4()) above is 16 x 4 instructions)
    } else if (C == READLANE) {
      REP16(asm volatile("v_readlane_b32 %0, %1, 3\n v_readlane_b32 %0, %2, 5\n v_readlane_b32 %0, %3, 7\n v_readlane_b32 %0, %4, 9\n" : "=s"(s0) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    } else if (C == MBCNT) {
      REP16(asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n v_mbcnt_hi_u32_b32 %0, -1, %0\n v_mbcnt_lo_u32_b32 %1, -1, 0\n v_mbcnt_hi_u32_b32 %1, -1, %1\n" : "+v"(a0), "+v"(a1));)
    } else if (C == DS_B128_BCAST || C == DS_B128_LANE) {
      const unsigned ad = C == DS_B128_BCAST ? zero_addr : lane_addr128;
      REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(ad) : "memory");)
    } else if (C == DS_B32_LANE) {
      REP16(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(lane_addr32) : "memory");)
    } else if (C == DS_B64_LANE) {
      REP16(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(lane_addr64) : "memory");)
    } else if (C == FWD_BODY) {
      // the forward blend's per-survivor body, C++ as in gm_render.hip blend16 (GM_FWD_SUB = 4: four alpha evaluations interleaved, then the
      // T / C recurrence in list order), 16 survivors per iteration; the compiler emits 11 VALU per survivor (exp, cmp, min, cndmask, mul,
      // sub, cmp, cndmask, cndmask, pk_fma, fma).  Exponents and colours are made opaque per iteration so nothing is hoisted.
      float E[16] = {a0, a1, a2, a3, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a0, a1};
      float T = a4, Cb = a5; v2f Crg = p0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float4 S[4];
#pragma unroll
        for (int t = 0; t < 4; t++) { S[t] = q0; asm volatile("" : "+v"(S[t].x), "+v"(S[t].y), "+v"(S[t].z)); asm volatile("" : "+v"(E[4 * q + t])); }
        float al[4];
#pragma unroll
        for (int t = 0; t < 4; t++) { const float oG = __builtin_amdgcn_exp2f(E[4 * q + t]); al[t] = (oG >= 1.0f / 255.0f) ? fminf(0.99f, oG) : 0.0f; }
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const float wa = al[t] * T, tt = T - wa;
          const bool stop = tt < 0.0001f;
          const float w = stop ? 0.0f : wa;
          T = stop ? -__builtin_fabsf(T) : tt;
          const v2f rg = {S[t].x, S[t].y}, ww = {w, w};
          Crg = rg * ww + Crg; Cb += S[t].z * w;
        }
      }
      a4 = T; a5 = Cb; p0 = Crg;
    } else if (C == BWD_BODY) {
      // the backward walk's phase-1 entry body, C++ as in gm_render.hip render_bwd_kernel (staged record from LDS at a uniform address,
      // pixel-relative exponent, alpha, 1/(1 - alpha), the A recurrence, (w, h) to an LDS row), 8 entries per iteration
      float T = a4, A = a5; const v2f pix = p1;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        float4 RA = lds[3 * j], RB = lds[3 * j + 1], RC = lds[3 * j + 2];
        asm volatile("" : "+v"(RA.x), "+v"(RA.y), "+v"(RA.z), "+v"(RA.w)); asm volatile("" : "+v"(RB.x), "+v"(RB.y), "+v"(RB.z), "+v"(RB.w)); asm volatile("" : "+v"(RC.x), "+v"(RC.y));
        const v2f xy = {RA.x, RA.y}, ac = {RA.z, RA.w};
        const v2f d = xy - pix; v2f q = ac * d; q.x = __builtin_fmaf(RB.x, d.y, q.x); const v2f r = q * d;
        const float e = r.x + r.y;
        const float G = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(e), 0.0f, 1.0f);
        const float oG = RB.y * G;
        const bool valid = (__float_as_int(RC.y) < __float_as_int(a6)) && (oG >= 1.0f / 255.0f);
        const float oGe = valid ? oG : 0.0f;
        const float al = __builtin_amdgcn_fmed3f(oGe, 0.0f, 0.99f);
        const float inv = __builtin_amdgcn_rcpf(1.f - al);
        const float cd = __builtin_fmaf(RC.x, a7, __builtin_fmaf(RB.w, a8, RB.z * a9));
        T = T * inv;
        const float wv = al * T;
        const float dL = T * cd - A * inv;
        A = __builtin_fmaf(wv, cd, A);
        reinterpret_cast<float2*>(&lds[64])[(threadIdx.x & 63) + 65 * j] = make_float2(wv, oGe * dL);
      }
      a4 = T; a5 = A;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long w1 = wall_clock64();
  // keep every register alive
  float keep = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y + q0.x + q1.y + q2.z + q3.w + (float)s0;
  if (keep == 123.456f) out[3] = 1;
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    out[4 + 2 * w] = t1 - t0; out[5 + 2 * w] = w1 - w0;
  }
}

typedef void (*kern_t)(int, unsigned long long*, float);
template <int C> static kern_t get() { return probe_kernel<C>; }
static kern_t kernels[NCLS] = {get<0>(), get<1>(), get<2>(), get<3>(), get<4>(), get<5>(), get<6>(), get<7>(), get<8>(), get<9>(), get<10>(), get<11>(), get<12>(),
                               get<13>(), get<14>(), get<15>(), get<16>(), get<17>(), get<18>(), get<19>(), get<20>(), get<21>()};

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 2048;
  hipDeviceProp_t prop; HC(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# %s, %d CUs, clockRate %d kHz; %d iterations of a %d-instruction block per wave\n", prop.name, cus, prop.clockRate, iters, 64);
  printf("# wave = s_memtime ticks per instruction seen by one wave (median over waves); simd = launch time x shader clock x SIMDs / wave-instructions\n");
  unsigned long long* out; const size_t slots = 4 + 2 * (size_t)cus * 8 * 4;
  HC(hipMalloc(&out, slots * 8));
  std::vector<unsigned long long> h(slots);
  hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
  printf("%-62s %3s %10s %10s %10s %9s\n", "class", "W", "wave cyc", "simd cyc", "launch us", "clock MHz");
  for (int c = 0; c < NCLS; c++) {
    for (int W = 1; W <= 8; W *= 2) {
      const int grid = cus * W;
      HC(hipMemset(out, 0, slots * 8));
      hipLaunchKernelGGL(kernels[c], dim3(grid), dim3(256), 0, 0, iters / 8, out, 0.5f);      // warm-up
      HC(hipDeviceSynchronize());
      HC(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(kernels[c], dim3(grid), dim3(256), 0, 0, iters, out, 0.5f);
      HC(hipEventRecord(e1, 0));
      HC(hipEventSynchronize(e1));
      float ms = 0; HC(hipEventElapsedTime(&ms, e0, e1));
      HC(hipMemcpy(h.data(), out, slots * 8, hipMemcpyDeviceToHost));
      std::vector<double> tk, mhz;
      for (int w = 0; w < grid * 4; w++) { tk.push_back((double)h[4 + 2 * w]); mhz.push_back((double)h[4 + 2 * w] / ((double)h[5 + 2 * w] / 100.0)); }
      std::sort(tk.begin(), tk.end()); std::sort(mhz.begin(), mhz.end());
      const double ninst = (double)iters * cls_insts[c];
      const double clock = mhz[mhz.size() / 2];                       // MHz, from s_memtime vs the 100-MHz wall clock
      const double simd_cyc = (ms * 1e3 * clock) / (ninst * W);      // cycles of one SIMD per wave-instruction
      printf("%-62s %3d %10.2f %10.2f %10.1f %9.0f\n", cls_name[c], W, tk[tk.size() / 2] / ninst, simd_cyc, ms * 1e3, clock);
    }
  }
  return 0;
}
