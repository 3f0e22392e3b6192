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
           DS_B64_LANE, FWD_BODY, BWD_BODY,
           AND_B32, OR_B32, NOT_B32, ASHR_I32, SUB_U32, MAX_U32, MIN_F32, MAX_F32, MOV_B32, BFI_B32, AND_OR_B32, CNDMASK_SGPR, CNDMASK_CONST0, MUL_SGPR, CMP_SGPR, CMPX, FMAC_F32, FWD_BODY_B, FWD_BODY_A, FWD_GROUP_V0, FWD_GROUP_VA, NCLS };
static const char* cls_name[NCLS] = {"v_fma_f32 (16 independent)", "v_fma_f32 (dependent chain)", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32",
                                     "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_sqrt_f32", "v_cndmask_b32 (vcc)", "v_cmp_ge_f32 (-> vcc)",
                                     "v_cmp_ge_f32 + v_cndmask_b32 pair", "v_med3_f32", "v_readlane_b32 (-> sgpr)", "v_mbcnt_lo/hi pair",
                                     "ds_read_b128, one address per wave (broadcast)", "ds_read_b128, lane-linear", "ds_read_b32, lane-linear",
                                     "ds_read_b64, lane-linear",
                                     "forward-blend body, cycles per SURVIVOR (15 VALU as compiled here)", "backward-blend phase-1 body, cycles per ENTRY (28 VALU+3 ds_read+1 ds_write)",
                                     "v_and_b32", "v_or_b32", "v_not_b32", "v_ashrrev_i32 (31)", "v_sub_u32", "v_max_u32", "v_min_f32", "v_max_f32", "v_mov_b32", "v_bfi_b32",
                                     "v_and_or_b32", "v_cndmask_b32_e64 (mask in an SGPR pair)", "v_cndmask_b32 v, 0, v, vcc", "v_mul_f32 with an SGPR operand",
                                     "v_cmp_ge_f32_e64 (-> SGPR pair)", "v_cmpx_ge_f32 (-> exec, always true)", "v_fmac_f32 (VOP2)",
                                     "forward-blend body B: selects as integer masks (C++), cycles per SURVIVOR",
                                     "forward-blend body A: stop through EXEC (asm), cycles per SURVIVOR",
                                     "forward GROUP as in the kernel (3 MFMA + 16 survivors, colours from LDS), compare + select: cycles per SURVIVOR",
                                     "forward GROUP, stop through EXEC + alpha mask + clamp-scaled exponent (asm): cycles per SURVIVOR"};
// instructions per block of the class (for the per-instruction figures)
static const int cls_insts[NCLS] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 16, 8, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 16, 16, 16, 16};

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
    } else if (C == AND_B32 || C == OR_B32 || C == SUB_U32 || C == MAX_U32 || C == MIN_F32 || C == MAX_F32 || C == FMAC_F32) {
      if (C == AND_B32) { ONE16("v_and_b32", ", %16") } else if (C == OR_B32) { ONE16("v_or_b32", ", %16") } else if (C == SUB_U32) { ONE16("v_sub_u32", ", %16") }
      else if (C == MAX_U32) { ONE16("v_max_u32", ", %16") } else if (C == MIN_F32) { ONE16("v_min_f32", ", %16") } else if (C == MAX_F32) { ONE16("v_max_f32", ", %16") }
      else { ONE16("v_fmac_f32", ", %16") }
    } else if (C == NOT_B32) { ONE16("v_not_b32", "")
    } else if (C == BFI_B32) { ONE16("v_bfi_b32", ", %16, %17")
    } else if (C == AND_OR_B32) { ONE16("v_and_or_b32", ", %16, %17")
    } else if (C == ASHR_I32 || C == MOV_B32 || C == CNDMASK_CONST0) {
#define PRE16(OP, PRE, TAIL) \
      REP4(asm volatile(OP " %0, " PRE "%0" TAIL "\n" OP " %1, " PRE "%1" TAIL "\n" OP " %2, " PRE "%2" TAIL "\n" OP " %3, " PRE "%3" TAIL "\n" \
                        OP " %4, " PRE "%4" TAIL "\n" OP " %5, " PRE "%5" TAIL "\n" OP " %6, " PRE "%6" TAIL "\n" OP " %7, " PRE "%7" TAIL "\n" \
                        OP " %8, " PRE "%8" TAIL "\n" OP " %9, " PRE "%9" TAIL "\n" OP " %10, " PRE "%10" TAIL "\n" OP " %11, " PRE "%11" TAIL "\n" \
                        OP " %12, " PRE "%12" TAIL "\n" OP " %13, " PRE "%13" TAIL "\n" OP " %14, " PRE "%14" TAIL "\n" OP " %15, " PRE "%15" TAIL "\n" \
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11), \
                          "+v"(a12), "+v"(a13), "+v"(a14), "+v"(a15) : "v"(b), "v"(c) : "vcc");)
      if (C == ASHR_I32) { PRE16("v_ashrrev_i32", "31, ", "") }
      else if (C == MOV_B32) {
        REP16(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(b));)
      } else {
        asm volatile("v_cmp_ge_f32 vcc, %0, %1" :: "v"(a0), "v"(b) : "vcc");
        PRE16("v_cndmask_b32", "0, ", ", vcc")
      }
    } else if (C == CNDMASK_SGPR) {
      unsigned long long msk;
      asm volatile("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(msk) : "v"(a0), "v"(b));
      REP4(asm volatile("v_cndmask_b32_e64 %0, %0, %16, %17\n v_cndmask_b32_e64 %1, %1, %16, %17\n v_cndmask_b32_e64 %2, %2, %16, %17\n v_cndmask_b32_e64 %3, %3, %16, %17\n"
                        "v_cndmask_b32_e64 %4, %4, %16, %17\n v_cndmask_b32_e64 %5, %5, %16, %17\n v_cndmask_b32_e64 %6, %6, %16, %17\n v_cndmask_b32_e64 %7, %7, %16, %17\n"
                        "v_cndmask_b32_e64 %8, %8, %16, %17\n v_cndmask_b32_e64 %9, %9, %16, %17\n v_cndmask_b32_e64 %10, %10, %16, %17\n v_cndmask_b32_e64 %11, %11, %16, %17\n"
                        "v_cndmask_b32_e64 %12, %12, %16, %17\n v_cndmask_b32_e64 %13, %13, %16, %17\n v_cndmask_b32_e64 %14, %14, %16, %17\n v_cndmask_b32_e64 %15, %15, %16, %17\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11),
                          "+v"(a12), "+v"(a13), "+v"(a14), "+v"(a15) : "v"(b), "s"(msk));)
    } else if (C == MUL_SGPR) {
      float sb = __builtin_amdgcn_readfirstlane(__float_as_int(b)) ? 0.999f : 0.5f;
      REP4(asm volatile("v_mul_f32 %0, %16, %0\n v_mul_f32 %1, %16, %1\n v_mul_f32 %2, %16, %2\n v_mul_f32 %3, %16, %3\n"
                        "v_mul_f32 %4, %16, %4\n v_mul_f32 %5, %16, %5\n v_mul_f32 %6, %16, %6\n v_mul_f32 %7, %16, %7\n"
                        "v_mul_f32 %8, %16, %8\n v_mul_f32 %9, %16, %9\n v_mul_f32 %10, %16, %10\n v_mul_f32 %11, %16, %11\n"
                        "v_mul_f32 %12, %16, %12\n v_mul_f32 %13, %16, %13\n v_mul_f32 %14, %16, %14\n v_mul_f32 %15, %16, %15\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11),
                          "+v"(a12), "+v"(a13), "+v"(a14), "+v"(a15) : "s"(sb));)
    } else if (C == CMP_SGPR) {
      unsigned long long m0, m1, m2, m3;
      REP16(asm volatile("v_cmp_ge_f32_e64 %0, %4, %8\n v_cmp_ge_f32_e64 %1, %5, %8\n v_cmp_ge_f32_e64 %2, %6, %8\n v_cmp_ge_f32_e64 %3, %7, %8\n"
                         : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b));)
      s0 += (int)(m0 ^ m1 ^ m2 ^ m3);
    } else if (C == CMPX) {
      // |x| >= 0 is true for every non-NaN lane: EXEC stays full
      REP16(asm volatile("v_cmpx_ge_f32_e64 vcc, |%0|, 0\n v_cmpx_ge_f32_e64 vcc, |%1|, 0\n v_cmpx_ge_f32_e64 vcc, |%2|, 0\n v_cmpx_ge_f32_e64 vcc, |%3|, 0\n" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
    } else if (C == FWD_BODY_B) {
      // the forward body with every select spelled as an integer mask (positive floats compare like their bit patterns):
      //   alpha:  m1 = ((C1 - bits(oG)) >> 31) [all ones when oG >= 1/255];  al = min(0.99, oG) & m1
      //   stop:   live transmittance Tl (0 once stopped) and final transmittance Tf;  tt = Tl - al Tl;  m2 = ((C2 - bits(tt)) >> 31) [tt >= 1e-4];
      //           w = (al Tl) & m2;  Tl = tt & m2;  Tf = min(Tf, tt | ~m2)  (v_min_f32 returns the other operand for a NaN pattern)
      float E[16] = {a0, a1, a2, a3, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a0, a1};
      float Tl = a4, Tf = a4, Cb = a5; v2f Crg = p0;
      const int C1 = 0x3B808081 - 1, C2 = 0x38D1B717 - 1;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float4 S[4];
#pragma unroll
        for (int t = 0; t < 4; t++) { S[t] = q0; asm volatile("" : "+v"(S[t].x), "+v"(S[t].y), "+v"(S[t].z)); asm volatile("" : "+v"(E[4 * q + t])); }
        float al[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const float oG = __builtin_amdgcn_exp2f(E[4 * q + t]);
          int m1; asm("v_sub_u32 %0, %1, %2\n v_ashrrev_i32 %0, 31, %0" : "=&v"(m1) : "v"(C1), "v"(__float_as_int(oG)));
          al[t] = __int_as_float(__float_as_int(fminf(0.99f, oG)) & m1);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const float wa = al[t] * Tl, tt = Tl - wa;
          int m2; asm("v_sub_u32 %0, %1, %2\n v_ashrrev_i32 %0, 31, %0" : "=&v"(m2) : "v"(C2), "v"(__float_as_int(tt)));
          const float w = __int_as_float(__float_as_int(wa) & m2);
          Tl = __int_as_float(__float_as_int(tt) & m2);
          int nm; asm("v_not_b32 %0, %1" : "=v"(nm) : "v"(m2));
          Tf = fminf(Tf, __int_as_float(__float_as_int(tt) | nm));
          Crg.x = __builtin_fmaf(S[t].x, w, Crg.x); Crg.y = __builtin_fmaf(S[t].y, w, Crg.y); Cb = __builtin_fmaf(S[t].z, w, Cb);
        }
      }
      a4 = Tl + Tf; a5 = Cb; p0 = Crg;
    } else if (C == FWD_BODY_A) {
      // the forward body with the stop decision on the EXECUTION MASK: a pixel whose T (1 - alpha) falls below 1e-4 leaves EXEC (v_cmpx) and
      // takes nothing any more; T then simply keeps its final value.  16 survivors in one asm block (EXEC saved and restored around it).
      float E[16] = {a0, a1, a2, a3, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a0, a1};
      float T = a4, Cb = a5, Cr = p0.x, Cg = p0.y;
      const int C1 = 0x3B808081 - 1; const float thr = 0.0001f, cap = 0.99f;
      unsigned long long saved;
      asm volatile("s_mov_b64 %0, exec" : "=s"(saved));
#pragma unroll
      for (int t = 0; t < 16; t++) {
        float Sx = q0.x, Sy = q0.y, Sz = q0.z;
        asm volatile("" : "+v"(Sx), "+v"(Sy), "+v"(Sz)); asm volatile("" : "+v"(E[t]));
        float oG, m, wa, tt;
        asm volatile("v_exp_f32 %4, %8\n v_sub_u32 %5, %9, %4\n v_ashrrev_i32 %5, 31, %5\n v_min_f32 %4, %10, %4\n v_and_b32 %4, %5, %4\n"
                     "v_mul_f32 %6, %4, %0\n v_sub_f32 %7, %0, %6\n v_cmpx_le_f32_e64 vcc, %11, %7\n v_mov_b32 %0, %7\n"
                     "v_fmac_f32 %1, %12, %6\n v_fmac_f32 %2, %13, %6\n v_fmac_f32 %3, %14, %6\n"
                     : "+v"(T), "+v"(Cr), "+v"(Cg), "+v"(Cb), "=&v"(oG), "=&v"(m), "=&v"(wa), "=&v"(tt)
                     : "v"(E[t]), "v"(C1), "v"(cap), "v"(thr), "v"(Sx), "v"(Sy), "v"(Sz) : "vcc");
      }
      asm volatile("s_mov_b64 exec, %0" :: "s"(saved));
      a4 = T; a5 = Cb; p0.x = Cr; p0.y = Cg;
    } else if (C == FWD_GROUP_V0 || C == FWD_GROUP_VA) {
      // One group of the forward blend as the kernel runs it: the A operand from LDS, three chained v_mfma_f32_32x32x2_f32, then sixteen
      // survivors whose (r, g, b, -) come from LDS as one ds_read_b128 each (uniform address).  V0: gm_render.hip blend16 as it stands
      // (image-only build).  VA: the alpha >= 1/255 skip as an integer mask, min(0.99, .) folded into the exponent (clamp bit of v_exp on
      // e' - log2(0.99), the factor 0.99 in the colours and in T - 0.99 w'), the stop test through EXEC (v_cmpx: a stopped pixel leaves
      // the wave's execution mask for the rest of the group; T keeps its final value).
      typedef float v16f __attribute__((ext_vector_type(16)));
      const int lane = threadIdx.x & 63;
      const float* ctg = reinterpret_cast<const float*>(lds) + lane;
      const float A0 = ctg[0], A1 = ctg[64], A2 = ctg[128];
      v16f E = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      E = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, a6, E, 0, 0, 0);
      E = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, a7, E, 0, 0, 0);
      E = __builtin_amdgcn_mfma_f32_32x32x2f32(A2, a8, E, 0, 0, 0);
      const float4* sb = &lds[64];
      if (C == FWD_GROUP_V0) {
        float T = a4, Cb = a5; v2f Crg = p0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          float4 S[4];
#pragma unroll
          for (int t = 0; t < 4; t++) S[t] = sb[(it & 3) * 16 + 4 * q + t];
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
      } else {
        float T = a4, Cb = a5, Cr = p0.x, Cg = p0.y;
        unsigned long long saved;
        asm volatile("s_mov_b64 %0, exec" : "=s"(saved));
#pragma unroll
        for (int q = 0; q < 4; q++) {
          float4 S[4];
#pragma unroll
          for (int t = 0; t < 4; t++) S[t] = sb[(it & 3) * 16 + 4 * q + t];
          float x0, x1, x2, x3, m0, m1, m2, m3, tt;
          // 0x3B81CDC6 = bits(1/255 / 0.99) : oG' = 2^(e' - log2 0.99) clamped to 1 is alpha / 0.99
          asm volatile(
              "v_exp_f32_e64 %0, %9 clamp\n v_exp_f32_e64 %1, %10 clamp\n v_exp_f32_e64 %2, %11 clamp\n v_exp_f32_e64 %3, %12 clamp\n"
              "v_sub_u32 %4, 0x3b81cdc5, %0\n v_sub_u32 %5, 0x3b81cdc5, %1\n v_sub_u32 %6, 0x3b81cdc5, %2\n v_sub_u32 %7, 0x3b81cdc5, %3\n"
              "v_ashrrev_i32 %4, 31, %4\n v_ashrrev_i32 %5, 31, %5\n v_ashrrev_i32 %6, 31, %6\n v_ashrrev_i32 %7, 31, %7\n"
              "v_and_b32 %0, %4, %0\n v_and_b32 %1, %5, %1\n v_and_b32 %2, %6, %2\n v_and_b32 %3, %7, %3\n"
              : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3), "=&v"(tt)
              : "v"(E[4 * q]), "v"(E[4 * q + 1]), "v"(E[4 * q + 2]), "v"(E[4 * q + 3]));
#define SURV(X, SX, SY, SZ) \
          asm volatile("v_mul_f32 %0, %0, %1\n v_fma_f32 %5, %0, %6, %1\n v_cmpx_le_f32_e64 vcc, %7, %5\n v_mov_b32 %1, %5\n" \
                       "v_fmac_f32 %2, %8, %0\n v_fmac_f32 %3, %9, %0\n v_fmac_f32 %4, %10, %0\n" \
                       : "+v"(X), "+v"(T), "+v"(Cr), "+v"(Cg), "+v"(Cb), "=&v"(tt) : "v"(b), "v"(c), "v"(SX), "v"(SY), "v"(SZ) : "vcc")
          SURV(x0, S[0].x, S[0].y, S[0].z); SURV(x1, S[1].x, S[1].y, S[1].z); SURV(x2, S[2].x, S[2].y, S[2].z); SURV(x3, S[3].x, S[3].y, S[3].z);
        }
        asm volatile("s_mov_b64 exec, %0" :: "s"(saved));
        a4 = T; a5 = Cb; p0.x = Cr; p0.y = Cg;
      }
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
                               get<13>(), get<14>(), get<15>(), get<16>(), get<17>(), get<18>(), get<19>(), get<20>(), get<21>(),
                               get<22>(), get<23>(), get<24>(), get<25>(), get<26>(), get<27>(), get<28>(), get<29>(), get<30>(), get<31>(), get<32>(), get<33>(), get<34>(),
                               get<35>(), get<36>(), get<37>(), get<38>(), get<39>(), get<40>(), get<41>(), get<42>()};

__global__ void nan_min_kernel(float* out) {      // what v_min_f32 / v_max_f32 make of the all-ones NaN pattern (forward body B relies on min(x, NaN) = x)
  const float nan_ = __int_as_float(0xFFFFFFFF), x = 0.25f;
  float r0, r1, r2;
  asm volatile("v_min_f32 %0, %3, %4\n v_min_f32 %1, %4, %3\n v_max_f32 %2, %3, %4" : "=v"(r0), "=v"(r1), "=v"(r2) : "v"(x), "v"(nan_));
  out[0] = r0; out[1] = r1; out[2] = r2;
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 2048;
  const int first_cls = argc > 2 ? atoi(argv[2]) : 0;
  hipDeviceProp_t prop; HC(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# %s, %d CUs, clockRate %d kHz; %d iterations of a %d-instruction block per wave\n", prop.name, cus, prop.clockRate, iters, 64);
  printf("# wave = s_memtime ticks per instruction seen by one wave (median over waves); simd = launch time x shader clock x SIMDs / wave-instructions\n");
  unsigned long long* out; const size_t slots = 4 + 2 * (size_t)cus * 8 * 4;
  HC(hipMalloc(&out, slots * 8));
  std::vector<unsigned long long> h(slots);
  hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
  printf("%-62s %3s %10s %10s %10s %9s\n", "class", "W", "wave cyc", "simd cyc", "launch us", "clock MHz");
  {
    float* nm; HC(hipMalloc(&nm, 16)); hipLaunchKernelGGL(nan_min_kernel, dim3(1), dim3(64), 0, 0, nm); float hn[3]; HC(hipMemcpy(hn, nm, 12, hipMemcpyDeviceToHost));
    printf("# v_min_f32(0.25, NaN) = %g, v_min_f32(NaN, 0.25) = %g, v_max_f32(0.25, NaN) = %g\n", hn[0], hn[1], hn[2]);
  }
  for (int c = first_cls; c < NCLS; c++) {
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
