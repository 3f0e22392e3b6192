// gm_render.hip -- per-tile alpha blending, forward and backward.
//
// Replaces (reference, RAST = gaussian_renderer/diff_gaussian_rasterizater/cuda_rasterizer):
//   RAST/forward.cu:261-374   renderCUDA (forward)
//   RAST/backward.cu:399-557  renderCUDA (backward)
//
// CDNA4 mapping: the reference gives each 16x16 tile to a 256-thread block that stages
// 256-entry batches in shared memory behind two block barriers and lets every pixel thread re-read the batch from LDS
// (and the colour from global memory).  Here a 16x16 tile is four independent wave64s, each owning one 8x8 pixel
// quadrant (one pixel per lane, lane = y * 8 + x) and launched as a one-wave workgroup:
//   * the wave walks the list of its PARENT tile (gm_common.h: emission policies); a two-stage front end (Gather, FwdLds /
//     BwdLds below) turns it into dense batches of up to 64 candidate records: a scan of the contiguous (key, id) stream picks
//     the entries whose key carries the tile's child bit, their records are gathered into registers two iterations ahead of
//     their use;
//   * while lane j holds candidate j it tests, once per batch and for all 64 in parallel, whether the entry can reach
//     alpha >= 1/255 anywhere inside the bounding box of the quadrant's still-live pixels (exact minimum of the conic's
//     quadratic form over the rectangle, with a rounding margin).  The cull is conservative: a culled entry would have been
//     skipped by every live pixel (alpha < 1/255), so results and n_contrib are unchanged;
//   * forward: the survivors' exponents come from the matrix core, 16 survivors x 64 pixels per three chained
//     v_mfma_f32_32x32x2_f32 (see render_fwd_kernel); the vector ALU keeps exp, the alpha decisions and the T / C recurrence;
//   * backward: a scalar suffix recurrence per pixel, then (weight, h) per pixel through an LDS slot matrix into a row-parallel
//     fold of the nine gradient sums and one atomic instruction per seven 36-byte records (see render_bwd_kernel);
//   * no s_barrier; "is every pixel done" is a ballot instead of __syncthreads_count.
// Discrete semantics are the reference's - skip alpha < 1/255, stop (without applying the entry) when T (1 - alpha) < 1e-4,
// n_contrib = 1-based list position of the last accepted entry - except that "skip power > 0" is "power := min(power, 0)"
// (see render_fwd_kernel: the two differ only where the reference's own rounding decides).
// FMA contraction is allowed here and exp() is v_exp_f32 on power * log2(e); see DESIGN.md for the tolerance argument.
#include "gm_common.h"
#include "gm_cull.h"
#include "gm_tile_order.h"
#include <cstdlib>

namespace gm {

#define LOG2E 1.4426950408889634f

__device__ __forceinline__ float bcast(float v, int j) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}

// Workgroup -> 16-px tile, its parent tile (whose list it walks) and its bit in the instance keys' child mask.
// Consecutive workgroup ids go to consecutive XCDs (8 of them, each with its own L2), so the 2^s x 2^s children of
// a parent are given ids that are 8 apart: they run on the same XCD and share the parent's list and records in L2.
struct TileMap {
  int gx, gy, pgx, pgy, s;
  const uint32_t* order;        // list tiles by descending list length
  __device__ __forceinline__ bool locate(int b, int& tx, int& ty, int& parent, uint32_t& child_bit) const {
    const int nch = 1 << (2 * s);
    const int xcd = b & 7, j = b >> 3;
    const int slot = (j >> (2 * s)) * 8 + xcd, c = j & (nch - 1);
    if (slot >= pgx * pgy) return false;
    const int p = (int)order[slot];
    const int py = p / pgx, px = p - py * pgx;
    const int cy = c >> s, cx = c & ((1 << s) - 1);
    tx = (px << s) + cx; ty = (py << s) + cy;
    parent = p;
    child_bit = 1u << (GM_KEY_MASK_SHIFT + c);
    return tx < gx && ty < gy;
  }
  int blocks() const { return ((pgx * pgy + 7) / 8) * 8 << (2 * s); }
};

// Policy 2 (32-px parents): the key's mask has one bit per 8x8 QUADRANT of the parent, bit qy * 4 + qx; quadrant (wave & 1, wave >> 1)
// of tile (tx, ty) is quadrant (2 (tx & 1) + (wave & 1), 2 (ty & 1) + (wave >> 1)) of its parent.
__device__ __forceinline__ uint32_t quadrant_bit(int tx, int ty, int wave) {
  return 1u << (GM_KEY_MASK_SHIFT + (2 * (ty & 1) + (wave >> 1)) * 4 + 2 * (tx & 1) + (wave & 1));
}

// Dispatch order of the blend kernels (gm_tile_order.h) as a launch of its own: only when the tile pass did not produce it.
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint2* __restrict__ ranges, int tiles, uint32_t* __restrict__ order,
                                                          uint32_t* __restrict__ hint, uint32_t* __restrict__ epoch,
                                                          uint32_t* __restrict__ scratch) {
  __shared__ uint32_t cnt[256];
  __shared__ uint32_t wsum[16];
  tile_order_block<1024>(ranges, tiles, order, cnt, wsum, hint, epoch, scratch);
}

// Dispatch order of the BACKWARD blend.  What a quadrant's wave has to walk is known exactly after the forward: the list
// up to the deepest contributor of its pixels (n_contrib); list length says little (a 12 k-entry list that saturates after
// 200 entries is cheap, a 1.5 k-entry silhouette list walked to the end is not).  One workgroup per list tile takes the maximum
// of n_contrib over the tile's area, one workgroup sorts the tiles by it.
__global__ __launch_bounds__(256) void tile_work_kernel(const uint32_t* __restrict__ n_contrib, int W, int H, int pgx, int shift,
                                                        uint32_t* __restrict__ work) {
  __shared__ uint32_t part[4];
  const int p = blockIdx.x, py = p / pgx, px = p - py * pgx;
  const int side = GM_TILE << shift;                          // 16, 32 or 64 pixels
  const int x0 = px * side, y0 = py * side;
  uint32_t m = 0;
  for (int i = threadIdx.x; i < side * side; i += 256) {
    const int x = x0 + (i & (side - 1)), y = y0 + (i >> (4 + shift));
    if (x < W && y < H) m = max(m, n_contrib[(size_t)y * W + x]);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) work[p] = max(max(part[0], part[1]), max(part[2], part[3]));
}
__global__ __launch_bounds__(1024) void tile_order_work_kernel(const uint32_t* __restrict__ work, int tiles, uint32_t* __restrict__ order) {
  __shared__ uint32_t cnt[256];
  __shared__ uint32_t wsum[16];
  tile_order_by<1024>([&](int t) { return work[t]; }, tiles, order, cnt, wsum);
}

int launch_tile_order(ImageState& img, int tiles, uint32_t* work_hint, int debug, hipStream_t s) {
  StageScope sc(ST_RANGES, s);
  hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, img.ranges, tiles, img.tile_order, work_hint, img.epoch, img.tile_work);   // tile_work: scratch here, the backward fills it anew
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Front end shared by both blend kernels: from the parent tile's list to dense batches of candidate records.
//   stage A (key scan): every iteration the wave looks at up to RQ_K chunks of 64 consecutive (key, id) pairs (loaded the
//            iteration before, 512 contiguous bytes per chunk) and appends the entries whose key carries this tile's child
//            bit, in list order, to a wave-private ring of (id, list position) in LDS - one ballot + mbcnt per chunk;
//   stage B (gather): up to 64 candidates are popped (lane j <- candidate j) and their 36-byte splat records loaded into
//            registers; three register sets rotate (being issued / in flight / being consumed), so the batch issued in
//            iteration i is consumed in iteration i + 2 and the dependent chain list position -> id -> record never stalls
//            a wave that finds few entries of its own.
// Each iteration issues exactly RQ_K pair loads followed by 3 record loads (clamped addresses when there is nothing to
// fetch), so `s_waitcnt vmcnt(3)` at the top of an iteration means "everything except the gather issued last iteration
// has landed"; left to itself the compiler puts vmcnt(0) in the middle of the iteration.  A batch holds up to 64 REAL
// candidates instead of the ~16 of 64 list entries that concern a 16-px tile of a 32-px parent: the per-batch work
// (culling, staging) is paid a quarter as often and nothing is fetched for entries of the other children.
// (Measured and dropped: the same front end with the records landing in a three-slot LDS ring by LDS-DMA instead of
// registers - 10 KiB of LDS per wave, 3-4 workgroups per CU instead of 6: render 0.19-0.22 ms against 0.18 before.)
#define RQ_K 4                   // key chunks scanned per iteration
#define RQ_QA 128                // candidate ring entries per wave (power of two)

struct Gather {                  // lane j: record of candidate j
  float4 a;                      // x, y, conic.x, conic.y
  float4 b;                      // conic.z, opacity, r, g
  float c;                       // b
  uint32_t id, pos;              // Gaussian id, list position
};
// exponent of one staged entry at one pixel, e = power * log2(e), on the packed-f32 pipe (v_pk_add / v_pk_mul operate on a
// register PAIR in one issue slot): d = (x, y) - pix, q = (a', c') * d, q.x += b' d.y, e = q.x d.x + q.y d.y.
// Staged record: RA = (x, y, a', c'), RB = (b', opacity, r, g) with a' = -log2e/2 conic.x, b' = -log2e conic.y, c' = -log2e/2 conic.z.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float staged_exponent(const float4& RA, float bq, v2f pix, v2f& d) {
  const v2f xy = {RA.x, RA.y}, ac = {RA.z, RA.w};
  d = xy - pix;
  v2f q = ac * d;
  q.x = __builtin_fmaf(bq, d.y, q.x);
  const v2f r = q * d;
  return r.x + r.y;
}

__device__ __forceinline__ Gather issue_gather(const float4* __restrict__ splat, uint2 cand) {
  Gather g;
  g.a = splat_row(splat, (size_t)cand.x, 0); g.b = splat_row(splat, (size_t)cand.x, 1); g.c = splat_blue(splat, (size_t)cand.x);
  g.id = cand.x; g.pos = cand.y;
  return g;
}

// ---------------------------------------------------------------------------------------------
// Forward blend, round 3: the exponent of every (survivor, pixel) pair comes from the MATRIX core.
//
// With c = (cx, cy) the pixel's offset from the centre of its 8x4 half of the quadrant and (u, v) the splat centre's offset
// from the same point, the exponent e = log2(e) * power is a quadratic polynomial in c whose six coefficients depend on the
// survivor only:   e = a' cx^2 + b' cx cy + c' cy^2 + (-2a'u - b'v) cx + (-b'u - 2c'v) cy + (a'u^2 + b'uv + c'v^2).
// For 16 survivors and the 64 pixels of the wave that is D[32 x 32] = A[32 x 6] B[6 x 32]: row (half h, survivor s) of A holds
// survivor s's coefficients about the centre of half h, column j of B the six monomials of pixel j of a half - three chained
// v_mfma_f32_32x32x2_f32.  Row 8 (s / 4) + 4 h + s % 4 lands in register s of the lanes of half h (tools/mfma_probe/probe32.hip),
// i.e. lane = pixel ends up with the exponents of the 16 survivors in 16 registers, which is exactly what the per-pixel
// recurrence wants.  The coefficients are computed once per candidate, lane-parallel, when the batch is staged, and fetched
// as three 4-byte LDS reads per lane and group; what is still broadcast per survivor is (r, g, b, opacity).
// Per survivor the vector ALU keeps exp, the opacity product, the two alpha decisions and the T / C recurrence: ~13
// instructions instead of ~20, and the LDS return traffic drops from 40 to ~17 bytes per lane.  192 matrix cycles per 16
// survivors ride beside ~800 vector cycles.
// Arithmetic: the polynomial is evaluated about a point at most 3.5 / 1.5 pixels away from every pixel, so its terms are at most
// a few tens for the narrowest splat the 0.3-px^2 low-pass filter allows; the matrix core sums the six products to within 2 ulp
// of the largest (probe32: 1.2e-7 of the sum of magnitudes): |e - e_exact| <~ 1e-5, alpha to ~1e-5 relative (the pixel-relative
// form of round 2: ~5e-7).  Because e now carries an absolute error, the reference's "skip when power > 0" (a guard against
// rounding at the splat's own centre, where power = -0 +- 1e-7) becomes e := min(e, 0): skipping would drop a splat at its
// brightest pixel whenever the polynomial came out at +1e-6.  The backward kernel clamps likewise.
typedef float v16f __attribute__((ext_vector_type(16)));

// The A-operand table of the exponent polynomial (image-only frames; a training step's two halves take the SAME alpha >= 1/255 decision
// for every (entry, pixel) - backward.cu repeats forward.cu's expression for exactly that reason - by both using staged_exponent: EXACT
// below.  The other way round, the backward on this polynomial, was built in round 4 and measured 26 % slower: NOTEBOOK.md section 2,
// tools/experiments/render_bwd_matrix_exponent_round4.hip.txt).  ct[4 * 6 * 32]: [group of 16 survivors][monomial][MFMA row]; row of (half h,
// survivor sg of the group) = 8 (sg / 4) + 4 h + sg % 4.  e' = log2(e) power + log2(opacity): the opacity rides in the constant
// coefficient, so that 2^e' IS opacity * G and the per-survivor opacity product (and its broadcast) is gone from both kernels.
// Contraction is off and every product is spelled out: every build of the kernel must produce bit-identical coefficients, whatever the
// compiler would fuse in one of them.  (ucx, vcy): centre of half 0 of the wave's quadrant; half 1 lies four rows below.
#define GM_POLY_PAD -1.0e30f      // constant coefficient of a row that holds no survivor: 2^e' = 0, the entry is skipped by every pixel
__device__ __forceinline__ void stage_poly(float* __restrict__ ct, int slot, float x, float y, float conx, float cony, float conz, float opacity,
                                           float ucx, float vcy, float kshift = 0.0f) {
#pragma clang fp contract(off)
  const int g = slot >> 4, sg = slot & 15;
  const float a = (-0.5f * LOG2E) * conx, b = (-LOG2E) * cony, c = (-0.5f * LOG2E) * conz;
  const float u = x - ucx, v0 = y - vcy, v1 = v0 - 4.0f;
  // log2(opacity) (v_log_f32); a conic that is not positive (an overflowing or rounded-away determinant in the preprocess) has
  // power > 0 wherever the reference evaluates it and is skipped there (forward.cu:336): the row is padded out
  const float au = a * u, bu = b * u, lo = (conx > 0.f && conz > 0.f) ? __builtin_amdgcn_logf(opacity) + kshift : GM_POLY_PAD;
  float* row = &ct[192 * g + 8 * (sg >> 2) + (sg & 3)];                           // row of (half 0, survivor sg); half 1: + 4
  row[0] = a; row[4] = a; row[32] = b; row[36] = b; row[64] = c; row[68] = c;
  row[96] = -2.0f * au - b * v0;  row[100] = -2.0f * au - b * v1;
  row[128] = -bu - 2.0f * c * v0; row[132] = -bu - 2.0f * c * v1;
  row[160] = (u * (au + b * v0) + c * v0 * v0) + lo; row[164] = (u * (au + b * v1) + c * v1 * v1) + lo;
}
__device__ __forceinline__ void pad_poly(float* __restrict__ ct, int slot) {       // slot < 64
  float* row = &ct[192 * (slot >> 4) + 8 * ((slot & 15) >> 2) + (slot & 3)];
  row[160] = GM_POLY_PAD; row[164] = GM_POLY_PAD;
}
// the three chained steps; B0 / B1 / B2: the lane's monomials (cx^2 | cx cy), (cy^2 | cx), (cy | 1) for k = lane / 32
__device__ __forceinline__ v16f poly_exponents(const float* __restrict__ ct, int j, int lane, float B0, float B1, float B2) {
  const float* ctg = &ct[12 * j + lane];                                          // 192 (j / 16) + lane
  const float A0 = ctg[0], A1 = ctg[64], A2 = ctg[128];
  v16f E = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  E = __builtin_amdgcn_mfma_f32_32x32x2f32(A0, B0, E, 0, 0, 0);
  E = __builtin_amdgcn_mfma_f32_32x32x2f32(A1, B1, E, 0, 0, 0);
  E = __builtin_amdgcn_mfma_f32_32x32x2f32(A2, B2, E, 0, 0, 0);
  return E;
}
// The matrix-core forward folds min(0.99, .) into the exponent: 2^(e' - log2 0.99) with v_exp's clamp bit IS alpha / 0.99; the factor
// 0.99 rides in the staged colours and in T - 0.99 w' (10 instead of 11 vector instructions per survivor).  The EXACT build keeps `min`.
// Alternates that were compile-time switches until round 6 (three register sets, unfolded clamp, 256-thread workgroups) and their
// numbers: tools/experiments/render_blend_compile_time_alternates_round5.hip.txt.
#define GM_FWD_SUB 4              // survivors whose alpha evaluations interleave in the forward recurrence
struct RenderFrames {            // the frames of a batched forward blend (gridDim.z): buffer distances + where each frame's status words go
  int frames;
  FrameOfs go, bo, io, co;
  int* status[GM_BATCH_MAX];
};
struct FwdLds {                  // per wave: 5.1 KiB
  uint2 qa[RQ_QA];               // candidate ring: (Gaussian id, list position)
  float ct[4 * 6 * 32];          // [group of 16 survivors][monomial][MFMA row]: lane l of MFMA step m reads ct[192 g + 64 m + l]
  float4 sb[68];                 // (r, g, b, opacity) per survivor (+ 4 entries of padding with opacity 0 behind the last)
};

// STATE = false: image-only frame (GM_FWD_IMAGE_ONLY) - final_T / n_contrib, which only a backward pass reads, are neither
// tracked nor written.  TRACE: per-wave start / end / list length / iterations / candidates / survivors (tools/wave_trace.py).
// EXACT (GM_FWD_EXACT_EXPONENT: the forward of a training step; gm_debug_forward_exact_exponent forces it for every frame): the exponents of a group come from the
// pixel-relative form of round 2 / of the backward kernel (staged_exponent: |e - e_exact| ~ 5e-7) instead of the matrix core's
// polynomial (~1e-5).  Everything else - lists, cull, decisions, recurrence - is the same code, so the two builds may differ only
// where an entry's alpha or a pixel's T sits within the polynomial's error of a threshold (tests/test_gpu_parity.py).
template <bool STATE, bool TRACE, bool EXACT = false>
__global__ __launch_bounds__(64) void render_fwd_kernel(const uint2* __restrict__ ranges, const uint2* __restrict__ pairs,
                                                        const float4* __restrict__ splat, int W, int H, TileMap tm,
                                                        const float* __restrict__ bg, float* __restrict__ out_color,
                                                        float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                        unsigned long long* __restrict__ trace,
                                                        const uint32_t* __restrict__ counters, int* __restrict__ status_host,
                                                        uint32_t* __restrict__ hint, const uint32_t* __restrict__ epoch, const RenderFrames rf) {
  const unsigned long long t_start = TRACE ? wall_clock64() : 0ull;
  // frame blockIdx.z of a batch (gm_common.h FrameOfs; a single frame: zero distances, its status words through status_host)
  ranges = frame_ptr(ranges, rf.io); pairs = frame_ptr(pairs, rf.bo); splat = frame_ptr(splat, rf.go); tm.order = frame_ptr(tm.order, rf.io);
  out_color = frame_ptr(out_color, rf.co); counters = frame_ptr(counters, rf.go); epoch = frame_ptr(epoch, rf.io);
  if (STATE) { final_T = frame_ptr(final_T, rf.io); n_contrib = frame_ptr(n_contrib, rf.io); }
  if (rf.frames > 1) status_host = rf.status[blockIdx.z];
  int tr_iters = 0, tr_cand = 0;
  // One 8x8 pixel quadrant = one wave = one workgroup (placed and retired on its own); ids 8 apart share an XCD:
  // id = ((tile slot j) * 4 + quadrant) * 8 + xcd.
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x >> 3) & 3);
  const int tile_block = (int)(((blockIdx.x >> 5) << 3) | (blockIdx.x & 7));
  if (status_host && blockIdx.x == 0 && threadIdx.x < 4)                 // the frame's status words {num_rendered, -, policy, refused}
    __hip_atomic_store(status_host + threadIdx.x, (int)counters[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // straight into
  int tx, ty, parent;                                                    // the caller's page-locked words: no copy launch behind the frame
  uint32_t child_bit;
  if (!tm.locate(tile_block, tx, ty, parent, child_bit)) return;
  if (tm.s == 1) child_bit = quadrant_bit(tx, ty, wave);                 // policy 2: the keys carry one bit per 8x8 quadrant of the parent
  const uint2 range = ranges[parent];
  const int n = (int)(range.y - range.x);
  const uint2* list = pairs + range.x;           // (key, Gaussian id) per list entry

  const int px = tx * GM_TILE + (wave & 1) * 8 + (lane & 7);
  const int py = ty * GM_TILE + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  // T > 0: transmittance of a live pixel.  T < 0: the pixel has stopped (reference `done`) and |T| is its final transmittance - a
  // stopped pixel then takes nothing with no extra state: T (1 - alpha) < 0 < 1e-4 is the stop test itself.
  float T = inside ? 1.0f : -1.0f, Cb = 0.f;
  v2f Crg = {0.f, 0.f};
  uint32_t last = 0;
  int work = 0;                                                               // entries this wave evaluated (wave-uniform): the work hint
  if (n > 0) {
    const float rx0 = (float)(tx * GM_TILE + (wave & 1) * 8), ry0 = (float)(ty * GM_TILE + (wave >> 1) * 8);
    __shared__ FwdLds L;
    __shared__ float4 x_ra[EXACT ? 68 : 1];                                     // EXACT: (x, y, a', c') and b' per survivor
    __shared__ float x_bq[EXACT ? 68 : 1];
    __shared__ uint32_t x_sp[EXACT ? 68 : 1];                                   //        list position + 1 (its place in the colour record holds the opacity)
    const v2f pixf = {(float)px, (float)py};
    // B operand of the three MFMA steps: monomials (cx^2, cx cy) / (cy^2, cx) / (cy, 1) of this lane's pixel column; k = lane / 32
    const float ccx = (float)(lane & 7) - 3.5f, ccy = (float)((lane >> 3) & 3) - 1.5f;
    const bool khi = lane >= 32;
    const float B0 = khi ? ccx * ccy : ccx * ccx, B1 = khi ? ccx : ccy * ccy, B2 = khi ? 1.0f : ccy;
    const float ucx = rx0 + 3.5f, vcy = ry0 + 1.5f;                            // centre of half 0 (half 1: + 4 rows)
    // rows of a group that hold no survivor are multiplied all the same: they must be finite (their opacity is 0), so the table
    // starts out as zeros and afterwards only ever holds coefficients of real entries
#pragma unroll
    for (int i = 0; i < 12; i++) L.ct[64 * i + lane] = 0.f;
    const int nlast = n - 1;
    int kpos = 0;                                  // next list position to scan
    uint32_t qa_head = 0, qa_cnt = 0;              // candidate ring (wave-uniform)
    uint2 kv[RQ_K];
    auto scan = [&]() {                            // stage A: the chunks in kv, in order, while the ring has room
      bool go = true;
#pragma unroll
      for (int k = 0; k < RQ_K; k++) {
        go = go && kpos < n && qa_cnt + 64u <= (uint32_t)RQ_QA;
        if (go) {
          const int p = kpos + lane;
          const bool mine = p < n && (kv[k].x & child_bit) != 0u;
          const unsigned long long bal = __ballot(mine);
          if (mine) L.qa[(qa_head + qa_cnt + lanes_below(bal)) & (RQ_QA - 1)] = make_uint2(kv[k].y, (uint32_t)p);
          qa_cnt += (uint32_t)__popcll(bal);
          kpos += 64;
        }
      }
    };
    auto load_keys = [&]() {
#pragma unroll
      for (int k = 0; k < RQ_K; k++) kv[k] = list[min(kpos + k * 64 + lane, nlast)];
    };
    auto pop = [&](int& count) {                   // stage B: up to 64 candidates, lane j <- candidate j, record loads issued
      count = (int)min(qa_cnt, 64u);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint2 cand = lane < count ? L.qa[(qa_head + (uint32_t)lane) & (RQ_QA - 1)] : make_uint2(0u, 0u);
      qa_head += (uint32_t)count; qa_cnt -= (uint32_t)count;
      return issue_gather(splat, cand);
    };
    load_keys();
    scan();                                        // (waits for the first keys)
    load_keys();
    auto step = [&](Gather& cur, int& n0, Gather& nxt, int& n2) -> bool {      // register sets rotate by call site
      if (TRACE) { tr_iters++; tr_cand += n0; }
      const unsigned long long live = __ballot(T > 0.0f);
      if (live == 0ull) return false;
      if (n0 == 0 && qa_cnt == 0u && kpos >= n) return false;
      // cull against the bounding box of the pixels that are still live (lane = y * 8 + x; scalar bit arithmetic)
      uint32_t cols = (uint32_t)live | (uint32_t)(live >> 32);
      cols |= cols >> 16; cols |= cols >> 8; cols &= 0xFFu;
      const float cx0 = rx0 + (float)(__ffs((int)cols) - 1), cx1 = rx0 + (float)(31 - __clz((int)cols));
      const float cy0 = ry0 + (float)((__ffsll(live) - 1) >> 3), cy1 = ry0 + (float)((63 - __clzll((long long)live)) >> 3);
      __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0): the keys and the gather issued last iteration
      scan();
      load_keys();
      nxt = pop(n2);
      if (n0 > 0) {
        const bool keep = lane < n0 && may_touch(cur.a.x, cur.a.y, cur.a.z, cur.a.w, cur.b.x, cur.b.y, cx0, cx1, cy0, cy1);
        const unsigned long long kb = __ballot(keep);
        const int ns = __popcll(kb);
        work += ns;
        // stage the survivors, compacted (slot = rank among the survivors, list order): polynomial coefficients about the two
        // half centres into the MFMA's A layout, colour + opacity, list position
        if (keep) {
          const int slot = (int)lanes_below(kb);
          constexpr bool FOLD = !EXACT;
          if (!EXACT) stage_poly(L.ct, slot, cur.a.x, cur.a.y, cur.a.z, cur.a.w, cur.b.x, cur.b.y, ucx, vcy, FOLD ? 0.0144995696951f : 0.0f);   // -log2(0.99)
          // (r, g, b, w): w = the opacity in the EXACT build, otherwise (the opacity lives in the polynomial) the 1-based list position
          // the backward state wants - one broadcast read per survivor for colour AND n_contrib
          const float cs = FOLD ? 0.99f : 1.0f;
          L.sb[slot] = make_float4(cs * cur.b.z, cs * cur.b.w, cs * cur.c, EXACT ? cur.b.y : (STATE ? __uint_as_float(cur.pos + 1u) : 0.f));
          if (EXACT) {
            x_ra[slot] = make_float4(cur.a.x, cur.a.y, (-0.5f * LOG2E) * cur.a.z, (-0.5f * LOG2E) * cur.b.x);
            x_bq[slot] = (-LOG2E) * cur.a.w;
          }
          if (STATE && EXACT) x_sp[slot] = cur.pos + 1u;   // 1-based list position: n_contrib
        }
        // survivors are taken four at a time: the up to three slots behind the last must come out as alpha = 0
        if (!EXACT && lane < 4 && ns + lane < 64) pad_poly(L.ct, ns + lane);
        if (lane < 4) L.sb[ns + lane] = make_float4(0.f, 0.f, 0.f, 0.f);     // (their colours are multiplied by that 0: they must be finite)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // groups of 16 survivors.  (Measured and dropped: issuing the NEXT group's three MFMA steps before the current group's
        // exponents are consumed - two accumulator sets, 142 VGPRs, three waves per SIMD instead of four: 4430 vs 4700 frames/s.)
        auto exponents = [&](const int j) -> v16f {
          if (EXACT) {                                                    // slots behind the last survivor: opacity 0, any exponent will do
            v16f E;
#pragma unroll
            for (int t = 0; t < 16; t++) {
              const int sl = min(j + t, 67);
              v2f dd;
              E[t] = staged_exponent(x_ra[sl], x_bq[sl], pixf, dd);
            }
            return E;
          }
          return poly_exponents(L.ct, j, lane, B0, B1, B2);
        };
        auto blend16 = [&](const v16f& E, const int j) -> bool {          // false: every pixel of the wave has stopped
          constexpr int SUB = GM_FWD_SUB;                                  // survivors per sub-block: their alpha evaluations interleave
#pragma unroll
          for (int q = 0; q < 16 / SUB; q++) {
            if (q > 0 && j + SUB * q >= ns) break;
            float4 S[SUB];
#pragma unroll
            for (int t = 0; t < SUB; t++) S[t] = L.sb[j + SUB * q + t];
            uint32_t SP[SUB];
            if (STATE) {
#pragma unroll
              for (int t = 0; t < SUB; t++) SP[t] = EXACT ? x_sp[j + SUB * q + t] : __float_as_uint(S[t].w);
            }
            float al[SUB]; bool ok[SUB];
#pragma unroll
            for (int t = 0; t < SUB; t++) {
              // opacity * G = 2^e' in one instruction (the opacity is part of the polynomial); EXACT: min(2^e, 1) * opacity, the exponent
              // clamped at 0 by v_exp_f32's clamp bit (see above)
              if (!EXACT) {
                // al[t] = alpha / 0.99: 2^(e' - log2 0.99), clamped to 1 by v_exp's clamp bit (= min(0.99, .) of alpha), 0 where alpha < 1/255
                const float oGp = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(E[SUB * q + t]), 0.0f, 1.0f);
                ok[t] = oGp >= (1.0f / 255.0f) / 0.99f;
                al[t] = ok[t] ? oGp : 0.0f;
                continue;
              }
              const float oG = S[t].w * __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(E[SUB * q + t]), 0.0f, 1.0f);
              ok[t] = oG >= 1.0f / 255.0f;
              al[t] = ok[t] ? fminf(0.99f, oG) : 0.0f;                                        // skip alpha < 1/255; alpha = min(0.99, .)
            }
#pragma unroll
            for (int t = 0; t < SUB; t++) {          // in list order
              // weight alpha T; T (1 - alpha) as T - alpha T.  Folded form: wa = (alpha / 0.99) T weighs colours staged as 0.99 c
              const float wa = al[t] * T, tt = !EXACT ? __builtin_fmaf(-0.99f, wa, T) : T - wa;
              const bool stop = tt < 0.0001f;                                               // (tt == T >= 1e-4 when alpha == 0; tt < 0 once stopped)
              const float w = stop ? 0.0f : wa;
              T = stop ? -__builtin_fabsf(T) : tt;                                          // stop WITHOUT applying the entry
              const v2f rg = {S[t].x, S[t].y}, ww = {w, w};
              Crg = rg * ww + Crg; Cb += S[t].z * w;
              // accepted <=> alpha >= 1/255 and not stopping (a stopped pixel's T < 0 stops again): the two compares' masks combined on the
              // scalar unit select the list position - one vector instruction where `w > 0 ? .. : ..` took two
              if (STATE) last = (ok[t] && !stop) ? SP[t] : last;
            }
            if (!__any(T > 0.0f)) return false;
          }
          return true;
        };
        for (int j = 0; j < ns; j += 16) {
          const v16f E = exponents(j);
          if (!blend16(E, j)) break;
        }
      }
      return true;
    };
    // TWO register sets (round 4): the batch issued in iteration i is consumed in iteration i + 1 (an iteration is 0.3 - 2 us, an L2
    // hit 0.2 - 0.4 us) and every load issued before an iteration has landed at its top (vmcnt(0)).  Nine registers fewer than with a
    // third set in flight: the image-only kernel needs 95 VGPRs instead of 111 - FIVE waves per SIMD instead of four - and the
    // pipelined loop gains 3.2 % (4820 -> 4980 frames/s, A/B in one call, profiles/r04_ab_sets.txt; the training forward, 104 VGPRs,
    // stays at four waves and gains 3 % from the shorter iteration).  Round 2 chose three sets for a lone wave's latency; what the
    // loop is short of is resident waves.
    int n0, n1 = 0;
    Gather g0 = pop(n0), g1 = g0;
    for (;;) {
      if (!step(g0, n0, g1, n1)) break;
      if (!step(g1, n1, g0, n0)) break;
    }
  }
  if (hint && work > 0 && lane == 0)                                     // (gm_tile_order.h: the next frames' dispatch order)
    atomicMax(&hint[1 + parent], (epoch[0] << 20) | min((uint32_t)work, GM_HINT_WORK_MASK));
  if (inside) {
    const size_t HW = (size_t)H * W, pid = (size_t)W * py + px;
    T = __builtin_fabsf(T);
    if (STATE) { final_T[pid] = T; n_contrib[pid] = last; }
    out_color[pid] = Crg.x + T * bg[0];
    out_color[HW + pid] = Crg.y + T * bg[1];
    out_color[2 * HW + pid] = Cb + T * bg[2];
  }
  if (TRACE && lane == 0) {
    unsigned long long* t = trace + 8 * ((size_t)tile_block * 4 + wave);
    t[0] = t_start; t[1] = wall_clock64(); t[2] = (unsigned long long)n;
    t[3] = (unsigned long long)tr_iters | ((unsigned long long)tr_cand << 16) | ((unsigned long long)work << 40);
    t[4] = 0; t[5] = 0;
  }
}

static unsigned long long* g_render_trace = nullptr;      // debugging aid (tools/wave_trace.py), never set by the package
extern "C" void gm_debug_render_trace(void* buffer) { g_render_trace = reinterpret_cast<unsigned long long*>(buffer); }
static float* g_bwd_front_T = nullptr;                   // verification aid (tests only): see render_bwd_kernel
extern "C" void gm_debug_backward_front_T(void* plane) { g_bwd_front_T = reinterpret_cast<float*>(plane); }
static bool g_fwd_exact = false;                          // verification aid (tests only): the EXACT build of the forward blend for EVERY frame
extern "C" void gm_debug_forward_exact_exponent(int on) { g_fwd_exact = on != 0; }

int launch_render_fwd(const GeomState& g, const uint2* pairs, ImageState& img, int W, int H, int mode,
                      const float* background, float* out_color, int* status_host, bool image_only, uint32_t* work_hint, int debug,
                      hipStream_t s, bool exact_exponent, const BatchOfs* bt) {
  StageScope sc(ST_RENDER, s);
  const TileGrid tg(W, H, mode);
  const TileMap tm{tg.gx, tg.gy, tg.pgx, tg.pgy, tg.s, img.tile_order};
  RenderFrames rf{};
  rf.frames = 1;
  if (bt) {
    rf.frames = bt->frames; rf.go = bt->geom; rf.bo = bt->binning; rf.io = bt->image; rf.co = bt->color;
    for (int f = 0; f < GM_BATCH_MAX; f++) rf.status[f] = bt->status[f];
  }
  if (rf.frames > 1 && tg.ptiles <= 0) { set_error("batched blend: empty tile grid"); return 1; }
  if (tg.ptiles > 0) {
    const dim3 grid(tm.blocks() * 4, 1, (uint32_t)rf.frames), block(64);     // one wave (8x8 quadrant) per workgroup
    if (g_fwd_exact || exact_exponent)
      hipLaunchKernelGGL((render_fwd_kernel<true, false, true>), grid, block, 0, s, img.ranges, pairs, g.splat, W, H, tm,
                         background, out_color, img.final_T, img.n_contrib, nullptr, g.counters, status_host, work_hint, img.epoch, rf);
    else if (g_render_trace)
      hipLaunchKernelGGL((render_fwd_kernel<true, true>), grid, block, 0, s, img.ranges, pairs, g.splat, W, H, tm,
                         background, out_color, img.final_T, img.n_contrib, g_render_trace, g.counters, status_host, work_hint, img.epoch, rf);
    else if (image_only)
      hipLaunchKernelGGL((render_fwd_kernel<false, false>), grid, block, 0, s, img.ranges, pairs, g.splat, W, H, tm,
                         background, out_color, img.final_T, img.n_contrib, nullptr, g.counters, status_host, work_hint, img.epoch, rf);
    else
      hipLaunchKernelGGL((render_fwd_kernel<true, false>), grid, block, 0, s, img.ranges, pairs, g.splat, W, H, tm,
                         background, out_color, img.final_T, img.n_contrib, nullptr, g.counters, status_host, work_hint, img.epoch, rf);
  } else if (status_host) {
    GM_HIP(hipMemcpyAsync(status_host, g.counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  }
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Backward blend.
#define GM_ACC_STRIDE 12   // floats per Gaussian in grad_acc: dcolor rgb (0-2), moments of h: 1, dx, dy, dx^2, dx dy, dy^2 (3-8)


// Backward blend, two phases per wave (round 3).  The exponent of an entry is evaluated per pixel in the pixel-relative form
// (staged_exponent, |e - e_exact| ~ 5e-7) and the opacity multiplied in afterwards; the forward's come from the matrix core (~1e-5), so
// the two halves can disagree on an entry whose alpha lies within 1e-5 of 1/255 (weight <= 0.4 %).  Round 4 built
// the backward on the matrix core with the forward's polynomial - identical decisions, every gradient test green - and it was 26 % slower:
// tools/experiments/render_bwd_matrix_exponent_round4.hip.txt.
//
// Phase 1, lane = pixel: the walk of backward.cu:441-556 with everything that is not the per-pixel recurrence taken out.
// Going back to front, with T_i the transmittance in front of entry i, w_i = alpha_i T_i and cd_i = c_i . dL/dpixel,
//   dL/dalpha_i = T_i cd_i - (A_i + T_final bg . dL/dpixel) / (1 - alpha_i),   A_{i-1} = A_i + w_i cd_i
// (the reference carries the three-channel `accum_rec` = A / T and the last colour / alpha for the same quantity: one
// scalar recurrence instead of three vector ones).  An entry a lane does not take gets alpha = 0, which makes every update
// the identity - no execution-mask juggling.  All an entry leaves behind per pixel are the two factors every gradient sum is
// a multiple of: w (dL/dcolour) and h = G dL/dG.  They go to a row of an LDS matrix M[slot][pixel]; the entry's centre and
// id are kept per slot.  Entries no pixel of the wave takes use no slot.
// Phase 2, lane = (pixel row r, slot es), once per SEVEN used slots: each lane folds the eight pixels of its row into the nine
// sums of its slot's Gaussian - dL/dcolour rgb and the six moments (1, dx, dy, dx^2, dx dy, dy^2) of h, dy being constant
// along a row - six vector instructions per pixel, no cross-lane traffic; the eight row partials of a slot meet through LDS
// (written [row][slot * 9 + value], read back by lane = slot * 9 + value: both conflict-free), and ONE atomic instruction
// with 63 active lanes commits seven 36-byte records (one L2 transaction per record, as before).
// Per entry: ~28 + ~9 vector instructions instead of ~44 + 28 for the per-entry cross-lane reduction of round 2
// (v_permlane32/16_swap + DPP butterfly), which is gone.
struct StagedB {                 // one survivor of the staged batch (one LDS address per entry in the walk)
  float4 a;                      // x, y, conic.x', conic.z'   (conic pre-multiplied for the exp2 argument)
  float4 b;                      // conic.y', opacity, r, g
  float4 c;                      // b, list position (bits), Gaussian id (bits), -
};
struct SlotB { float2 xy; uint32_t id, pad; };   // splat centre and id of a phase-2 slot
struct BwdLds {                  // per wave: 7.7 KiB
  uint2 qa[RQ_QA];               // candidate ring of the front end: (Gaussian id, list position)
  StagedB st[64];                // staged batch
  union {
    float2 M[7][65];             // (w, h) per slot and pixel; row stride 65 keeps phase 2's row reads conflict-free
    float part[8][72];           // phase 2: row partials [row][slot * 9 + value] (columns 63.. belong to the idle lanes)
    struct { float2 rg[64]; float bl[64]; } dpt;   // kernel start only: dL/dpixel of every pixel
  };
  SlotB slot[8];
};

__global__ __launch_bounds__(64) void render_bwd_kernel(const uint2* __restrict__ ranges,
                                                               const uint2* __restrict__ pairs,
                                                               const float4* __restrict__ splat, int W, int H, TileMap tm,
                                                               const float* __restrict__ bg, const float* __restrict__ final_T,
                                                               const uint32_t* __restrict__ n_contrib,
                                                               const float* __restrict__ dL_dpix, float* __restrict__ grad_acc,
                                                               const uint32_t* __restrict__ counters, int mode, float* __restrict__ front_T) {
  const int lane = threadIdx.x & 63;
  const int wave = (int)((blockIdx.x >> 3) & 3);
  const int tile_block = (int)(((blockIdx.x >> 5) << 3) | (blockIdx.x & 7));
  int tx, ty, parent;
  uint32_t child_bit;
  if ((int)counters[GM_CNT_POLICY] != mode || counters[GM_CNT_REFUSED] != 0u) return;   // lists were built under another emission policy: contribute nothing
  if (!tm.locate(tile_block, tx, ty, parent, child_bit)) return;
  if (tm.s == 1) child_bit = quadrant_bit(tx, ty, wave);
  const uint2 range = ranges[parent];
  const int n = (int)(range.y - range.x);
  if (n == 0) return;
  const uint2* list = pairs + range.x;           // (key, Gaussian id) per list entry
  const size_t HW = (size_t)H * W;

  const int px = tx * GM_TILE + (wave & 1) * 8 + (lane & 7);
  const int py = ty * GM_TILE + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const size_t pid = inside ? (size_t)W * py + px : 0;
  const float T_final = inside ? final_T[pid] : 0.f;
  float T = T_final;
  const v2f pix = {(float)px, (float)py};
  const int last = inside ? (int)n_contrib[pid] : 0;
  const float dpr = inside ? dL_dpix[pid] : 0.f, dpg = inside ? dL_dpix[HW + pid] : 0.f, dpb = inside ? dL_dpix[2 * HW + pid] : 0.f;
  float A = T_final * (bg[0] * dpr + bg[1] * dpg + bg[2] * dpb);      // A_i + T_final bg . dL/dpixel (see above)
  // entries at list positions >= max over the wave of n_contrib are never used: start there
  int max_last = last;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) max_last = max(max_last, __shfl_xor(max_last, d));
  const int start = __builtin_amdgcn_readfirstlane(max_last);   // number of list entries this wave has to visit (positions start-1 .. 0);
                                                                // readfirstlane: the compiler cannot see that the butterfly left a uniform value,
                                                                // and everything the walk's loops carry would otherwise live in vector registers
  if (start == 0) return;

  const float rx0 = (float)(tx * GM_TILE + (wave & 1) * 8), ry0 = (float)(ty * GM_TILE + (wave >> 1) * 8);
  __shared__ BwdLds B;
  BwdLds& L = B;
  // phase 2 geometry of this lane: slot es (7: idle), pixel row r; dL/dpixel of the row's eight pixels stays in registers
  const int es = lane & 7, r = lane >> 3, esc = min(es, 6);
  float2 dq[8];
  float dqb[8];
  B.dpt.rg[lane] = make_float2(dpr, dpg); B.dpt.bl[lane] = dpb;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 8; i++) {
    dq[i] = B.dpt.rg[r * 8 + i];
    dqb[i] = B.dpt.bl[r * 8 + i];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float rowy = ry0 + (float)r;
  const int cl = min(lane, 62), ce = cl / 9, ck = cl - 9 * ce;      // commit role of this lane: value ck of slot ce
  int m = 0;                                                          // slots in use (wave-uniform)
  float2* mrow = &B.M[0][lane];
  SlotB* mslot = &B.slot[0];

  auto phase2 = [&](const int cnt) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float2 c = B.slot[esc].xy;
    const float x0 = c.x - rx0, dy = c.y - rowy;
    v2f s01 = {0.f, 0.f};
    float s2 = 0.f, s3 = 0.f, s4 = 0.f, s6 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const float2 v = B.M[esc][r * 8 + i];                       // (w, h) of pixel i of this lane's row
      const float dx = x0 - (float)i, hx = v.y * dx;
      const v2f ww = {v.x, v.x}, drg = {dq[i].x, dq[i].y};
      s3 += v.y; s4 += hx;
      s6 = __builtin_fmaf(hx, dx, s6);
      s01 = ww * drg + s01;
      s2 = __builtin_fmaf(v.x, dqb[i], s2);
    }
    const v2f s34 = {s3, s4};
    const float s5 = dy * s34.x, s7 = dy * s34.y, s8 = dy * s5;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // every lane has read M before `part` (same storage) is written
    __builtin_amdgcn_wave_barrier();
    float* prow = &B.part[r][es * 9];
    prow[0] = s01.x; prow[1] = s01.y; prow[2] = s2; prow[3] = s34.x; prow[4] = s34.y; prow[5] = s5; prow[6] = s6; prow[7] = s7; prow[8] = s8;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 8; q++) tot += B.part[q][cl];
    const uint32_t gid = B.slot[ce].id;
    if (lane < 63 && ce < cnt) atomicAdd(grad_acc + (size_t)gid * GM_ACC_STRIDE + ck, tot);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // `part` has been read before phase 1 writes M again
    __builtin_amdgcn_wave_barrier();
  };

  // Same front end as render_fwd_kernel, walking the list back to front: chunk lane j <-> position start-1-kpos-j
  // (positions below 0 re-read entry 0 and are not "mine"); candidates enter the ring in descending list position.
  int kpos = 0;                                  // entries scanned so far (from the back)
  uint32_t qa_head = 0, qa_cnt = 0;
  uint2 kv[RQ_K];
  auto scan = [&]() {
    bool go = true;
#pragma unroll
    for (int k = 0; k < RQ_K; k++) {
      go = go && kpos < start && qa_cnt + 64u <= (uint32_t)RQ_QA;
      if (go) {
        const int p = start - 1 - kpos - lane;
        const bool mine = p >= 0 && (kv[k].x & child_bit) != 0u;
        const unsigned long long bal = __ballot(mine);
        if (mine) L.qa[(qa_head + qa_cnt + lanes_below(bal)) & (RQ_QA - 1)] = make_uint2(kv[k].y, (uint32_t)p);
        qa_cnt += (uint32_t)__popcll(bal);
        kpos += 64;
      }
    }
  };
  auto load_keys = [&]() {
#pragma unroll
    for (int k = 0; k < RQ_K; k++) kv[k] = list[max(start - 1 - kpos - k * 64 - lane, 0)];
  };
  auto pop = [&](int& count) {
    count = (int)min(qa_cnt, 64u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint2 cand = lane < count ? L.qa[(qa_head + (uint32_t)lane) & (RQ_QA - 1)] : make_uint2(0u, 0u);
    qa_head += (uint32_t)count; qa_cnt -= (uint32_t)count;
    return issue_gather(splat, cand);
  };
  load_keys();
  scan();
  load_keys();
  auto step = [&](Gather& cur, int& n0, const int n1, Gather& nxt, int& n2) -> bool {      // see render_fwd_kernel
    if (n0 == 0 && n1 == 0 && qa_cnt == 0u && kpos >= start) return false;
    __builtin_amdgcn_s_waitcnt(0x0F73);                                  // vmcnt(3): all but the gather issued last iteration (three register sets)
    scan();
    load_keys();
    nxt = pop(n2);
    if (n0 > 0) {
      // A pixel takes part in this batch only if its last contributor lies above the batch's lowest position: cull against
      // the bounding box of those pixels (at the deep end of the walk only the few pixels that reached far into the list
      // are still in play).  Lane = y * 8 + x.
      const int pos_lo = __builtin_amdgcn_readlane((int)cur.pos, n0 - 1);
      const unsigned long long live = __ballot(last > pos_lo);
      float cx0 = rx0, cx1 = rx0 + 7.0f, cy0 = ry0, cy1 = ry0 + 7.0f;
      if (live != 0ull) {
        uint32_t cols = (uint32_t)live | (uint32_t)(live >> 32);
        cols |= cols >> 16; cols |= cols >> 8; cols &= 0xFFu;
        cx0 = rx0 + (float)(__ffs((int)cols) - 1);
        cx1 = rx0 + (float)(31 - __clz((int)cols));
        cy0 = ry0 + (float)((__ffsll(live) - 1) >> 3);
        cy1 = ry0 + (float)((63 - __clzll((long long)live)) >> 3);
      }
      const bool keep = live != 0ull && lane < n0 && may_touch(cur.a.x, cur.a.y, cur.a.z, cur.a.w, cur.b.x, cur.b.y, cx0, cx1, cy0, cy1);
      // compacted, conic pre-multiplied for the exp2 argument, as in render_fwd_kernel
      const unsigned long long kb = __ballot(keep);
      const int ns = __popcll(kb);
      if (keep) {
        const int sl = (int)lanes_below(kb);
        StagedB& o = B.st[sl];
        o.a = make_float4(cur.a.x, cur.a.y, (-0.5f * LOG2E) * cur.a.z, (-0.5f * LOG2E) * cur.b.x);
        o.b = make_float4((-LOG2E) * cur.a.w, cur.b.y, cur.b.z, cur.b.w);
        o.c = make_float4(cur.c, __uint_as_float(cur.pos), __uint_as_float(cur.id), 0.f);   // 0-based list position == reference `contributor`
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // One staged entry of the walk (records in registers).  Returns nothing; an entry no lane takes costs the exponent and the vote.
      auto entry = [&](const float4& RA, const float4& RB, const float4& RC) {
        v2f dd;
        const float e = staged_exponent(RA, RB.x, pix, dd);             // power * log2(e), pixel-relative form (the forward kernel's polynomial
                                                                         // agrees to ~1e-5; its decisions can differ on an entry in a few 10^5)
        const float G = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(e), 0.0f, 1.0f);   // exponent clamped at 0, as in the forward kernel
        const float oG = RB.y * G;                                       // opacity * G (the unclamped alpha)
        const int pos = (int)__float_as_uint(RC.y);
        const bool valid = (pos < last) && (oG >= 1.0f / 255.0f);        // (alpha = min(0.99, oG) >= 1/255  <=>  oG >= 1/255)
        if (!__any(valid)) return;
        const float oGe = valid ? oG : 0.0f;                             // a lane that skips the entry: alpha 0, every update the identity
        const float al = __builtin_amdgcn_fmed3f(oGe, 0.0f, 0.99f);      // alpha = min(0.99, opacity G)
        const float inv = __builtin_amdgcn_rcpf(1.f - al);               // 1 / (1 - alpha)
        const float cd = __builtin_fmaf(RC.x, dpb, __builtin_fmaf(RB.w, dpg, RB.z * dpr));
        T = T * inv;                                                     // transmittance in front of the entry
        const float wv = al * T;
        const float dL_dalpha = T * cd - A * inv;
        A = __builtin_fmaf(wv, cd, A);
        *mrow = make_float2(wv, oGe * dL_dalpha);                        // M[m][lane]: w ; h = G dL/dG with dL/dG = opacity dL/dalpha
        mslot->xy = make_float2(RA.x, RA.y);                             // slot[m] (uniform address, uniform value)
        mslot->id = __float_as_uint(RC.z);
        mrow += 65; mslot += 1;                                          // the two LDS addresses advance as vector registers of their own:
        m += 1;                                                          // the slot count itself then lives in a scalar register
        if (m == 7) { phase2(7); m = 0; mrow = &B.M[0][lane]; mslot = &B.slot[0]; }
      };
      // Software pipeline of the staged records, round 5: TWO register sets by call site (the loop body is the entry twice).  Entry
      // j + 2's three LDS reads are issued when entry j is done with its set and land while entry j + 1 is walked.  With ONE rotating set
      // (round 4) the compiler copies four values per entry out of the registers the prefetch is about to
      // overwrite (opacity, b', list position, the next LDS address): 4 of the ~28 vector instructions of an entry in a kernel that keeps
      // its SIMDs' vector ALUs 78 % busy (profiles/r05_blend_sq_pmc.txt).
      // (ns == 0 - nothing survived the cull - reads slot 0 and walks nothing)
      const int j1 = max(min(1, ns - 1), 0);
      float4 RA0 = B.st[0].a, RB0 = B.st[0].b, RC0 = B.st[0].c;
      float4 RA1 = B.st[j1].a, RB1 = B.st[j1].b, RC1 = B.st[j1].c;
      for (int j = 0; j < ns; j += 2) {
        entry(RA0, RB0, RC0);
        { const int jn = min(j + 2, ns - 1); RA0 = B.st[jn].a; RB0 = B.st[jn].b; RC0 = B.st[jn].c; }
        if (j + 1 >= ns) break;
        entry(RA1, RB1, RC1);
        { const int jn = min(j + 3, ns - 1); RA1 = B.st[jn].a; RB1 = B.st[jn].b; RC1 = B.st[jn].c; }
      }
    }
    return true;
  };
  int n0, n1, n2 = 0;
  Gather g0 = pop(n0), g1 = pop(n1), g2 = g1;
  for (;;) {
    if (!step(g0, n0, n1, g2, n2)) break;
    if (!step(g1, n1, n2, g0, n0)) break;
    if (!step(g2, n2, n0, g1, n1)) break;
  }
  if (m > 0) phase2(m);
  // verification aid (gm_debug_backward_front_T; null on the product path): the transmittance the walk arrives at in FRONT of a pixel's
  // first entry.  It is final_T divided by (1 - alpha) of every entry the backward took for the pixel: 1 up to rounding when those
  // are the entries the forward blended, off by a factor (1 - alpha) >= 0.4 % for every entry the two halves disagree about.
  if (front_T && inside) front_T[pid] = T;
}


int launch_render_bwd(const GeomState& g, const uint2* pairs, ImageState& img, int W, int H, int mode,
                      const float* background, const float* dL_dpix, int debug, hipStream_t s) {
  StageScope sc(ST_RENDER_BWD, s);
  const TileGrid tg(W, H, mode);
  const TileMap tm{tg.gx, tg.gy, tg.pgx, tg.pgy, tg.s, img.tile_order_bwd};
  if (tg.ptiles > 0) {
    hipLaunchKernelGGL(tile_work_kernel, dim3(tg.ptiles), dim3(256), 0, s, img.n_contrib, W, H, tg.pgx, tg.s, img.tile_work);
    hipLaunchKernelGGL(tile_order_work_kernel, dim3(1), dim3(1024), 0, s, img.tile_work, tg.ptiles, img.tile_order_bwd);
  }
  if (tg.ptiles > 0)
    hipLaunchKernelGGL(render_bwd_kernel, dim3(tm.blocks() * 4), dim3(64), 0, s, img.ranges, pairs, g.splat, W, H, tm,
                       background, img.final_T, img.n_contrib, dL_dpix, g.grad_acc, g.counters, mode, g_bwd_front_T);
  GM_LAUNCH_CHECK(debug, s);
  return 0;
}

}  // namespace gm
