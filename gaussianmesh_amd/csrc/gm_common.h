// gm_common.h -- internal declarations shared by the translation units of libgmesh_hip.so.
// Target: gfx950 (MI355X, CDNA4), wave64.  No CUDA compatibility layer, no dual paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <type_traits>

#define GM_TILE 16              // tile edge (pixels); reference cuda_rasterizer/config.h:15-16
#define GM_WAVE 64

namespace gm {

// ---------------------------------------------------------------------------------------------
// error plumbing
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define GM_HIP(call)                                                            \
  do {                                                                          \
    hipError_t _e = (call);                                                     \
    if (_e != hipSuccess) return gm::hip_fail(_e, #call, __FILE__, __LINE__);   \
  } while (0)

// after a kernel launch: always check the launch error; in debug mode also synchronise
// (reference CHECK_CUDA, cuda_rasterizer/auxiliary.h:165-172)
#define GM_LAUNCH_CHECK(debug, stream)                                          \
  do {                                                                          \
    GM_HIP(hipGetLastError());                                                  \
    if (debug) GM_HIP(hipStreamSynchronize(stream));                            \
  } while (0)

// ---------------------------------------------------------------------------------------------
// per-stage timing (gm_profile_*)
enum Stage {
  ST_PREPROCESS = 0, ST_DEPTH_SORT, ST_SCAN, ST_DUPLICATE, ST_TILE_SORT, ST_RANGES, ST_RENDER,
  ST_RENDER_BWD, ST_PREPROCESS_BWD, ST_DEFORM, ST_SH_COLORS, ST_LOSS, ST_LOSS_BWD, ST_MESH_RS, ST_COUNT
};
struct StageScope {            // records start/stop events on `s` if profiling is enabled
  StageScope(Stage st, hipStream_t s);
  ~StageScope();
  Stage st; hipStream_t s; void* rec;
};

// ---------------------------------------------------------------------------------------------
// scratch layouts.  Bump allocation inside caller-owned byte buffers, 256-byte aligned absolute
// addresses (reference: obtain<T>() with 128-byte alignment, rasterizer_impl.h:21-27).
template <typename T>
static inline T* carve(char*& p, size_t count) {
  uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 255) & ~uintptr_t(255);
  T* out = reinterpret_cast<T*>(a);
  p = reinterpret_cast<char*>(out + count);
  return out;
}

#define GM_SORT_ITEMS 4096      // keys per workgroup per radix pass (256 threads x 16)
#define GM_SCAN_ITEMS 256       // sorted positions per entry of GeomState::chunk_inst; an emission workgroup takes one or two such runs

#define GM_SORT_SMALL_N (3u << 19)   // up to this many keys (1.5 M) the radix sort uses 1024-key tiles (more workgroups),
#define GM_SORT_MID_N (4u << 20)     // up to this many 2048-key tiles, 4096-key tiles above
// keys per workgroup / number of histogram columns (workgroups) the radix sort uses for n keys
static inline size_t sort_tile_keys(size_t n) { return n <= GM_SORT_SMALL_N ? 1024 : (n <= GM_SORT_MID_N ? 2048 : GM_SORT_ITEMS); }
static inline size_t sort_blocks(size_t n) {
  const size_t tile = sort_tile_keys(n);
  return (n + tile - 1) / tile;
}
#define GM_SORT_CHUNK 32             // histogram rows (workgroups) per scan workgroup
// counters of the per-chunk digit totals: [chunks][256]
static inline size_t sort_chunk_counters(size_t n) { return 256 * ((sort_blocks(n) + GM_SORT_CHUNK - 1) / GM_SORT_CHUNK); }

// ordering of a forward (gm_bucket.hip)
#define GM_BUCKET_BITS 11            // MSD partition of the depth keys into <= 2048 buckets; also the digit of the one-pass tile sort
// keys per thread of the depth partition (DP), of the tile pass (TP) and of the in-LDS bucket sort (BS): 16 each.  Compile-time
// knobs because register use follows them (8: 68-77 VGPRs instead of 117-133) and with it where a workgroup can be placed while
// other frames' blend kernels fill the chip (measured, 16 stays)
#ifndef GM_DP_ROUNDS
#define GM_DP_ROUNDS 16
#endif
#ifndef GM_TP_ROUNDS
#define GM_TP_ROUNDS 16
#endif
#ifndef GM_BS_ROUNDS
#define GM_BS_ROUNDS 16
#endif
#define GM_BK_ROUNDS_MIN (GM_DP_ROUNDS < GM_TP_ROUNDS ? GM_DP_ROUNDS : GM_TP_ROUNDS)
#define GM_BK_TILE (4 * GM_BK_ROUNDS_MIN * 64)   // fewest keys a workgroup of a partition / tile-sort pass takes (sizes the histogram rows)
#define GM_BK_CHUNK 32               // histogram rows per scan workgroup
static inline size_t bk_blocks(size_t n) { return (n + GM_BK_TILE - 1) / GM_BK_TILE; }
static inline size_t bk_chunks(size_t n) { return (bk_blocks(n) + GM_BK_CHUNK - 1) / GM_BK_CHUNK; }
#define GM_ACC_SLOTS 512             // accumulators of a pass: [8 digit groups][64] group-sum slots, then chunk_total [chunks][2048]
static inline size_t bk_acc_words(size_t n) { return GM_ACC_SLOTS + (bk_chunks(n) << GM_BUCKET_BITS); }
// Depth buckets that follow the data: a coarse histogram of the visible depth keys (float bits >> 20: eight bins per octave
// of z, 2048 bins cover every positive float) is accumulated by the preprocess kernel; each coarse bin then gets a share of
// the <= 2048 buckets in proportion to its count and is subdivided linearly (gm_bucket.hip, DepthMap).  A background far
// behind (or a floater right in front of) a dense object no longer squeezes the object into a few hundred buckets.
#define GM_COARSE_SHIFT 20
#define GM_COARSE_BINS 2048
#define GM_COARSE_COPIES 8           // copies of the coarse histogram the atomics are spread over (copy = workgroup & 7)
// word of (coarse bin, copy): neighbouring bins - the ones a launch hammers - lie 512 bytes apart and the copies of a bin too, so
// that their atomics do not queue up behind each other on one cache line (bins 16 apart share a line)
__host__ __device__ __forceinline__ uint32_t coarse_index(uint32_t bin, uint32_t copy) {
  return ((bin & 15u) * GM_COARSE_COPIES + copy) * (GM_COARSE_BINS / 16u) + (bin >> 4);
}
#define GM_BUCKET_BUDGET 1792        // buckets handed out in proportion to the bins' weights; every non-empty coarse bin gets at least one
// device scalars (GeomState::counters)
#define GM_CNT_RENDERED 0            // num_rendered (instance total of this forward)
#define GM_CNT_PREFILTER 1        // set when a Gaussian was frustum-culled although the caller declared the cloud prefiltered
#define GM_CNT_POLICY 2              // emission policy the counts were made under
#define GM_CNT_REFUSED 3             // emission refused (policy mismatch / capacity overflow): every list stays empty
#define GM_CNT_VISIBLE 4             // Gaussians with a depth key (written by the depth partition's histogram launch)
#define GM_CNT_NBUCKETS 5            // depth buckets in use
#define GM_CNT_CMIN 6                // first / last coarse bin that holds a visible key
#define GM_CNT_CMAX 7
#define GM_CNT_DIRECT_FAIL 8         // direct depth placement (DepthSlab below) could not order this frame: emission is refused with status 2
#define GM_CNT_COUNT 32
#define GM_SLOTS 256                 // atomic slots {instance sum, visible count, 2047 - first coarse bin, last coarse bin} filled by the preprocess kernel
#define GM_SLOT_STRIDE 32            // words between slots: one 128-byte line each (atomics on one line serialise in the memory-side atomic unit)

// Emission record of a Gaussian, 16 bytes: tile rectangle origin and size in 12 bits each (grids up to 4095 x 4095 tiles), the
// instance count in the four spare nibbles (0xFFFF = "65535 or more: see tiles_touched"), 64-bit emit mask.
#define GM_BIN_COUNT_SAT 0xFFFFu
__host__ __device__ __forceinline__ uint4 bin_pack(uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, uint32_t count, unsigned long long mask) {
  const uint32_t c = count < GM_BIN_COUNT_SAT ? count : GM_BIN_COUNT_SAT;
  return make_uint4(x0 | ((c & 0xFu) << 12) | (y0 << 16) | (((c >> 4) & 0xFu) << 28),
                    w | (((c >> 8) & 0xFu) << 12) | (h << 16) | ((c >> 12) << 28), (uint32_t)mask, (uint32_t)(mask >> 32));
}
__host__ __device__ __forceinline__ uint32_t bin_x0(const uint4& b) { return b.x & 0xFFFu; }
__host__ __device__ __forceinline__ uint32_t bin_y0(const uint4& b) { return (b.x >> 16) & 0xFFFu; }
__host__ __device__ __forceinline__ uint32_t bin_w(const uint4& b) { return b.y & 0xFFFu; }
__host__ __device__ __forceinline__ uint32_t bin_h(const uint4& b) { return (b.y >> 16) & 0xFFFu; }
__host__ __device__ __forceinline__ uint32_t bin_count(const uint4& b) {       // GM_BIN_COUNT_SAT: look tiles_touched up
  return ((b.x >> 12) & 0xFu) | ((b.x >> 28) << 4) | (((b.y >> 12) & 0xFu) << 8) | ((b.y >> 28) << 12);
}

// The blend kernels' record of a Gaussian: x, y, conic.x, conic.y | conic.z, opacity, r, g | b - nine floats, 36 bytes, records back
// to back (GM_SPLAT_STRIDE 9: what the preprocess kernels write and the blends gather is what is used; 12 = the 48-byte record of
// rounds 1-4 with the depth and two pads behind b, 16-byte aligned rows).  Rows are accessed as 16-byte vectors at 4-byte alignment.
#ifndef GM_SPLAT_STRIDE
#define GM_SPLAT_STRIDE 9
#endif
typedef float gm_f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float4 splat_row(const float4* splat, size_t id, int row) {          // row 0 or 1
  const gm_f4u v = *reinterpret_cast<const gm_f4u*>(reinterpret_cast<const float*>(splat) + GM_SPLAT_STRIDE * id + 4 * row);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float splat_blue(const float4* splat, size_t id) { return reinterpret_cast<const float*>(splat)[GM_SPLAT_STRIDE * id + 8]; }
__device__ __forceinline__ void splat_store(float4* splat, size_t id, float x, float y, float conx, float cony, float conz, float opacity,
                                            float r, float g, float b, float depth) {
  float* p = reinterpret_cast<float*>(splat) + GM_SPLAT_STRIDE * id;
  *reinterpret_cast<gm_f4u*>(p) = gm_f4u{x, y, conx, cony};
  *reinterpret_cast<gm_f4u*>(p + 4) = gm_f4u{conz, opacity, r, g};
  if (GM_SPLAT_STRIDE >= 12) *reinterpret_cast<gm_f4u*>(p + 8) = gm_f4u{b, depth, 0.f, 0.f};
  else p[8] = b;
}

struct GeomState {              // per-Gaussian state (P-sized)
  float4* splat;                // [P][GM_SPLAT_STRIDE floats]: splat_row / splat_blue / splat_store above
  int* radii;                   // internal radii when the caller passes none
  uint32_t* tiles_touched;      // [P]
  uint4* bin;                   // [P] emission record (bin_pack below): candidate tile rectangle, instance count and, for rectangles
                                //     of <= 64 tiles, the bit mask (row-major) of the tiles actually emitted
  float* cov3D;                 // [P][6] (computed from scale/rot)
  uint8_t* clamped;             // [P] bit ch = SH colour channel ch was clamped at 0
  uint32_t* depth_key;          // [P] float bits of view z per Gaussian (0xFFFFFFFF = culled)
  uint2* dpairs[2];             // [P] (depth key, id): [1] partitioned into buckets, [0] scratch of an overfull bucket's sort
  uint32_t* order;              // [P] ids of the VISIBLE Gaussians in (depth, id) order
  uint4* bin_sorted;            // [P] the emission records in that order (duplicate_kernel reads them sequentially)
  uint32_t* hist;               // [bk_blocks(P)][2048] bucket histograms of the partition -> absolute output offsets
  uint32_t* bucket_start;       // [2049] first sorted position of each bucket (+ total)
  uint32_t* chunk_inst;         // [P / GM_SCAN_ITEMS + 1] instances emitted by each run of GM_SCAN_ITEMS sorted positions (zeroed with the slots)
  uint32_t* dmap;               // [GM_COARSE_BINS] coarse bin -> first bucket << 16 | buckets (written by the partition's histogram launch)
  uint32_t* bmap;               // [2048][2] bucket -> {first key, bits of (key - first key) inside the bucket}
  uint32_t* slots;              // [GM_SLOTS][GM_SLOT_STRIDE] (4 words used); slots, counters, coarse and acc are contiguous: one memset re-arms all of them
  uint32_t* coarse;             // [GM_COARSE_COPIES][GM_COARSE_BINS] coarse histogram of the visible depth keys
  uint32_t* counters;           // [GM_CNT_COUNT] device scalars
  uint32_t* acc;                // [bk_acc_words(P)] accumulators of the partition pass
  float* grad_acc;              // [P][12] backward accumulators: dcolor rgb | dmean2D xy | dconic x,y,w | dopacity | pad
  static GeomState from(void* buf, size_t P) {
    char* p = reinterpret_cast<char*>(buf);
    GeomState g;
    constexpr size_t ND = size_t(1) << GM_BUCKET_BITS;
    g.splat = reinterpret_cast<float4*>(carve<float>(p, GM_SPLAT_STRIDE * P + 4));
    g.radii = carve<int>(p, P);
    g.tiles_touched = carve<uint32_t>(p, P);
    g.bin = carve<uint4>(p, P);
    g.cov3D = carve<float>(p, 6 * P);
    g.clamped = carve<uint8_t>(p, P);
    g.depth_key = carve<uint32_t>(p, P);
    g.dpairs[0] = carve<uint2>(p, P);
    g.dpairs[1] = carve<uint2>(p, P);
    g.order = carve<uint32_t>(p, P);
    g.bin_sorted = carve<uint4>(p, P);
    g.hist = carve<uint32_t>(p, ND * bk_blocks(P));
    g.bucket_start = carve<uint32_t>(p, ND + 1);
    g.dmap = carve<uint32_t>(p, GM_COARSE_BINS);
    g.bmap = carve<uint32_t>(p, 2 * ND);
    const size_t nchunk = (P / GM_SCAN_ITEMS + 1 + 63) & ~size_t(63);      // (the armed block stays a multiple of 256 bytes: one fill kernel)
    g.slots = carve<uint32_t>(p, GM_SLOT_STRIDE * GM_SLOTS + GM_CNT_COUNT + GM_COARSE_COPIES * GM_COARSE_BINS + bk_acc_words(P) + nchunk);
    g.counters = g.slots + GM_SLOT_STRIDE * GM_SLOTS;
    g.coarse = g.counters + GM_CNT_COUNT;
    g.acc = g.coarse + GM_COARSE_COPIES * GM_COARSE_BINS;
    g.chunk_inst = g.acc + bk_acc_words(P);
    g.arm_words = GM_SLOT_STRIDE * GM_SLOTS + GM_CNT_COUNT + GM_COARSE_COPIES * GM_COARSE_BINS + bk_acc_words(P) + nchunk;
    g.grad_acc = carve<float>(p, 12 * P);
    g.end = p;
    return g;
  }
  size_t arm_words;
  char* end;
};

// Direct depth placement (edit-loop frames of one view stream, gm_forward_0_deformed_stream_async): the fused preprocess kernel
// looks every visible Gaussian's depth bucket up in a table built from an EARLIER frame of the stream (any monotone table gives the
// same order; only the balance of the buckets depends on it) and appends (key, id) and the emission record to that bucket's slab.
// bk_hist / bk_scan / bk_scatter of the depth partition and the random gather of the records disappear (gm_bucket.hip, "C.").
#define GM_PLAN_SLOTS 16             // tables a DepthPlan keeps: the writer of sequence number q uses slot q % 16, readers take the newest complete one
#define GM_PLAN_WORDS (2 + GM_PLAN_SLOTS + GM_PLAN_SLOTS * GM_COARSE_BINS)     // {newest sequence, sequence allocator}, stamps, tables
#define GM_SLAB_CNT_STRIDE 32        // words between bucket counters: one 128-byte line each (see GM_SLOT_STRIDE)
#define GM_SLAB_HDR_WORDS 64         // {sequence number of the frame, table valid, bucket capacity}
static inline uint32_t slab_capacity(size_t P) {       // entries a bucket's slab holds: 16x the mean of P keys over 2048 buckets, 256 .. 4096 (the in-LDS sort's limit)
  size_t c = 256;
  while (c < 4096 && c * 128 < P) c <<= 1;
  return (uint32_t)c;
}
struct DepthSlab {              // per frame in flight (caller-owned, gm_depth_slab_bytes(P))
  uint32_t* hdr;                // [GM_SLAB_HDR_WORDS]
  uint32_t* cnt;                // [2048][GM_SLAB_CNT_STRIDE] entries appended to each bucket (zeroed by the arm kernel)
  uint2* pairs;                 // [2048][cap] (depth key, id) in arrival order
  uint4* recs;                  // [2048][cap] emission records, same positions
  uint32_t cap;
  static DepthSlab from(void* buf, size_t P) {
    char* p = reinterpret_cast<char*>(buf);
    DepthSlab d;
    constexpr size_t ND = size_t(1) << GM_BUCKET_BITS;
    d.cap = slab_capacity(P);
    d.hdr = carve<uint32_t>(p, GM_SLAB_HDR_WORDS);
    d.cnt = carve<uint32_t>(p, ND * GM_SLAB_CNT_STRIDE);
    d.pairs = carve<uint2>(p, ND * d.cap);
    d.recs = carve<uint4>(p, ND * d.cap);
    d.end = p;
    return d;
  }
  char* end;
};

struct ImageState {             // per-pixel / per-tile state
  float* final_T;               // [H*W]
  uint32_t* n_contrib;          // [H*W]
  uint2* ranges;                // [tiles]
  uint32_t* tile_order;         // [tiles] list tiles by descending list length (the forward blend's dispatch order)
  uint32_t* tile_work;          // [tiles] deepest contributor (max n_contrib) among the pixels of a list tile's area (backward); during a forward: scratch of
                                //         the dispatch-order sort (gm_tile_order.h)
  uint32_t* tile_order_bwd;     // [tiles] list tiles by descending tile_work (the backward blend's dispatch order)
  uint32_t* epoch;              // [1] frame number of the work hint this frame's blend tags its entries with (gm_tile_order.h)
  static ImageState from(void* buf, int W, int H) {
    char* p = reinterpret_cast<char*>(buf);
    const size_t N = (size_t)W * H;
    const size_t T = (size_t)((W + GM_TILE - 1) / GM_TILE) * ((H + GM_TILE - 1) / GM_TILE);
    ImageState s;
    s.final_T = carve<float>(p, N);
    s.n_contrib = carve<uint32_t>(p, N);
    s.ranges = carve<uint2>(p, T);
    s.tile_order = carve<uint32_t>(p, T);
    s.tile_work = carve<uint32_t>(p, T);
    s.tile_order_bwd = carve<uint32_t>(p, T);
    s.epoch = carve<uint32_t>(p, 1);
    s.end = p;
    return s;
  }
  char* end;
};

struct BinningState {           // per-instance state (R-sized)
  uint2* pairs[2];              // [R] (list tile id | child mask << 16, Gaussian id) per instance, ping-pong
  uint32_t* hist;               // [bk_blocks(R)][2048] -> absolute output offsets
  uint32_t* acc;                // [bk_acc_words(R)] accumulators of a tile pass (zeroed by duplicate_kernel)
  static BinningState from(void* buf, size_t R) {
    char* p = reinterpret_cast<char*>(buf);
    BinningState b;
    const size_t Rp = R ? R : 1;
    constexpr size_t ND = size_t(1) << GM_BUCKET_BITS;
    b.pairs[0] = carve<uint2>(p, Rp);
    b.pairs[1] = carve<uint2>(p, Rp);
    b.hist = carve<uint32_t>(p, ND * bk_blocks(Rp));
    b.acc = carve<uint32_t>(p, bk_acc_words(Rp));
    b.end = p;
    return b;
  }
  char* end;
};

// Frames of ONE batched forward (gm_forward_deformed_batch_async): every kernel of the chain runs with gridDim.z = frames in the batch and
// takes frame 0's pointers plus, per caller-owned buffer kind, the byte distance of frame f's buffer from frame 0's - the scratch layouts
// depend only on (base address mod 256, sizes), and the batch entry point insists on 256-byte aligned bases, so a field of frame f is the
// same field of frame 0 moved by that distance.  A single-frame launch passes zeros and gridDim.z == 1.
#define GM_BATCH_MAX 8
struct FrameOfs { long long d[GM_BATCH_MAX]; };
template <class T>
__device__ __forceinline__ T* frame_ptr(T* p, const FrameOfs& o) {       // (pointer arithmetic on p itself: the result keeps p's __restrict__ provenance)
  typedef typename std::conditional<std::is_const<T>::value, const char, char>::type B;
  return p ? reinterpret_cast<T*>(reinterpret_cast<B*>(p) + o.d[blockIdx.z]) : p;
}
struct BatchOfs {                 // host side: the distances of a batch (all zero + frames == 1 for a single frame)
  int frames;
  FrameOfs geom, binning, image, color;
  int* status[GM_BATCH_MAX];
};
static inline BatchOfs single_frame() { BatchOfs b{}; b.frames = 1; return b; }

static inline int tile_bits(int tiles) {      // bits needed to hold tile ids 0..tiles-1 (>=1)
  int b = 1;
  while ((1 << b) < tiles) b++;
  return b;
}
// which ping-pong slot holds the tile-sorted instance stream (launch_tile_sort: one 11-bit pass for up to 2048 list tiles,
// two 8-bit passes above)
static inline int sort_final_slot(int tiles) { return tiles <= (1 << GM_BUCKET_BITS) ? 1 : 0; }

// Emission policy (gm_set_tile_culling): 0 = the reference's lists (every tile of the rectangle, 16-px tiles);
// 1 = exact culling, lists per 16-px tile; 2 / 3 = exact culling, lists per 32- / 64-px PARENT tile (the instance
// stream shrinks ~2x / ~3x): an instance is one (Gaussian, parent tile) pair whose key carries, above bit 16, the
// mask of the parent's children the Gaussian reaches: policy 3 one bit per 16-px child tile (4 x 4), policy 2 one bit per
// 8x8-pixel QUADRANT of the 32-px parent (4 x 4: bit 4 qy + qx; gm_cull.h quad_rect).  The blend kernels run one wave per
// quadrant; it walks its parent's list and takes the entries whose mask has its tile's / its quadrant's bit.
#define GM_KEY_MASK_SHIFT 16
#define GM_KEY_TILE_MASK 0xFFFFu
static inline int tile_shift_of(int mode) { return mode >= 2 ? mode - 1 : 0; }
struct TileGrid {
  int s, gx, gy, pgx, pgy, ptiles;
  TileGrid(int W, int H, int mode) {
    s = tile_shift_of(mode);
    gx = (W + GM_TILE - 1) / GM_TILE; gy = (H + GM_TILE - 1) / GM_TILE;
    pgx = (gx + (1 << s) - 1) >> s; pgy = (gy + (1 << s) - 1) >> s;
    ptiles = pgx * pgy;
  }
};

// ---------------------------------------------------------------------------------------------
// host launchers implemented in the other translation units (all stream-ordered, return GM_* codes)
struct RasterArgs {
  int P, D, M, W, H;
  const float *background, *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
  const float *viewmatrix, *projmatrix, *cam_pos;
  float scale_modifier, tan_fovx, tan_fovy;
  int prefiltered, debug;
  int tile_cull;                // emission policy 0..3, see tile_shift_of()
  hipStream_t stream;
};

int launch_preprocess(const RasterArgs& a, GeomState& g, int* radii);
int launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s);
int launch_preprocess_bwd(const RasterArgs& a, GeomState& g, const int* radii, float* dL_dmean2D, float* dL_dconic,
                          float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                          float* dL_dscale, float* dL_drot, const struct ShAdamArgs* sh_adam = nullptr);   // reads g.grad_acc, writes every gradient output

// stable LSD radix sort of (u32 key, u32 value) pairs on key bits [0, bits), 8 bits per pass, ping-pong
// between slot 0 and slot 1.  n_dev (optional) = device pointer to the element count; n_max = host upper bound
// used for grid sizing.  iota_values: pass 1 synthesises values = index instead of reading vals[0].
int radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], uint32_t* hist, uint32_t* digit_total, size_t n,
                     int bits, bool iota_values, int debug, hipStream_t s);

// (depth, id) order of the visible Gaussians + per-bucket instance totals; counters[GM_CNT_RENDERED] = num_rendered.
// num_rendered_host (optional, pinned): receives the count by a stream-ordered copy issued as soon as it is known;
// count_event (optional) is recorded right behind that copy.
int launch_depth_order(GeomState& g, int P, int debug, hipStream_t s, int* num_rendered_host, hipEvent_t count_event, const BatchOfs* bt = nullptr);
int launch_arm_counters(GeomState& g, hipStream_t s, const BatchOfs* bt = nullptr);   // zero slots + counters (first launch of a forward)
// direct depth placement: arm (zero + table snapshot into g.dmap), and - after the fused preprocess - bucket starts, the next table, the sort
int launch_arm_direct(GeomState& g, DepthSlab& d, uint32_t* plan, hipStream_t s);
int launch_depth_order_direct(GeomState& g, DepthSlab& d, uint32_t* plan, int P, int debug, hipStream_t s, int* num_rendered_host, hipEvent_t count_event);
int launch_publish_depth_plan(GeomState& g, uint32_t* plan, hipStream_t s);        // classic path: leave this frame's table for the stream's next frames
int launch_duplicate(GeomState& g, BinningState& b, int P, int W, int H, int tile_cull, size_t capacity, int debug, hipStream_t s, const BatchOfs* bt = nullptr);
int launch_tile_sort(GeomState& g, BinningState& b, ImageState& img, size_t n, const uint32_t* n_dev, int tiles, bool* order_done,
                     uint32_t* work_hint, int debug, hipStream_t s, const BatchOfs* bt = nullptr);      // order_done: img.tile_order was written too (one-pass case)
int launch_tile_ranges(const GeomState& g, BinningState& b, int slot, ImageState& img, int R, const uint32_t* R_dev, int tiles, int debug, hipStream_t s);
int launch_tile_order(ImageState& img, int tiles, uint32_t* work_hint, int debug, hipStream_t s);          // ranges -> tile_order
int launch_render_fwd(const GeomState& g, const uint2* pairs, ImageState& img, int W, int H, int mode,
                      const float* background, float* out_color, int* status_host, bool image_only, uint32_t* work_hint, int debug,
                      hipStream_t s, bool exact_exponent = false, const BatchOfs* bt = nullptr);
int launch_render_bwd(const GeomState& g, const uint2* pairs, ImageState& img, int W, int H, int mode,
                      const float* background, const float* dL_dpix, int debug, hipStream_t s);   // accumulates into g.grad_acc

int launch_deform(int N, const int* tri, const float* w, const float* dV, const float* Rv, const float* Sv,
                  const float* cov, const float* pos, float* pos_out, float* cov_out, float* rot_out, float* cov6_out,
                  hipStream_t s);
int launch_sh_colors(int N, int deg, int M, const float* pos, const float* campos, const float* rot,
                     const float* shs, float* rgb, hipStream_t s);
int launch_deform_shade(int N, int deg, int M, const int* tri, const float* w, const float* dV, const float* Rv, const float* Sv,
                        const float* cov, const float* pos, const float* shs, const float* campos, float* pos_out,
                        float* cov6_out, float* rgb_out, float* cov_out, float* rot_out, hipStream_t s);
int launch_deform_shade_pre(const RasterArgs& r, GeomState& g, int* radii, int deg, const int* tri, const float* w, const float* packed,
                            const float* cov, const float* pos, const float* shs, float* pos_out, float* cov6_out, float* rgb_out,
                            const struct DepthSlab* slab = nullptr, bool cov6 = false);
int launch_pack_mesh_state(int Vm, const float* state, const float* verts, float* packed, hipStream_t s);
int launch_deform_shade_packed(int N, int deg, int M, const int* tri, const float* w, const float* packed, const float* cov,
                               const float* pos, const float* shs, const float* campos, float* pos_out, float* cov6_out,
                               float* rgb_out, float* cov_out, float* rot_out, hipStream_t s);
int launch_cov_to_scale_rot(int N, const float* cov, float* scales, float* rots, hipStream_t s);
int launch_mesh_rs(int Vm, const float* V0, const float* V1, const int* faces, const int* adj_offsets, const int* adj_faces, float* R,
                   float* S, float* state, float* packed, hipStream_t s);
// the packed gather tables of `frames` deformation frames in one launch (gridDim.z = frame): V1[f] -> packed[f]
int launch_mesh_rs_batch(int frames, int Vm, const float* V0, const float* const* V1, const int* faces, const int* adj_offsets, const int* adj_faces,
                         float* const* packed, hipStream_t s);
// one frame of a batched fused pass (gm_deform.hip): what differs between the frames that share one pass over the static cloud
struct BatchFrameArgs {
  const float *packed, *viewmatrix, *projmatrix, *cam_pos;
  float tan_fovx, tan_fovy;
  GeomState g;
  int* radii;
};
int launch_deform_shade_pre_batch(int frames, const BatchFrameArgs* fr, int P, int deg, int W, int H, int tile_cull, const int* tri, const float* w,
                                  const float* cov, const float* pos, const float* shs, const float* opacities, bool cov6, int debug, hipStream_t s);
int launch_ssim_fwd(const float* img1, const float* img2, int planes, int H, int W, float* d_mu1, float* d_e11, float* d_e12,
                    float* partial, hipStream_t s);
int launch_loss_combine(const float* partial, long long n, double c_ssim, double c_l1, double offset, float* out, hipStream_t s);
int launch_ssim_bwd(const float* img1, const float* img2, const float* d_mu1, const float* d_e11, const float* d_e12, int planes,
                    int H, int W, const float* g_ssim, const float* g_l1, float* dL_dimg1, hipStream_t s);
struct ActArgs {                 // mesh-bound parameter -> rasterizer-input map (gm_train.hip)
  int N;
  float alpha;
  const float *bc, *dist, *scaling, *rotation, *opacity, *v1, *v2, *v3, *normal, *r;
};
struct AdamTensor {
  float* p; const float* g; float* m; float* v;
  unsigned long long n;
  float step_lo, step_hi;      // lr * sqrt(1-b2^t)/(1-b1^t) for elements with (index % period) < split / the others
  unsigned period, split;      // period == 0: one rate (step_lo) for the whole tensor
  unsigned active;             // 0: every element; else only elements with (index % period) < active (rounded up to a 16-byte granule) are touched
};
struct AdamTable { AdamTensor t[8]; int count; float b1, b2, eps, omb1, omb2; };   // omb = (float)(1 - beta), formed in double
// jittor.nn.Adam's rule for one element (jittor/optim.py Adam.step), spelled with explicit fused multiply-adds so that adam_kernel
// (gm_train.hip, contraction allowed) and the SH step fused into the preprocess backward (gm_preprocess.hip, contraction off) round alike
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, float b1, float b2, float c1, float c2, float eps, float st) {
  m = __builtin_fmaf(b1, m, c1 * g);
  v = __builtin_fmaf(b2, v, c2 * g * g);
  p -= st * m / (sqrtf(v) + eps);
}
// Adam step of the SH rows applied INSIDE the preprocess backward (gm_backward_sh_step): the row's gradient never travels through HBM
struct ShAdamArgs {
  float *p, *m, *v;              // the [rows,16,3] parameter (the `shs` operand's leading rows), exp_avg, exp_avg_sq
  int rows;                      // trainable rows (the operand may continue with frozen rows behind them)
  float b1, b2, c1, c2, eps, step_lo, step_hi;   // step_lo: coefficient 0 (the reference's "f_dc" group), step_hi: the others ("f_rest")
};
int launch_mesh_activate_fwd(const ActArgs& a, float* xyz, float* scales, float* rots, float* opac, float mr_weight, float* mr_partial,
                             hipStream_t s);
int launch_mesh_activate_bwd(const ActArgs& a, const float* d_xyz, const float* d_scales, const float* d_rots, const float* d_opac,
                             float* d_bc, float* d_dist, float* d_scaling, float* d_rotation, float* d_opacity, float mr_weight,
                             const float* d_mr, hipStream_t s);
int launch_adam(const AdamTable& tab, hipStream_t s);
int launch_densify_stats(int N, const int* radii, const float* grad2d, float* max_radii2D, float* grad_accum, float* denom, hipStream_t s);
int launch_knn(int P, const float* points, float* meanDists, void* ws, size_t ws_bytes, hipStream_t s);
size_t knn_workspace_bytes(int P);

// number of set bits of a wave-wide 64-bit mask (a ballot) at positions BELOW the calling lane: v_mbcnt_lo + v_mbcnt_hi, two
// vector instructions and no per-lane mask registers (popcount(mask & lanes_lt) costs four and two registers)
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

}  // namespace gm
