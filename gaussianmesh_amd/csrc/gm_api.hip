// gm_api.hip -- extern "C" entry points of libgmesh_hip.so (include/gmesh_hip.h) and host orchestration.
//
// Orchestration replaces CudaRasterizer::Rasterizer::{forward_0, forward_1, backward, markVisible}
// (cuda_rasterizer/rasterizer_impl.cu:338-413, 416-511, 515-609, 141-153).
#include "gm_common.h"
#include <cmath>
#include <cstdlib>
#include "../../include/gmesh_hip.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace gm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) at %s:%d in %s", (int)e, hipGetErrorString(e), file, line, what);
  return GM_ERR_HIP;
}

// ---- per-stage profiling -------------------------------------------------------------------
static const char* kStageNames[ST_COUNT] = {"preprocess", "depth_sort", "scan", "duplicate", "tile_sort", "ranges",
                                            "render", "render_bwd", "preprocess_bwd", "deform", "sh_colors", "loss", "loss_bwd", "mesh_rs"};
struct EvPair { hipEvent_t a, b; int st; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<EvPair> g_pending;
static double g_ms[ST_COUNT];
static int64_t g_n[ST_COUNT];

StageScope::StageScope(Stage st_, hipStream_t s_) : st(st_), s(s_), rec(nullptr) {
  if (!g_prof_on) return;
  EvPair* p = new EvPair;
  p->st = st;
  if (hipEventCreate(&p->a) != hipSuccess || hipEventCreate(&p->b) != hipSuccess) { delete p; return; }
  (void)hipEventRecord(p->a, s);
  rec = p;
}
StageScope::~StageScope() {
  if (!rec) return;
  EvPair* p = reinterpret_cast<EvPair*>(rec);
  (void)hipEventRecord(p->b, s);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_pending.push_back(*p);
  delete p;
}
static void drain_profile() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& p : g_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      g_ms[p.st] += ms;
      g_n[p.st] += 1;
    }
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  g_pending.clear();
}

static int check_raster_args(const RasterArgs& a) {
  if (a.P < 0 || a.W <= 0 || a.H <= 0) { set_error("invalid sizes P=%d W=%d H=%d", a.P, a.W, a.H); return GM_ERR_INVALID_ARG; }
  if (TileGrid(a.W, a.H, a.tile_cull).ptiles > 65536) {
    set_error("%dx%d has more than 65536 list tiles under emission policy %d; use policy 2 or 3", a.W, a.H, a.tile_cull);
    return GM_ERR_INVALID_ARG;
  }
  if (a.P == 0) return 0;                       // empty cloud: every per-Gaussian pointer may be null
  if ((a.shs == nullptr) == (a.colors_precomp == nullptr)) {
    set_error("provide exactly one of shs / colors_precomp"); return GM_ERR_INVALID_ARG;
  }
  const bool sr = a.scales != nullptr && a.rotations != nullptr;
  if ((a.scales != nullptr) != (a.rotations != nullptr) || sr == (a.cov3D_precomp != nullptr)) {
    set_error("provide exactly one of (scales, rotations) / cov3D_precomp"); return GM_ERR_INVALID_ARG;
  }
  if (a.shs && (a.D < 0 || a.D > 3 || a.M < (a.D + 1) * (a.D + 1))) {
    set_error("SH degree %d needs M >= %d coefficients (got %d); degree must be 0..3", a.D, (a.D + 1) * (a.D + 1), a.M);
    return GM_ERR_INVALID_ARG;
  }
  if (a.P > 0 && (!a.means3D || !a.opacities || !a.viewmatrix || !a.projmatrix || !a.cam_pos)) {
    set_error("null required input"); return GM_ERR_INVALID_ARG;
  }
  return 0;
}

}  // namespace gm

using namespace gm;

extern "C" {

int gm_abi_version(void) { return GM_ABI_VERSION; }
const char* gm_last_error(void) { return g_err; }

size_t gm_geom_bytes(int P) {
  GeomState g = GeomState::from(nullptr, (size_t)(P > 0 ? P : 1));
  return (size_t)g.end + 256;
}
size_t gm_image_bytes(int W, int H) {
  ImageState s = ImageState::from(nullptr, W > 0 ? W : 1, H > 0 ? H : 1);
  return (size_t)s.end + 256;
}
size_t gm_work_hint_bytes(int W, int H) {          // frame counter + one word per list tile (at most one per 16-px tile)
  const size_t T = (size_t)((W > 0 ? W : 1) + GM_TILE - 1) / GM_TILE * (size_t)(((H > 0 ? H : 1) + GM_TILE - 1) / GM_TILE);
  return 4 * (T + 1);
}
size_t gm_binning_bytes(int64_t R) {
  BinningState b = BinningState::from(nullptr, (size_t)(R > 0 ? R : 1));
  return (size_t)b.end + 256;
}

static int check_policy(int p) {
  if (p < 0 || p > 3) { set_error("emission policy %d out of range 0..3", p); return GM_ERR_INVALID_ARG; }
  return 0;
}

#define FILL_ARGS(a, POLICY)                                                                                    \
  RasterArgs a;                                                                                                 \
  a.P = P; a.D = D; a.M = M; a.W = width; a.H = height; a.background = background; a.means3D = means3D;        \
  a.shs = shs; a.colors_precomp = colors_precomp; a.opacities = opacities; a.scales = scales;                  \
  a.rotations = rotations; a.cov3D_precomp = cov3D_precomp; a.viewmatrix = viewmatrix; a.projmatrix = projmatrix; \
  a.cam_pos = cam_pos; a.scale_modifier = scale_modifier; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy;        \
  a.prefiltered = prefiltered; a.debug = debug; a.tile_cull = (POLICY); a.stream = reinterpret_cast<hipStream_t>(stream);

// Measurement aid (tools/stage_marginal.sh): GM_DEBUG_STOP_AFTER = deform | depth | dup | tile in the environment makes every forward
// stop launching after that stage - the frames are garbage, the loop's rate tells what the remaining stages cost the pipeline.
static int debug_stop_after() {
  static const int v = [] {
    const char* e = getenv("GM_DEBUG_STOP_AFTER");
    if (!e) return 0;
    return !strcmp(e, "deform") ? 1 : !strcmp(e, "depth") ? 2 : !strcmp(e, "dup") ? 3 : !strcmp(e, "tile") ? 4 : 0;
  }();
  return v;
}

// first half of a forward after the per-Gaussian kernel: order the visible Gaussians, hand the instance count to the host
static int order_and_count(GeomState& g, int P, int debug, hipStream_t st, int* num_rendered_host, void* count_event) {
  if (debug_stop_after() == 1) return GM_OK;
  return launch_depth_order(g, P, debug, st, num_rendered_host, reinterpret_cast<hipEvent_t>(count_event));
}

int gm_forward_0_async(int emission_policy, void* geom_buffer, int P, int D, int M, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                       const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                       float tan_fovy, int prefiltered, int* radii, int debug, void* stream, int* num_rendered_host,
                       void* count_event) {
  if (int rc = check_policy(emission_policy)) return rc;
  FILL_ARGS(a, emission_policy)
  if (int rc = check_raster_args(a)) return rc;
  if (P == 0) { if (num_rendered_host) *num_rendered_host = 0; return GM_OK; }
  if (!geom_buffer) { set_error("geom_buffer is null"); return GM_ERR_INVALID_ARG; }
  GeomState g = GeomState::from(geom_buffer, (size_t)P);
  if (int rc = launch_arm_counters(g, a.stream)) return rc;
  if (int rc = launch_preprocess(a, g, radii)) return rc;
  return order_and_count(g, P, debug, a.stream, num_rendered_host, count_event);
}

size_t gm_depth_plan_bytes(void) { return sizeof(uint32_t) * GM_PLAN_WORDS; }
size_t gm_depth_slab_bytes(int P) {
  DepthSlab d = DepthSlab::from(nullptr, (size_t)(P > 0 ? P : 1));
  return (size_t)d.end + 256;
}

int gm_forward_0_deformed_async(int emission_policy, void* geom_buffer, int P, int deg, int M, int width, int height, const int* tri,
                                const float* w, const float* packed, const float* cov, const float* pos, const float* shs,
                                const float* opacities, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                float tan_fovx, float tan_fovy, float* pos_out, float* cov6_out, float* rgb_out, int* radii, int debug,
                                void* stream, int* num_rendered_host, void* count_event) {
  return gm_forward_0_deformed_stream_async(emission_policy, geom_buffer, P, deg, M, width, height, tri, w, packed, cov, pos, shs, opacities,
                                            viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy, pos_out, cov6_out, rgb_out, radii, debug, stream,
                                            num_rendered_host, count_event, nullptr, nullptr, 0);
}

int gm_forward_0_deformed_stream_async(int emission_policy, void* geom_buffer, int P, int deg, int M, int width, int height, const int* tri,
                                       const float* w, const float* packed, const float* cov, const float* pos, const float* shs,
                                       const float* opacities, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                       float tan_fovx, float tan_fovy, float* pos_out, float* cov6_out, float* rgb_out, int* radii,
                                       int debug, void* stream, int* num_rendered_host, void* count_event, void* depth_slab,
                                       unsigned int* depth_plan, int flags) {
  if (int rc = check_policy(emission_policy)) return rc;
  if (flags & ~(GM_STREAM_DIRECT | GM_STREAM_COV6)) { set_error("gm_forward_0_deformed_stream: unknown flags 0x%x", flags); return GM_ERR_INVALID_ARG; }
  const int direct = flags & GM_STREAM_DIRECT;
  const bool cov6 = (flags & GM_STREAM_COV6) != 0;
  if (direct && (!depth_slab || !depth_plan)) { set_error("gm_forward_0_deformed_stream: direct placement needs depth_slab and depth_plan"); return GM_ERR_INVALID_ARG; }
  if (P < 0 || width <= 0 || height <= 0) { set_error("invalid sizes P=%d W=%d H=%d", P, width, height); return GM_ERR_INVALID_ARG; }
  if (P == 0) { if (num_rendered_host) *num_rendered_host = 0; return GM_OK; }
  if (deg < 0 || deg > 3 || M != 16) { set_error("gm_forward_0_deformed: needs SH rows of M == 16 coefficients, degree 0..3"); return GM_ERR_INVALID_ARG; }
  if (!geom_buffer || !tri || !w || !packed || !cov || !pos || !shs || !opacities || !viewmatrix || !projmatrix || !cam_pos) {
    set_error("gm_forward_0_deformed: null required input"); return GM_ERR_INVALID_ARG;
  }
  const int nout = (pos_out != nullptr) + (cov6_out != nullptr) + (rgb_out != nullptr);
  if (nout != 0 && nout != 3) { set_error("gm_forward_0_deformed: pass pos_out, cov6_out and rgb_out together or none"); return GM_ERR_INVALID_ARG; }
  RasterArgs a{};
  a.P = P; a.D = deg; a.M = M; a.W = width; a.H = height; a.opacities = opacities; a.viewmatrix = viewmatrix; a.projmatrix = projmatrix;
  a.cam_pos = cam_pos; a.scale_modifier = 1.0f; a.tan_fovx = tan_fovx; a.tan_fovy = tan_fovy; a.debug = debug; a.tile_cull = emission_policy;
  a.stream = reinterpret_cast<hipStream_t>(stream);
  if (TileGrid(width, height, emission_policy).ptiles > 65536) { set_error("too many list tiles for emission policy %d", emission_policy); return GM_ERR_INVALID_ARG; }
  GeomState g = GeomState::from(geom_buffer, (size_t)P);
  if (direct) {
    DepthSlab d = DepthSlab::from(depth_slab, (size_t)P);
    if (int rc = launch_arm_direct(g, d, depth_plan, a.stream)) return rc;
    if (int rc = launch_deform_shade_pre(a, g, radii, deg, tri, w, packed, cov, pos, shs, pos_out, cov6_out, rgb_out, &d, cov6)) return rc;
    if (debug_stop_after() == 1) return GM_OK;
    return launch_depth_order_direct(g, d, depth_plan, P, debug, a.stream, num_rendered_host, reinterpret_cast<hipEvent_t>(count_event));
  }
  if (int rc = launch_arm_counters(g, a.stream)) return rc;
  if (int rc = launch_deform_shade_pre(a, g, radii, deg, tri, w, packed, cov, pos, shs, pos_out, cov6_out, rgb_out, nullptr, cov6)) return rc;
  if (int rc = order_and_count(g, P, debug, a.stream, num_rendered_host, count_event)) return rc;
  if (depth_plan && debug_stop_after() != 1) return launch_publish_depth_plan(g, depth_plan, a.stream);
  return GM_OK;
}

int gm_forward_1_geom(int emission_policy, void* geom_buffer, void* binning_buffer, void* image_buffer, int P, int num_rendered,
                      int64_t binning_capacity, const float* background, int width, int height, float* out_color, int debug, void* stream,
                      int* status_host, int flags, unsigned int* work_hint) {
  if (int rc = check_policy(emission_policy)) return rc;
  if (flags & ~(GM_FWD_IMAGE_ONLY | GM_FWD_EXACT_EXPONENT)) { set_error("unknown flags 0x%x", flags); return GM_ERR_INVALID_ARG; }
  if ((flags & GM_FWD_IMAGE_ONLY) && (flags & GM_FWD_EXACT_EXPONENT)) {
    set_error("GM_FWD_EXACT_EXPONENT is for a forward a backward pass follows: not together with GM_FWD_IMAGE_ONLY"); return GM_ERR_INVALID_ARG;
  }
  if (P < 0 || width <= 0 || height <= 0) { set_error("invalid sizes P=%d W=%d H=%d", P, width, height); return GM_ERR_INVALID_ARG; }
  if (!image_buffer || !out_color || !background) { set_error("null image_buffer / out_color / background"); return GM_ERR_INVALID_ARG; }
  const bool device_count = num_rendered < 0;          // sync-free: the count stays on the device, bounded by the capacity
  const int64_t cap = device_count ? binning_capacity : (int64_t)num_rendered;
  if (cap < 0) { set_error("negative binning capacity"); return GM_ERR_INVALID_ARG; }
  if (P > 0 && (!geom_buffer || (cap > 0 && !binning_buffer))) { set_error("null scratch buffer"); return GM_ERR_INVALID_ARG; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int mode = emission_policy;
  ImageState img = ImageState::from(image_buffer, width, height);
  const int tiles = TileGrid(width, height, mode).ptiles;
  if (tiles > 65536) { set_error("too many list tiles for emission policy %d", mode); return GM_ERR_INVALID_ARG; }
  GeomState g = GeomState::from(geom_buffer, (size_t)(P > 0 ? P : 1));
  BinningState b = BinningState::from(binning_buffer, (size_t)cap);
  const int slot = sort_final_slot(tiles);
  bool order_done = false;
  if (const int stop = debug_stop_after()) {
    if (stop <= 2) return GM_OK;
    if (stop == 3) return launch_duplicate(g, b, P, width, height, mode, (size_t)cap, debug, st);
  }
  if (P > 0 && cap > 0) {
    if (int rc = launch_duplicate(g, b, P, width, height, mode, (size_t)cap, debug, st)) return rc;
    if (int rc = launch_tile_sort(g, b, img, (size_t)cap, device_count ? g.counters + GM_CNT_RENDERED : nullptr, tiles, &order_done, work_hint, debug, st)) return rc;
    if (slot == 0)
      if (int rc = launch_tile_ranges(g, b, slot, img, (int)cap, device_count ? g.counters + GM_CNT_RENDERED : nullptr, tiles, debug, st)) return rc;
  } else {
    GM_HIP(hipMemsetAsync(img.ranges, 0, sizeof(uint2) * (size_t)tiles, st));
  }
  if (!order_done)
    if (int rc = launch_tile_order(img, tiles, work_hint, debug, st)) return rc;
  if (status_host && P == 0) {                       // no geometry state to report from
    status_host[0] = 0; status_host[1] = 0; status_host[2] = mode; status_host[3] = 0;
    status_host = nullptr;
  }
  if (debug_stop_after() == 4) return GM_OK;
  return launch_render_fwd(g, b.pairs[slot], img, width, height, mode, background, out_color, status_host, (flags & GM_FWD_IMAGE_ONLY) != 0,
                           work_hint, debug, st, (flags & GM_FWD_EXACT_EXPONENT) != 0);
}

int gm_forward_deformed_batch_async(int emission_policy, int K, const gm_batch_frame* frames, int P, int deg, int M, int width, int height,
                                    const int* tri, const float* w, const float* cov, const float* pos, const float* shs, const float* opacities,
                                    const float* background, int64_t binning_capacity, int flags, unsigned int* work_hint, int debug, void* stream) {
  if (int rc = check_policy(emission_policy)) return rc;
  if (K < 1 || K > GM_BATCH_MAX || !frames) { set_error("gm_forward_deformed_batch: 1..%d frames", GM_BATCH_MAX); return GM_ERR_INVALID_ARG; }
  if (flags & ~(GM_BATCH_IMAGE_ONLY | GM_BATCH_COV6)) { set_error("gm_forward_deformed_batch: unknown flags 0x%x", flags); return GM_ERR_INVALID_ARG; }
  if (P <= 0 || width <= 0 || height <= 0) { set_error("gm_forward_deformed_batch: invalid sizes P=%d W=%d H=%d (an empty cloud goes through the single-frame calls)", P, width, height); return GM_ERR_INVALID_ARG; }
  if (deg < 0 || deg > 3 || M != 16) { set_error("gm_forward_deformed_batch: needs SH rows of M == 16 coefficients, degree 0..3"); return GM_ERR_INVALID_ARG; }
  if (!tri || !w || !cov || !pos || !shs || !opacities || !background) { set_error("gm_forward_deformed_batch: null required input"); return GM_ERR_INVALID_ARG; }
  if (binning_capacity <= 0) { set_error("gm_forward_deformed_batch: binning_capacity must be positive (sync-free second half)"); return GM_ERR_INVALID_ARG; }
  const TileGrid tg(width, height, emission_policy);
  if (tg.ptiles > (1 << GM_BUCKET_BITS)) {
    set_error("gm_forward_deformed_batch: %dx%d has %d list tiles under policy %d; a batch needs the one-pass tile sort (<= 2048)", width, height, tg.ptiles, emission_policy);
    return GM_ERR_INVALID_ARG;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  BatchFrameArgs fr[GM_BATCH_MAX];
  BatchOfs B{};
  B.frames = K;
  for (int k = 0; k < K; k++) {
    const gm_batch_frame& f = frames[k];
    if (!f.packed || !f.viewmatrix || !f.projmatrix || !f.cam_pos || !f.geom_buffer || !f.binning_buffer || !f.image_buffer || !f.out_color) {
      set_error("gm_forward_deformed_batch: frame %d has a null pointer", k); return GM_ERR_INVALID_ARG;
    }
    if ((reinterpret_cast<uintptr_t>(f.geom_buffer) | reinterpret_cast<uintptr_t>(f.binning_buffer) | reinterpret_cast<uintptr_t>(f.image_buffer)) & 255) {
      set_error("gm_forward_deformed_batch: frame %d: scratch buffers must be 256-byte aligned", k); return GM_ERR_INVALID_ARG;
    }
    for (int j = 0; j < k; j++)
      if (frames[j].geom_buffer == f.geom_buffer || frames[j].binning_buffer == f.binning_buffer || frames[j].image_buffer == f.image_buffer || frames[j].out_color == f.out_color) {
        set_error("gm_forward_deformed_batch: frames %d and %d share a buffer", j, k); return GM_ERR_INVALID_ARG;
      }
    fr[k].packed = f.packed; fr[k].viewmatrix = f.viewmatrix; fr[k].projmatrix = f.projmatrix; fr[k].cam_pos = f.cam_pos;
    fr[k].tan_fovx = f.tan_fovx; fr[k].tan_fovy = f.tan_fovy; fr[k].radii = f.radii;
    fr[k].g = GeomState::from(f.geom_buffer, (size_t)P);
    B.geom.d[k] = (long long)(reinterpret_cast<intptr_t>(f.geom_buffer) - reinterpret_cast<intptr_t>(frames[0].geom_buffer));
    B.binning.d[k] = (long long)(reinterpret_cast<intptr_t>(f.binning_buffer) - reinterpret_cast<intptr_t>(frames[0].binning_buffer));
    B.image.d[k] = (long long)(reinterpret_cast<intptr_t>(f.image_buffer) - reinterpret_cast<intptr_t>(frames[0].image_buffer));
    B.color.d[k] = (long long)(reinterpret_cast<intptr_t>(f.out_color) - reinterpret_cast<intptr_t>(frames[0].out_color));
    B.status[k] = f.status_host;
  }
  GeomState g = fr[0].g;
  ImageState img = ImageState::from(frames[0].image_buffer, width, height);
  BinningState b = BinningState::from(frames[0].binning_buffer, (size_t)binning_capacity);
  if (int rc = launch_arm_counters(g, st, &B)) return rc;
  if (int rc = launch_deform_shade_pre_batch(K, fr, P, deg, width, height, emission_policy, tri, w, cov, pos, shs, opacities, (flags & GM_BATCH_COV6) != 0, debug, st)) return rc;
  if (int rc = launch_depth_order(g, P, debug, st, nullptr, nullptr, &B)) return rc;
  if (int rc = launch_duplicate(g, b, P, width, height, emission_policy, (size_t)binning_capacity, debug, st, &B)) return rc;
  bool order_done = false;
  if (int rc = launch_tile_sort(g, b, img, (size_t)binning_capacity, g.counters + GM_CNT_RENDERED, tg.ptiles, &order_done, work_hint, debug, st, &B)) return rc;
  if (!order_done) { set_error("gm_forward_deformed_batch: internal: the tile pass did not produce the dispatch order"); return GM_ERR_INVALID_ARG; }
  return launch_render_fwd(g, b.pairs[sort_final_slot(tg.ptiles)], img, width, height, emission_policy, background, frames[0].out_color, frames[0].status_host,
                           (flags & GM_BATCH_IMAGE_ONLY) != 0, work_hint, debug, st, false, &B);
}

int gm_mesh_rs_packed_batch(int K, int Vm, int nfaces, const float* V0, const float* const* V1, const int* faces, const int* adj_offsets,
                            const int* adj_faces, float* const* packed, void* stream) {
  if (K < 1 || K > GM_BATCH_MAX || Vm < 0 || nfaces < 0 || !V1 || !packed || (Vm > 0 && (!V0 || !adj_offsets || (nfaces > 0 && (!faces || !adj_faces))))) {
    set_error("gm_mesh_rs_packed_batch: bad args"); return GM_ERR_INVALID_ARG;
  }
  for (int k = 0; k < K; k++)
    if (Vm > 0 && (!V1[k] || !packed[k] || (reinterpret_cast<uintptr_t>(packed[k]) & 15))) { set_error("gm_mesh_rs_packed_batch: frame %d: null or unaligned pointer", k); return GM_ERR_INVALID_ARG; }
  return launch_mesh_rs_batch(K, Vm, V0, V1, faces, adj_offsets, adj_faces, packed, reinterpret_cast<hipStream_t>(stream));
}

int gm_forward_status_async(void* geom_buffer, int P, int* status_host, void* stream) {
  if (!status_host) { set_error("status_host is null"); return GM_ERR_INVALID_ARG; }
  if (P <= 0) { status_host[0] = 0; status_host[1] = 0; status_host[2] = 0; status_host[3] = 0; return GM_OK; }
  if (!geom_buffer) { set_error("geom_buffer is null"); return GM_ERR_INVALID_ARG; }
  GeomState g = GeomState::from(geom_buffer, (size_t)P);
  GM_HIP(hipMemcpyAsync(status_host, g.counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream)));
  return GM_OK;
}

int gm_forward_0(void* geom_buffer, int P, int D, int M, const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                 const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                 const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                 float tan_fovy, int prefiltered, int* radii, int debug, void* stream, int* num_rendered) {
  if (!num_rendered) { set_error("num_rendered is null"); return GM_ERR_INVALID_ARG; }
  *num_rendered = 0;
  if (int rc = gm_forward_0_async(GM_POLICY_DEFAULT, geom_buffer, P, D, M, background, width, height, means3D, shs, colors_precomp, opacities,
                                  scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, tan_fovx, tan_fovy,
                                  prefiltered, radii, debug, stream, nullptr, nullptr)) return rc;
  if (P == 0) return GM_OK;
  GeomState g = GeomState::from(geom_buffer, (size_t)P);
  uint32_t rw[2] = {0u, 0u};                 // {num_rendered, prefilter violation}
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  GM_HIP(hipMemcpyAsync(rw, g.counters + GM_CNT_RENDERED, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  GM_HIP(hipStreamSynchronize(st));         // the one host sync of a forward (reference: rasterizer_impl.cu:411)
  const uint32_t r = rw[0];
  if (rw[1]) { set_error("Point is filtered although prefiltered is set. This shouldn't happen!"); return GM_ERR_INVALID_ARG; }   // auxiliary.h:157
  if (r > 0x7FFFFFFFu) { set_error("num_rendered overflows int32 (%u)", r); return GM_ERR_INVALID_ARG; }
  *num_rendered = (int)r;
  return GM_OK;
}

int gm_forward_1(void* geom_buffer, void* binning_buffer, void* image_buffer, int P, int D, int M, int num_rendered,
                 const float* background, int width, int height, const float* means3D, const float* shs,
                 const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color, int* radii,
                 int debug, void* stream) {
  FILL_ARGS(a, GM_POLICY_DEFAULT)
  if (int rc = check_raster_args(a)) return rc;
  (void)radii;
  if (num_rendered < 0) { set_error("negative num_rendered"); return GM_ERR_INVALID_ARG; }
  return gm_forward_1_geom(GM_POLICY_DEFAULT, geom_buffer, binning_buffer, image_buffer, P, num_rendered, 0, background, width, height, out_color,
                           debug, stream, nullptr, 0, nullptr);
}

static int backward_impl(int emission_policy, int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                         const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                         const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                         const float* campos, float tan_fovx, float tan_fovy, const int* radii, void* geom_buffer,
                         void* binning_buffer, void* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                         float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                         float* dL_dscale, float* dL_drot, int debug, void* stream, const ShAdamArgs* sh_adam) {
  if (int rc = check_policy(emission_policy)) return rc;
  const float* opacities = reinterpret_cast<const float*>(1);   // not used by backward; satisfies the shared check
  const float* cam_pos = campos;
  const int prefiltered = 0;
  FILL_ARGS(a, emission_policy)
  if (int rc = check_raster_args(a)) return rc;
  if (P == 0) return GM_OK;
  // intermediates a caller may decline: dL/dconic always, dL/dcolour when the colours come from SH rows, dL/dcov3D when the covariances
  // come from scale / rotation
  if (!geom_buffer || !image_buffer || !dL_dpix || !dL_dmean2D || !dL_dopacity || (!shs && !dL_dcolor) ||
      !dL_dmean3D || (!scales && !dL_dcov3D) || (shs && !dL_dsh && !sh_adam) || (scales && (!dL_dscale || !dL_drot))) {
    set_error("gm_backward: null buffer"); return GM_ERR_INVALID_ARG;
  }
  GeomState g = GeomState::from(geom_buffer, (size_t)P);
  ImageState img = ImageState::from(image_buffer, width, height);
  BinningState b = BinningState::from(binning_buffer, (size_t)(R > 0 ? R : 0));
  const int slot = sort_final_slot(TileGrid(width, height, a.tile_cull).ptiles);
  GM_HIP(hipMemsetAsync(g.grad_acc, 0, sizeof(float) * 12 * (size_t)P, a.stream));   // the only zero-fill of a backward
  if (R > 0) {
    if (!binning_buffer) { set_error("gm_backward: null binning buffer"); return GM_ERR_INVALID_ARG; }
    if (int rc = launch_render_bwd(g, b.pairs[slot], img, width, height, a.tile_cull, background, dL_dpix, debug, a.stream)) return rc;
  }
  return launch_preprocess_bwd(a, g, radii, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
                               dL_dscale, dL_drot, sh_adam);
}

int gm_backward_p(int emission_policy, int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                  const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                  const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                  const float* campos, float tan_fovx, float tan_fovy, const int* radii, void* geom_buffer,
                  void* binning_buffer, void* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                  float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                  float* dL_dscale, float* dL_drot, int debug, void* stream) {
  return backward_impl(emission_policy, P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier, rotations, cov3D_precomp,
                       viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_dmean2D, dL_dconic,
                       dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, stream, nullptr);
}

int gm_backward_sh_step(int emission_policy, int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                        float* shs, const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                        const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                        void* geom_buffer, void* binning_buffer, void* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dopacity,
                        float* dL_dmean3D, float* dL_dcov3D, float* dL_dscale, float* dL_drot, int rows, float* exp_avg, float* exp_avg_sq,
                        float lr_dc, float lr_rest, double beta1, double beta2, double eps, int step, int debug, void* stream) {
  if (!shs || M != 16 || D < 0 || D > 3) { set_error("gm_backward_sh_step: needs the [P,16,3] SH operand"); return GM_ERR_INVALID_ARG; }
  if (rows < 0 || rows > P || !exp_avg || !exp_avg_sq || step < 1) { set_error("gm_backward_sh_step: bad optimizer state (rows %d of %d, step %d)", rows, P, step); return GM_ERR_INVALID_ARG; }
  ShAdamArgs ad;
  ad.p = shs; ad.m = exp_avg; ad.v = exp_avg_sq; ad.rows = rows;
  ad.b1 = (float)beta1; ad.b2 = (float)beta2; ad.c1 = (float)(1.0 - beta1); ad.c2 = (float)(1.0 - beta2); ad.eps = (float)eps;
  const double corr = sqrt(1.0 - pow(beta2, (double)step)) / (1.0 - pow(beta1, (double)step));      // as gm_adam_step_active
  ad.step_lo = (float)(lr_dc * corr); ad.step_hi = (float)(lr_rest * corr);
  return backward_impl(emission_policy, P, D, M, R, background, width, height, means3D, shs, nullptr, scales, scale_modifier, rotations, cov3D_precomp,
                       viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer, dL_dpix, dL_dmean2D, nullptr,
                       dL_dopacity, nullptr, dL_dmean3D, dL_dcov3D, nullptr, dL_dscale, dL_drot, debug, stream, &ad);
}

int gm_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                const float* campos, float tan_fovx, float tan_fovy, const int* radii, void* geom_buffer,
                void* binning_buffer, void* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                float* dL_dscale, float* dL_drot, int debug, void* stream) {
  return gm_backward_p(GM_POLICY_DEFAULT, P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier, rotations,
                       cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer, image_buffer,
                       dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, debug, stream);
}

int gm_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                    void* stream) {
  (void)projmatrix;
  if (P < 0 || (P > 0 && (!means3D || !viewmatrix || !present))) { set_error("gm_mark_visible: bad args"); return GM_ERR_INVALID_ARG; }
  return launch_mark_visible(P, means3D, viewmatrix, present, reinterpret_cast<hipStream_t>(stream));
}

int gm_splat_floats(void) { return GM_SPLAT_STRIDE; }
void* gm_geom_field(void* geom_buffer, int P, const char* name) {
  GeomState g = GeomState::from(geom_buffer, (size_t)(P > 0 ? P : 1));
  if (!strcmp(name, "splat")) return g.splat;
  if (!strcmp(name, "depth_key")) return g.depth_key;
  if (!strcmp(name, "radii")) return g.radii;
  if (!strcmp(name, "tiles_touched")) return g.tiles_touched;
  if (!strcmp(name, "cov3D")) return g.cov3D;
  if (!strcmp(name, "clamped")) return g.clamped;
  if (!strcmp(name, "order")) return g.order;
  if (!strcmp(name, "bucket_start")) return g.bucket_start;
  if (!strcmp(name, "counters")) return g.counters;
  return nullptr;
}
void* gm_image_field(void* image_buffer, int W, int H, const char* name) {
  ImageState s = ImageState::from(image_buffer, W, H);
  if (!strcmp(name, "final_T")) return s.final_T;
  if (!strcmp(name, "n_contrib")) return s.n_contrib;
  if (!strcmp(name, "ranges")) return s.ranges;
  if (!strcmp(name, "tile_order")) return s.tile_order;
  return nullptr;
}
void* gm_binning_field(void* binning_buffer, int64_t R, int W, int H, int emission_policy, const char* name) {
  BinningState b = BinningState::from(binning_buffer, (size_t)(R > 0 ? R : 0));
  const int slot = sort_final_slot(TileGrid(W, H, emission_policy < 0 ? 0 : (emission_policy > 3 ? 3 : emission_policy)).ptiles);
  if (!strcmp(name, "pairs")) return b.pairs[slot];
  return nullptr;
}

size_t gm_knn_workspace_bytes(int P) { return knn_workspace_bytes(P); }
int gm_knn(int P, const float* points, float* meanDists, void* workspace, size_t workspace_bytes, void* stream) {
  if (P < 0 || (P > 0 && (!points || !meanDists || !workspace))) { set_error("gm_knn: bad args"); return GM_ERR_INVALID_ARG; }
  return launch_knn(P, points, meanDists, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

int gm_deform(int N, const int* tri, const float* w, const float* dV, const float* Rv, const float* Sv,
              const float* cov, const float* pos, float* pos_out, float* cov_out, float* rot_out, float* cov6_out,
              void* stream) {
  if (N < 0 || (N > 0 && (!tri || !w || !dV || !Rv || !Sv || !cov || !pos || !pos_out || !cov_out || !rot_out))) {
    set_error("gm_deform: bad args"); return GM_ERR_INVALID_ARG;
  }
  return launch_deform(N, tri, w, dV, Rv, Sv, cov, pos, pos_out, cov_out, rot_out, cov6_out, reinterpret_cast<hipStream_t>(stream));
}

int gm_sh_colors(int N, int deg, int M, const float* pos, const float* campos, const float* rot, const float* shs,
                 float* rgb, void* stream) {
  if (N < 0 || deg < 0 || deg > 3 || M < (deg + 1) * (deg + 1) || (N > 0 && (!pos || !campos || !shs || !rgb))) {
    set_error("gm_sh_colors: bad args"); return GM_ERR_INVALID_ARG;
  }
  return launch_sh_colors(N, deg, M, pos, campos, rot, shs, rgb, reinterpret_cast<hipStream_t>(stream));
}

int gm_deform_shade(int N, int deg, int M, const int* tri, const float* w, const float* dV, const float* Rv, const float* Sv,
                    const float* cov, const float* pos, const float* shs, const float* campos, float* pos_out, float* cov6_out,
                    float* rgb_out, float* cov_out, float* rot_out, void* stream) {
  if (N < 0 || deg < 0 || deg > 3 || M < (deg + 1) * (deg + 1) ||
      (N > 0 && (!tri || !w || !dV || !Rv || !Sv || !cov || !pos || !shs || !campos || !pos_out || !cov6_out || !rgb_out)) ||
      ((cov_out == nullptr) != (rot_out == nullptr))) {
    set_error("gm_deform_shade: bad args"); return GM_ERR_INVALID_ARG;
  }
  return launch_deform_shade(N, deg, M, tri, w, dV, Rv, Sv, cov, pos, shs, campos, pos_out, cov6_out, rgb_out, cov_out, rot_out,
                             reinterpret_cast<hipStream_t>(stream));
}

int gm_pack_mesh_state(int Vm, const float* state, const float* verts, float* packed, void* stream) {
  if (Vm < 0 || (Vm > 0 && (!state || !verts || !packed))) { set_error("gm_pack_mesh_state: bad args"); return GM_ERR_INVALID_ARG; }
  if (Vm > 0 && (reinterpret_cast<uintptr_t>(packed) & 15)) { set_error("gm_pack_mesh_state: packed must be 16-byte aligned"); return GM_ERR_INVALID_ARG; }
  return launch_pack_mesh_state(Vm, state, verts, packed, reinterpret_cast<hipStream_t>(stream));
}

int gm_deform_shade_packed(int N, int deg, int M, const int* tri, const float* w, const float* packed, const float* cov,
                           const float* pos, const float* shs, const float* campos, float* pos_out, float* cov6_out,
                           float* rgb_out, float* cov_out, float* rot_out, void* stream) {
  if (N < 0 || deg < 0 || deg > 3 || M < (deg + 1) * (deg + 1) ||
      (N > 0 && (!tri || !w || !packed || !cov || !pos || !shs || !campos || !pos_out || !cov6_out || !rgb_out)) ||
      ((cov_out == nullptr) != (rot_out == nullptr))) {
    set_error("gm_deform_shade_packed: bad args"); return GM_ERR_INVALID_ARG;
  }
  return launch_deform_shade_packed(N, deg, M, tri, w, packed, cov, pos, shs, campos, pos_out, cov6_out, rgb_out, cov_out, rot_out,
                                    reinterpret_cast<hipStream_t>(stream));
}

int gm_mesh_rs(int Vm, int nfaces, const float* V0, const float* V1, const int* faces, const int* adj_offsets, const int* adj_faces,
               float* R, float* S, float* state, void* stream) {
  if (Vm < 0 || nfaces < 0 || (Vm > 0 && (!V0 || !V1 || !adj_offsets || (nfaces > 0 && (!faces || !adj_faces)) || (!R && !S && !state)))) {
    set_error("gm_mesh_rs: bad args"); return GM_ERR_INVALID_ARG;
  }
  if ((R == nullptr) != (S == nullptr)) { set_error("gm_mesh_rs: pass R and S together"); return GM_ERR_INVALID_ARG; }
  return launch_mesh_rs(Vm, V0, V1, faces, adj_offsets, adj_faces, R, S, state, nullptr, reinterpret_cast<hipStream_t>(stream));
}

int gm_mesh_rs_packed(int Vm, int nfaces, const float* V0, const float* V1, const int* faces, const int* adj_offsets, const int* adj_faces,
                      float* packed, void* stream) {
  if (Vm < 0 || nfaces < 0 || (Vm > 0 && (!V0 || !V1 || !adj_offsets || !packed || (nfaces > 0 && (!faces || !adj_faces))))) {
    set_error("gm_mesh_rs_packed: bad args"); return GM_ERR_INVALID_ARG;
  }
  if (reinterpret_cast<uintptr_t>(packed) & 15) { set_error("gm_mesh_rs_packed: packed must be 16-byte aligned"); return GM_ERR_INVALID_ARG; }
  return launch_mesh_rs(Vm, V0, V1, faces, adj_offsets, adj_faces, nullptr, nullptr, nullptr, packed, reinterpret_cast<hipStream_t>(stream));
}

int gm_cov_to_scale_rot(int N, const float* cov, float* scales, float* rots, void* stream) {
  if (N < 0 || (N > 0 && (!cov || !scales || !rots))) { set_error("gm_cov_to_scale_rot: bad args"); return GM_ERR_INVALID_ARG; }
  if (N > 0 && (reinterpret_cast<uintptr_t>(rots) & 15)) { set_error("gm_cov_to_scale_rot: rots must be 16-byte aligned"); return GM_ERR_INVALID_ARG; }
  return launch_cov_to_scale_rot(N, cov, scales, rots, reinterpret_cast<hipStream_t>(stream));
}

int64_t gm_ssim_partials(int planes, int H, int W) {
  if (planes < 0 || H < 0 || W < 0) return 0;
  return (int64_t)planes * ((H + 31) / 32) * ((W + 31) / 32);
}

int gm_ssim_fwd(const float* img1, const float* img2, int planes, int H, int W, float* dS_dmu1, float* dS_dE11, float* dS_dE12,
                float* partial, void* stream) {
  if (planes < 0 || H < 0 || W < 0) { set_error("gm_ssim_fwd: negative size"); return GM_ERR_INVALID_ARG; }
  if (planes == 0 || H == 0 || W == 0) return GM_OK;
  if (!img1 || !img2 || !partial) { set_error("gm_ssim_fwd: null image or partial buffer"); return GM_ERR_INVALID_ARG; }
  const int nmaps = (dS_dmu1 != nullptr) + (dS_dE11 != nullptr) + (dS_dE12 != nullptr);
  if (nmaps != 0 && nmaps != 3) { set_error("gm_ssim_fwd: pass all three derivative maps or none"); return GM_ERR_INVALID_ARG; }
  if (planes > 65535) { set_error("gm_ssim_fwd: at most 65535 image planes per call"); return GM_ERR_INVALID_ARG; }
  return launch_ssim_fwd(img1, img2, planes, H, W, dS_dmu1, dS_dE11, dS_dE12, partial, reinterpret_cast<hipStream_t>(stream));
}

int gm_ssim_bwd(const float* img1, const float* img2, const float* dS_dmu1, const float* dS_dE11, const float* dS_dE12, int planes,
                int H, int W, const float* g_ssim, const float* g_l1, float* dL_dimg1, void* stream) {
  if (planes < 0 || H < 0 || W < 0) { set_error("gm_ssim_bwd: negative size"); return GM_ERR_INVALID_ARG; }
  if (planes == 0 || H == 0 || W == 0) return GM_OK;
  if (!img1 || !img2 || !dS_dmu1 || !dS_dE11 || !dS_dE12 || !g_ssim || !dL_dimg1) { set_error("gm_ssim_bwd: null argument"); return GM_ERR_INVALID_ARG; }
  if (planes > 65535) { set_error("gm_ssim_bwd: at most 65535 image planes per call"); return GM_ERR_INVALID_ARG; }
  return launch_ssim_bwd(img1, img2, dS_dmu1, dS_dE11, dS_dE12, planes, H, W, g_ssim, g_l1, dL_dimg1, reinterpret_cast<hipStream_t>(stream));
}

int gm_loss_combine(const float* partial, int64_t n_partials, double c_ssim, double c_l1, double offset, float* out, void* stream) {
  if (n_partials < 0) { set_error("gm_loss_combine: negative count"); return GM_ERR_INVALID_ARG; }
  if (!out || (n_partials > 0 && !partial)) { set_error("gm_loss_combine: null partial / out"); return GM_ERR_INVALID_ARG; }
  return launch_loss_combine(partial, (long long)n_partials, c_ssim, c_l1, offset, out, reinterpret_cast<hipStream_t>(stream));
}

static int fill_act(ActArgs& a, int N, float alpha, const float* bc, const float* dist, const float* scaling, const float* rotation,
                    const float* opacity, const float* v1, const float* v2, const float* v3, const float* normal, const float* r) {
  if (N < 0) { set_error("gm_mesh_activate: negative N"); return GM_ERR_INVALID_ARG; }
  if (N > 0 && (!bc || !dist || !scaling || !rotation || !opacity || !v1 || !v2 || !v3 || !normal || !r)) {
    set_error("gm_mesh_activate: null input"); return GM_ERR_INVALID_ARG;
  }
  if (N > 0 && (reinterpret_cast<uintptr_t>(rotation) & 15)) { set_error("gm_mesh_activate: rotation must be 16-byte aligned"); return GM_ERR_INVALID_ARG; }
  a.N = N; a.alpha = alpha; a.bc = bc; a.dist = dist; a.scaling = scaling; a.rotation = rotation; a.opacity = opacity;
  a.v1 = v1; a.v2 = v2; a.v3 = v3; a.normal = normal; a.r = r;
  return GM_OK;
}

int gm_mesh_activate_fwd(int N, float alpha, const float* bc, const float* dist, const float* scaling, const float* rotation,
                         const float* opacity, const float* v1, const float* v2, const float* v3, const float* normal, const float* r,
                         float* xyz, float* scales, float* rots, float* opac, float mr_weight, float* mr_partial, void* stream) {
  ActArgs a;
  if (int rc = fill_act(a, N, alpha, bc, dist, scaling, rotation, opacity, v1, v2, v3, normal, r)) return rc;
  if (N > 0 && (!xyz || !scales || !rots || !opac || (reinterpret_cast<uintptr_t>(rots) & 15))) {
    set_error("gm_mesh_activate_fwd: null output or unaligned rots"); return GM_ERR_INVALID_ARG;
  }
  return launch_mesh_activate_fwd(a, xyz, scales, rots, opac, mr_weight, mr_partial, reinterpret_cast<hipStream_t>(stream));
}

int gm_mesh_activate_bwd(int N, float alpha, const float* bc, const float* dist, const float* scaling, const float* rotation,
                         const float* opacity, const float* v1, const float* v2, const float* v3, const float* normal, const float* r,
                         const float* d_xyz, const float* d_scales, const float* d_rots, const float* d_opac, float* d_bc, float* d_dist,
                         float* d_scaling, float* d_rotation, float* d_opacity, float mr_weight, const float* d_mr, void* stream) {
  ActArgs a;
  if (int rc = fill_act(a, N, alpha, bc, dist, scaling, rotation, opacity, v1, v2, v3, normal, r)) return rc;
  if (N > 0 && (!d_bc || !d_dist || !d_scaling || !d_rotation || !d_opacity || (reinterpret_cast<uintptr_t>(d_rotation) & 15) ||
                (d_rots && (reinterpret_cast<uintptr_t>(d_rots) & 15)))) {
    set_error("gm_mesh_activate_bwd: null output or unaligned rotation gradient"); return GM_ERR_INVALID_ARG;
  }
  return launch_mesh_activate_bwd(a, d_xyz, d_scales, d_rots, d_opac, d_bc, d_dist, d_scaling, d_rotation, d_opacity, mr_weight, d_mr,
                                  reinterpret_cast<hipStream_t>(stream));
}

int gm_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 const uint64_t* sizes, const float* lr, const float* lr_rest, const uint32_t* period, const uint32_t* split,
                 double beta1, double beta2, double eps, int step, void* stream) {
  return gm_adam_step_active(count, params, grads, exp_avg, exp_avg_sq, sizes, lr, lr_rest, period, split, nullptr, beta1, beta2, eps, step, stream);
}

int gm_adam_step_active(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                        const uint64_t* sizes, const float* lr, const float* lr_rest, const uint32_t* period, const uint32_t* split,
                        const uint32_t* active, double beta1, double beta2, double eps, int step, void* stream) {
  if (count < 0 || count > 8 || step < 1) { set_error("gm_adam_step: 0..8 tensors per call, step >= 1"); return GM_ERR_INVALID_ARG; }
  if (count > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !sizes || !lr)) { set_error("gm_adam_step: null table"); return GM_ERR_INVALID_ARG; }
  AdamTable tab;
  tab.count = 0; tab.b1 = (float)beta1; tab.b2 = (float)beta2; tab.eps = (float)eps;
  tab.omb1 = (float)(1.0 - beta1); tab.omb2 = (float)(1.0 - beta2);
  const double corr = sqrt(1.0 - pow(beta2, (double)step)) / (1.0 - pow(beta1, (double)step));
  for (int i = 0; i < count; i++) {
    if (sizes[i] == 0) continue;
    if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]) { set_error("gm_adam_step: null tensor %d", i); return GM_ERR_INVALID_ARG; }
    if ((reinterpret_cast<uintptr_t>(params[i]) | reinterpret_cast<uintptr_t>(grads[i]) | reinterpret_cast<uintptr_t>(exp_avg[i]) |
         reinterpret_cast<uintptr_t>(exp_avg_sq[i])) & 15) { set_error("gm_adam_step: tensor %d is not 16-byte aligned", i); return GM_ERR_INVALID_ARG; }
    AdamTensor& t = tab.t[tab.count++];
    t.p = params[i]; t.g = grads[i]; t.m = exp_avg[i]; t.v = exp_avg_sq[i]; t.n = sizes[i];
    t.step_lo = (float)(lr[i] * corr);
    t.period = period ? period[i] : 0u; t.split = split ? split[i] : 0u;
    if (t.period & 3u) { set_error("gm_adam_step: period must be a multiple of 4"); return GM_ERR_INVALID_ARG; }
    t.active = (active && active[i] && t.period && active[i] < t.period) ? active[i] : 0u;
    if (t.active && (t.n % t.period) != 0) { set_error("gm_adam_step_active: tensor %d is not a whole number of periods", i); return GM_ERR_INVALID_ARG; }
    t.step_hi = (float)((lr_rest && t.period) ? lr_rest[i] * corr : lr[i] * corr);
  }
  return launch_adam(tab, reinterpret_cast<hipStream_t>(stream));
}

int gm_densify_stats(int N, const int* radii, const float* viewspace_grad, float* max_radii2D, float* grad_accum, float* denom, void* stream) {
  if (N < 0 || (N > 0 && (!radii || !viewspace_grad || !max_radii2D || !grad_accum || !denom))) {
    set_error("gm_densify_stats: bad args"); return GM_ERR_INVALID_ARG;
  }
  return launch_densify_stats(N, radii, viewspace_grad, max_radii2D, grad_accum, denom, reinterpret_cast<hipStream_t>(stream));
}

void gm_profile_enable(int on) { g_prof_on = on != 0; }
void gm_profile_reset(void) {
  drain_profile();
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < ST_COUNT; i++) { g_ms[i] = 0; g_n[i] = 0; }
}
int gm_profile_read(const char* stage, double* total_ms, int64_t* launches) {
  drain_profile();
  for (int i = 0; i < ST_COUNT; i++)
    if (!strcmp(stage, kStageNames[i])) {
      if (total_ms) *total_ms = g_ms[i];
      if (launches) *launches = g_n[i];
      return GM_OK;
    }
  set_error("unknown stage '%s'", stage);
  return GM_ERR_INVALID_ARG;
}

}  // extern "C"
