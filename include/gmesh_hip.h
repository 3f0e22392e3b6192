/*
 * gmesh_hip.h -- C ABI of libgmesh_hip.so: the MI355X (gfx950) implementation of the GaussianMesh
 * hot path.  Plain pointers and sizes only; every device pointer is a HIP device address, every
 * call is stream-ordered on `stream` (a hipStream_t passed as void*; NULL = the null stream).
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * repo root; RAST = gaussian_renderer/diff_gaussian_rasterizater/cuda_rasterizer).  The reference
 * exposes these as C++ static members called from Jittor `jt.code` JIT stubs
 * (rasterize_points.py:123-190, 197-269, 311-396); INTEGRATION.md shows the stub a maintainer
 * would write against this header instead.
 *
 * Conventions shared by all calls
 *   - return value: 0 = GM_OK, otherwise a GM_ERR_* code; gm_last_error() gives a thread-local
 *     human-readable message (replaces the C++ exceptions of RAST/rasterizer_impl.cu:372-375 and
 *     the CHECK_CUDA macro, RAST/auxiliary.h:165-172).
 *   - `debug` != 0: synchronise and check for errors after every kernel launch (CHECK_CUDA semantics).
 *   - ownership: the caller owns every buffer; the library never allocates device memory
 *     (as in the reference, where python allocates geomBuffer/binningBuffer/imgBuffer,
 *     rasterize_points.py:118-121, 192-194).  Scratch buffers are opaque; their layout depends only
 *     on (base address mod 256, P/R/W/H) so the SAME buffers at the SAME addresses must be passed to
 *     gm_forward_1 and gm_backward after gm_forward_0 (reference: rasterizer_impl.cu:546-548).
 *   - nullable inputs: shs | colors_precomp (exactly one), (scales,rotations) | cov3D_precomp
 *     (exactly one), radii (optional output).  NULL means "feature off" for all of them.
 *   - all float data is IEEE binary32, contiguous, in the layouts of the reference python boundary:
 *     means3D [P,3], shs [P,M,3], colors_precomp [P,3], opacities [P], scales [P,3],
 *     rotations [P,4] (r,x,y,z), cov3D_precomp [P,6] (xx,xy,xz,yy,yz,zz),
 *     viewmatrix/projmatrix 16 floats (the transposed 4x4 of scene/cameras.py:47-49),
 *     cam_pos 3 floats, background 3 floats, out_color planar [3,H,W].
 */
#ifndef GMESH_HIP_H_INCLUDED
#define GMESH_HIP_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GM_OK 0
#define GM_ERR_INVALID_ARG 1   /* bad size / null where not allowed / misuse */
#define GM_ERR_HIP 2           /* a HIP runtime call or kernel failed */
#define GM_ERR_BUFFER 3        /* a caller-provided buffer is too small */

/* ABI version of this header; bumped on any signature change. */
#define GM_ABI_VERSION 3
int gm_abi_version(void);
const char* gm_last_error(void);

/* Instance emission policy: an ARGUMENT of the extended entry points (gm_forward_0_async, gm_forward_0_deformed_async,
 * gm_forward_1_geom, gm_backward_p, gm_binning_field); the entry points with the reference's signatures (gm_forward_0,
 * gm_forward_1, gm_backward) use GM_POLICY_DEFAULT.  The library keeps no policy state: calls on distinct buffers are
 * independent and thread-safe.  The halves of one pass must be given the same policy; it is latched in the geometry
 * buffer by the first half, and a second half under another policy emits nothing and renders the background.
 * out_color, radii and all gradients are the same under every policy; num_rendered and the internal lists differ.
 *   0: emit every tile of the bounding rectangle exactly as the reference does (RAST/rasterizer_impl.cu:98-109);
 *      num_rendered, point_list and ranges are then identical to the reference's.
 *   1: a (Gaussian, tile) instance is emitted only if the Gaussian can reach alpha >= 1/255 somewhere in the 16x16
 *      tile (exact conservative test).  Dropped instances are skipped by every pixel of the tile in the reference too
 *      (RAST/forward.cu:344).
 *   2 (default), 3: the same test, but instances are (Gaussian, PARENT tile) pairs for parents of 2x2 / 4x4 tiles; the
 *      key of an instance carries the mask of the parent's child tiles the Gaussian reaches (bits 16+), and the 16x16
 *      blend workgroups walk their parent's list.  The instance stream (and the sort over it) shrinks ~2x / ~3x. */
#define GM_POLICY_REFERENCE 0
#define GM_POLICY_DEFAULT 2

/* Scratch sizes.  Replace CudaRasterizer::required<GeometryState|ImageState|BinningState>(n)
 * (RAST/rasterizer_impl.h:67-73; python side rasterize_points.py:63-86). */
size_t gm_geom_bytes(int P);
size_t gm_image_bytes(int W, int H);
size_t gm_work_hint_bytes(int W, int H);      /* gm_forward_1_geom's optional work_hint buffer */
size_t gm_binning_bytes(int64_t num_rendered);

/* Replaces CudaRasterizer::Rasterizer::forward_0 (RAST/rasterizer.h:31-51, rasterizer_impl.cu:338-413):
 * per-Gaussian preprocess (cull, cov3D, EWA cov2D, conic, radius, tile rect, SH->RGB) and the
 * count of (Gaussian, tile) instances.  Performs the ONE host synchronisation of a forward pass
 * (reference: cudaMemcpy D2H at rasterizer_impl.cu:411) and stores the count in *num_rendered.
 * radii (int32 [P], may be NULL) receives the screen radius (0 = culled). */
int gm_forward_0(void* geom_buffer, int P, int D, int M, const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                 const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                 const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                 float tan_fovy, int prefiltered, int* radii, int debug, void* stream, int* num_rendered);

/* gm_forward_0 without its host synchronisation, with the emission policy as an argument: everything is enqueued on
 * `stream`, including a copy of the instance count into *num_rendered_host (page-locked host memory; may be NULL), issued as
 * soon as the count is known (before the depth ordering finishes).  count_event (a hipEvent_t, may be NULL) is recorded
 * right behind that copy: the caller keeps feeding the GPU (e.g. the next frame's gm_forward_0_async on another stream)
 * and completes the frame with gm_forward_1_geom once the event has fired.  This hides the one host round trip of the
 * reference design; gm_forward_1_geom's sync-free mode removes it. */
int gm_forward_0_async(int emission_policy, void* geom_buffer, int P, int D, int M, const float* background, int width, int height,
                       const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                       const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                       float tan_fovy, int prefiltered, int* radii, int debug, void* stream, int* num_rendered_host,
                       void* count_event);

/* Replaces CudaRasterizer::Rasterizer::forward_1 (RAST/rasterizer.h:53-76, rasterizer_impl.cu:416-511):
 * instance emission, (tile, depth) ordering, tile ranges, front-to-back alpha blend.
 * binning_buffer must hold gm_binning_bytes(num_rendered), image_buffer gm_image_bytes(W,H). */
int gm_forward_1(void* geom_buffer, void* binning_buffer, void* image_buffer, int P, int D, int M, int num_rendered,
                 const float* background, int width, int height, const float* means3D, const float* shs,
                 const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                 const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float* out_color, int* radii,
                 int debug, void* stream);

/* Replaces CudaRasterizer::Rasterizer::backward (RAST/rasterizer.h:103-132, rasterizer_impl.cu:515-609).
 * Gradient outputs: dL_dmean2D [P,3] (x,y used), dL_dconic [P,4] (slots 0,1,3 used), dL_dopacity [P],
 * dL_dcolor [P,3], dL_dmean3D [P,3], dL_dcov3D [P,6], dL_dsh [P,M,3], dL_dscale [P,3], dL_drot [P,4].
 * Unlike the reference (which needs them zero-filled by the caller, rasterize_points.py:302-310) the
 * library zeroes every gradient output itself, so on return they hold exactly this pass's gradients.
 * dL_dsh may be NULL when shs is NULL; dL_dscale/dL_drot may be NULL when scales is NULL.  The intermediates of the chain rule may be
 * declined with NULL (they are then not written: 52 of ~600 bytes per Gaussian): dL_dconic always, dL_dcolor when shs is given,
 * dL_dcov3D when scales is given. */
int gm_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                const float* campos, float tan_fovx, float tan_fovy, const int* radii, void* geom_buffer,
                void* binning_buffer, void* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                float* dL_dscale, float* dL_drot, int debug, void* stream);

/* gm_backward for lists built under an explicit emission policy (the one given to the forward halves). */
int gm_backward_p(int emission_policy, int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                  const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                  const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                  const float* campos, float tan_fovx, float tan_fovy, const int* radii, void* geom_buffer,
                  void* binning_buffer, void* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                  float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh,
                  float* dL_dscale, float* dL_drot, int debug, void* stream);

/* gm_backward_p for a TRAINING step whose SH rows are the optimizer's parameter (scene/mesh_based_gaussian_model.py:242-263: the "f_dc" and
 * "f_rest" groups of training_setup; jittor.nn.Adam): the Adam step of those rows is applied inside the backward pass instead of by
 * gm_adam_step afterwards.  dL/dSH of a Gaussian is produced whole by the thread that owns it (RAST/backward.cu:20-139), so the 192-byte
 * gradient row never has to travel: per trainable Gaussian 192 B of dL/dSH written + read and 192 B of parameter read disappear (at SH
 * degree 3, where 27 000 of the reference's 30 000 iterations run, the SH group is 48 of a Gaussian's 59 parameters).
 *   shs [P,16,3]: updated IN PLACE for rows [0, rows) (rows behind them - a frozen cloud sharing the operand - are read only);
 *   exp_avg / exp_avg_sq [rows,16,3]: Adam's moments; lr_dc steps coefficient 0, lr_rest the others; step >= 1 = the step being taken;
 *   the update is m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2; p -= lr sqrt(1 - b2^t) / (1 - b1^t) m / (sqrt(v) + eps), element for
 *   element what gm_adam_step computes from the same gradient (culled Gaussians: g = 0, their moments decay and p moves, as in the reference).
 * dL/dSH, dL/dcolour and dL/dconic are not produced.  The backward of a REFUSED forward (sync-free capacity overflow: zero gradients, the
 * caller repeats the iteration) leaves parameter and moments untouched.  colors_precomp input is not supported (nothing to step). */
int gm_backward_sh_step(int emission_policy, int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                        float* shs, const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                        const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                        void* geom_buffer, void* binning_buffer, void* image_buffer, const float* dL_dpix, float* dL_dmean2D, float* dL_dopacity,
                        float* dL_dmean3D, float* dL_dcov3D, float* dL_dscale, float* dL_drot, int rows, float* exp_avg, float* exp_avg_sq,
                        float lr_dc, float lr_rest, double beta1, double beta2, double eps, int step, int debug, void* stream);

/* Replaces CudaRasterizer::Rasterizer::markVisible (RAST/rasterizer.h:24-29, rasterizer_impl.cu:141-153).
 * present: uint8 [P], 1 if view-space z > 0.2. */
int gm_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                    void* stream);

/* Read-only views into the opaque scratch buffers (tests / debugging; the reference exposes the
 * same data as the GeometryState/ImageState/BinningState structs, RAST/rasterizer_impl.h:29-65).
 * Each returns a device pointer inside the given buffer, or NULL for an unknown name.
 *   geom:    "splat" float[P][gm_splat_floats()] = {x,y,conic.x,conic.y | conic.z,opacity,r,g | b} (9 floats: the record the blend
 *            kernels gather), "depth_key" uint32[P] (float bits of the view-space depth, 0xFFFFFFFF for a culled Gaussian; not
 *            written by the direct depth placement),
 *            "radii" int32[P] (internal copy: filled only by a forward that was given NO radii array; gm_backward reads it when it
 *            is given none either - pass the forward's array otherwise), "tiles_touched" uint32[P] (gm_forward_0_deformed_async
 *            fills it only for rectangles of 65535 instances or more: the count rides in the emission record), "cov3D" float[P][6],
 *            "clamped" uint8[P] (bit ch set = channel ch clamped), "order" uint32[V] (ids of the V visible Gaussians
 *            in (depth, id) order; V = "bucket_start"[2048]), "bucket_start" uint32[2049]
 *   image:   "final_T" float[H*W], "n_contrib" uint32[H*W], "ranges" uint32[T][2], "tile_order" uint32[T] (the forward blend's
 *            dispatch order: a permutation of the list tiles)
 *   binning: "pairs" uint32[R][2] = (list tile id | child mask << 16, Gaussian id) per instance, sorted by tile then
 *            (depth, id): column 1 is the reference's point_list, column 0 its sorted tile keys (child mask: policy 0/1
 *            = 1; policy 2 = the 4 x 4 8-pixel quadrants of the 32-px list tile, bit 4 qy + qx; policy 3 = the 4 x 4
 *            16-px tiles of the 64-px list tile, bit 4 ty + tx) */
int gm_splat_floats(void);
void* gm_geom_field(void* geom_buffer, int P, const char* name);
void* gm_image_field(void* image_buffer, int W, int H, const char* name);
void* gm_binning_field(void* binning_buffer, int64_t R, int W, int H, int emission_policy, const char* name);

/* Replaces SimpleKNN::knn (scene/simple_knn/cuda_headers/simple_knn.h:18, simple_knn.cu:185-221):
 * meanDists[i] = mean of the 3 smallest squared distances from point i to the other points.
 * The reference allocates its temporaries internally (cudaMalloc/thrust); here the caller passes a
 * workspace of gm_knn_workspace_bytes(P).  Performs one host synchronisation (bounding box readback,
 * as simple_knn.cu:197,200). */
size_t gm_knn_workspace_bytes(int P);
int gm_knn(int P, const float* points, float* meanDists, void* workspace, size_t workspace_bytes, void* stream);

/* Mesh-driven deformation of bound Gaussians; replaces the Jittor tensor algebra of
 * SingleObjectDeform.deform_gaussian (edittool/__init__.py:116-131), tensor-in form:
 *   tri int32 [N,3] vertex ids of the bound face, w float [N,3] barycentric weights,
 *   dV float [Vm,3] = V_deformed - V_rest, Rv / Sv float [Vm,3,3] per-vertex rotation / shear
 *   (pyACAP GetRS output), cov float [N,3,3] rest covariance, pos float [N,3] rest position.
 * Outputs: pos_out [N,3], cov_out [N,3,3] (= RS cov RS^T, RS = Rb^T Sb), rot_out [N,3,3] (= Rb^T).
 * cov6_out (may be NULL): float [N,6] strip_symmetric(cov_out) ready for cov3D_precomp
 * (edittool/general_utils.py:26-37). */
int gm_deform(int N, const int* tri, const float* w, const float* dV, const float* Rv, const float* Sv,
              const float* cov, const float* pos, float* pos_out, float* cov_out, float* rot_out, float* cov6_out,
              void* stream);

/* View-dependent colour of deformed Gaussians; replaces edittool/__init__.py:442-448:
 *   dir = normalize(pos - campos); dir_rot = rot^T dir; rgb = max(SH_deg(dir_rot) + 0.5, 0).
 * rot may be NULL (identity: the train-time convert_SHs_python path, gaussian_renderer/__init__.py:84-89). */
int gm_sh_colors(int N, int deg, int M, const float* pos, const float* campos, const float* rot, const float* shs,
                 float* rgb, void* stream);

/* Fused edit-loop step: gm_deform followed by gm_sh_colors(rot = the deformed rotation) in ONE pass over the cloud
 * (what ObjectVisualTool.render_gaussian consumes per frame, edittool/__init__.py:421-472): pos_out [N,3],
 * cov6_out [N,6] (cov3D_precomp), rgb_out [N,3] (colors_precomp).  cov_out / rot_out ([N,3,3] each) are optional
 * (both or neither) for callers that also want SingleObjectDeform's gaussian_deform_cov / gaussian_deform_rot. */
int gm_deform_shade(int N, int deg, int M, const int* tri, const float* w, const float* dV, const float* Rv, const float* Sv,
                    const float* cov, const float* pos, const float* shs, const float* campos, float* pos_out, float* cov6_out,
                    float* rgb_out, float* cov_out, float* rot_out, void* stream);

/* The same step for a render loop that receives the mesh state of a frame as ONE [Vm][21] array (V1 | R | S per vertex,
 * what rank 0 broadcasts per deformation frame): gm_pack_mesh_state subtracts the rest pose verts [Vm,3] and writes the
 * gather table packed (float [Vm][24], 16-byte aligned: {dV,0} {R0..3} {R4..7} {R8,S0,S1,S2} {S3..6} {S7,S8,0,0});
 * gm_deform_shade_packed is gm_deform_shade reading that table (18 16-byte gathers per Gaussian instead of 63 4-byte
 * ones).  Needs M == 16 and 16-byte aligned shs / cov / outputs. */
int gm_pack_mesh_state(int Vm, const float* state, const float* verts, float* packed, void* stream);
int gm_deform_shade_packed(int N, int deg, int M, const int* tri, const float* w, const float* packed, const float* cov,
                           const float* pos, const float* shs, const float* campos, float* pos_out, float* cov6_out,
                           float* rgb_out, float* cov_out, float* rot_out, void* stream);

/* Edit-loop fast path (ObjectVisualTool.render_gaussian, edittool/__init__.py:421-472, one frame): gm_deform_shade_packed
 * and the first half of the forward in ONE pass over the cloud - the deformed position / covariance / colour of a
 * Gaussian go straight into its projection, conic, radius and instance count without a round trip through HBM.
 * Equivalent, bit for bit, to gm_deform_shade_packed followed by gm_forward_0_async(colors_precomp = rgb_out,
 * cov3D_precomp = cov6_out, means3D = pos_out, scale_modifier 1).  pos_out / cov6_out / rgb_out: all three or all NULL.
 * Complete the frame with gm_forward_1_geom.
 * A deformed frame is FORWARD-ONLY: its geometry buffer does not hold everything gm_backward reads.  Not written by this pass (the
 * buffer keeps whatever an earlier call left there): "clamped" (the SH clamp flags), "tiles_touched" except for saturated rectangles,
 * the internal radii copy when `radii` is given, and - in the stream variant with a plan - "bin" and "depth_key".  Do not call
 * gm_backward / gm_backward_p on such a buffer; differentiate through gm_deform_shade_packed + gm_forward_0_async instead. */
int gm_forward_0_deformed_async(int emission_policy, void* geom_buffer, int P, int deg, int M, int width, int height, const int* tri,
                                const float* w, const float* packed, const float* cov, const float* pos, const float* shs,
                                const float* opacities, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                float tan_fovx, float tan_fovy, float* pos_out, float* cov6_out, float* rgb_out, int* radii, int debug,
                                void* stream, int* num_rendered_host, void* count_event);

/* The same first half for the frames of ONE VIEW STREAM (consecutive cameras of an orbit, an edit session: the edit tool's render
 * loop).  depth_plan (gm_depth_plan_bytes() bytes, zeroed once by the caller, shared by the stream's frames - also by frames in
 * flight on different HIP streams) carries depth-bucket tables from frame to frame.
 *   flags & GM_STREAM_DIRECT: the pass itself appends every visible Gaussian to its depth bucket, looked up in the newest table earlier frames
 *     left in the plan (depth_slab: gm_depth_slab_bytes(P) bytes of scratch PER FRAME IN FLIGHT, like geom_buffer); the depth
 *     partition's three launches and the gather of the emission records disappear.  The (depth, id) order does not depend on the table,
 *     only the balance of the buckets does.  A frame the direct placement cannot order - no table yet, a table too stale for this view
 *     (a bucket above its slab), piles of equal depths - is REFUSED: its second half renders the background and status word 3 reads 2
 *     (gm_forward_1_geom's status_host / gm_forward_status_async; num_rendered_host is the correct total either way); begin it again
 *     without the flag.
 *   without GM_STREAM_DIRECT: the partition path of gm_forward_0_deformed_async, which also leaves its table in the plan (depth_slab unused,
 *     may be NULL): how a stream's first frame and refused frames are rendered.
 * Images, radii and lists are those of gm_forward_0_deformed_async, bit for bit. */
size_t gm_depth_plan_bytes(void);
size_t gm_depth_slab_bytes(int P);
int gm_forward_0_deformed_stream_async(int emission_policy, void* geom_buffer, int P, int deg, int M, int width, int height, const int* tri,
                                       const float* w, const float* packed, const float* cov, const float* pos, const float* shs,
                                       const float* opacities, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                                       float tan_fovx, float tan_fovy, float* pos_out, float* cov6_out, float* rgb_out, int* radii,
                                       int debug, void* stream, int* num_rendered_host, void* count_event, void* depth_slab,
                                       unsigned int* depth_plan, int flags);
#define GM_STREAM_DIRECT 1   /* direct depth placement, see above */
#define GM_STREAM_COV6 2     /* cov holds the rest covariances as [N][6] rows xx xy xz yy yz zz instead of [N][9]: for clouds whose 3x3
                              * matrices are symmetric BIT FOR BIT (the six mirrored reads then return the same floats and every result
                              * is the one of the [N][9] call); 12 of the pass's 340 bytes per Gaussian less */

/* Second half of a forward without the per-Gaussian input pointers gm_forward_1 does not read: instance emission, tile
 * sort, tile ranges, blend.
 *   num_rendered >= 0: the instance count the first half reported; binning_buffer holds gm_binning_bytes(num_rendered);
 *                      binning_capacity is ignored.
 *   num_rendered <  0: SYNC-FREE mode - the host never learns the count.  binning_buffer holds
 *                      gm_binning_bytes(binning_capacity); the kernels read the count on the device.  If it exceeds the
 *                      capacity nothing is emitted, the image is the background and the overflow is reported by
 *                      gm_forward_status_async (grow the buffer and render the frame again).
 * status_host (page-locked, device-accessible host memory, 4 x int32; may be NULL) receives the frame's status words
 * {num_rendered, prefilter violation, policy, refused}, written by the blend kernel itself (no copy launch behind the frame); they are valid once
 * the stream has passed this call.  refused != 0: nothing was emitted (capacity overflow or policy mismatch).
 * gm_forward_status_async copies the same four words of the forward that last used geom_buffer, stream-ordered.
 * flags: 0, or GM_FWD_IMAGE_ONLY for a frame no backward pass will follow (the edit / viewer loop): the blend writes out_color
 * only - the per-pixel final transmittance and contributor count in image_buffer (forward.cu:369-370, read by
 * backward.cu:444-447 alone) are left untouched, so gm_backward on that image_buffer is undefined.
 * work_hint (may be NULL): device buffer of gm_work_hint_bytes(width, height) bytes, zeroed once by the caller and then handed
 * to the consecutive frames of one view stream (an orbit, an edit session at one resolution and policy).  The blend leaves in
 * it what each list tile cost; the next frames dispatch the tiles that were expensive first instead of the ones with the
 * longest lists, which shortens the kernel's tail (list length says little: a long list under an opaque surface saturates
 * early, a short one on a silhouette is walked to its end).  It only moves work in time: images are bit-identical with and
 * without it, frames in flight on several streams may share one buffer. */
#define GM_FWD_IMAGE_ONLY 1
/* GM_FWD_EXACT_EXPONENT: a forward that a BACKWARD pass follows (not with GM_FWD_IMAGE_ONLY).  The blend then evaluates every
 * exponent per pixel, with the expression gm_backward_p evaluates, instead of the matrix-core polynomial (absolute error ~1e-5 in
 * the exponent): both halves of the training step take the SAME alpha >= 1/255 decision for every (entry, pixel), as the
 * reference's do, whose backward.cu repeats forward.cu's expression (forward.cu:330-352, backward.cu:481-497).  Costs 17 us
 * of a 100-us blend at 1 M Gaussians / 1080p; the image differs from the default's by <= 3e-5 outside decision thresholds. */
#define GM_FWD_EXACT_EXPONENT 2
int gm_forward_1_geom(int emission_policy, void* geom_buffer, void* binning_buffer, void* image_buffer, int P, int num_rendered,
                      int64_t binning_capacity, const float* background, int width, int height, float* out_color, int debug, void* stream,
                      int* status_host, int flags, unsigned int* work_hint);
int gm_forward_status_async(void* geom_buffer, int P, int* status_host, void* stream);

/* K frames of ONE view stream per launch chain (the edit tool replaying a deformation sequence, a camera path: the frames the render
 * loop would otherwise keep in flight on K HIP streams).  Equivalent, frame by frame and bit for bit (radii, lists, image, status words), to
 *   gm_forward_0_deformed_stream_async(policy, frame.geom_buffer, ..., frame.packed, ..., frame.viewmatrix, ..., flags & GM_BATCH_COV6)
 *   gm_forward_1_geom(policy, frame.geom_buffer, frame.binning_buffer, frame.image_buffer, P, -1, binning_capacity, background, ...,
 *                     frame.out_color, debug, stream, frame.status_host, flags & GM_BATCH_IMAGE_ONLY, work_hint)
 * for each of the K frames, but
 *   - the static cloud (face ids, weights, rest covariance and position, SH rows, opacity: 256 of the 321 bytes per Gaussian the fused pass
 *     of a frame moves) is read from HBM ONCE for the batch: one pass loops over the frames' (gather table, camera) pairs
 *     (edittool/__init__.py:103-131, 421-472: what one frame consumes; RAST/forward.cu:155-256), and
 *   - every later stage is ONE launch over the K frames (grid z = frame): a batch is 12 launches instead of 12 K, each K times as large.
 * Sync-free second half only (the instance counts stay on the device; binning_capacity instances per frame; a frame that outgrows it is
 * refused in its own status words and rendered again by the caller through the single-frame calls).  1 <= K <= GM_BATCH_MAX; M == 16;
 * emission policies with at most 2048 list tiles (the one-pass tile sort); every scratch buffer base 256-byte aligned; the frames'
 * buffers distinct.  Like every deformed frame: forward only. */
#define GM_BATCH_MAX 8
#define GM_BATCH_IMAGE_ONLY 1    /* as GM_FWD_IMAGE_ONLY */
#define GM_BATCH_COV6 2          /* as GM_STREAM_COV6 */
typedef struct gm_batch_frame {
  const float* packed;           /* [Vm][24] gather table of this frame (gm_mesh_rs_packed / gm_mesh_rs_packed_batch / gm_pack_mesh_state) */
  const float* viewmatrix; const float* projmatrix; const float* cam_pos;
  float tan_fovx, tan_fovy;
  void* geom_buffer; void* binning_buffer; void* image_buffer;      /* gm_geom_bytes(P) / gm_binning_bytes(binning_capacity) / gm_image_bytes(W, H) */
  float* out_color;              /* [3,H,W] */
  int* radii;                    /* [P], may be NULL */
  int* status_host;              /* 4 x int32, page-locked, may be NULL */
} gm_batch_frame;
int gm_forward_deformed_batch_async(int emission_policy, int K, const gm_batch_frame* frames, int P, int deg, int M, int width, int height,
                                    const int* tri, const float* w, const float* cov, const float* pos, const float* shs, const float* opacities,
                                    const float* background, int64_t binning_capacity, int flags, unsigned int* work_hint, int debug, void* stream);
/* gm_mesh_rs_packed for the K deformed meshes of such a batch in one launch: V1[k] -> packed[k] (host arrays of K device pointers). */
int gm_mesh_rs_packed_batch(int K, int Vm, int nfaces, const float* V0, const float* const* V1, const int* faces, const int* adj_offsets,
                            const int* adj_faces, float* const* packed, void* stream);

/* Per-vertex rotation / stretch of a deformed proxy mesh: replaces pyACAP.GetRS(rest vertices, deformed vertices, ...) at
 * edittool/__init__.py:102, 109 (pyACAP is a binary missing from the reference tree, so the contract is the one its call
 * site implies).  V0 / V1 float [Vm,3] rest / deformed vertices, faces int32 [nfaces,3], adj_offsets int32 [Vm+1] and
 * adj_faces int32 [3 nfaces]: CSR list of the faces incident to each vertex (built once per mesh by the caller).
 * Per vertex the affine map of its one-ring, T = argmin sum_j c_ij |(p'_i - p'_j) - T (p_i - p_j)|^2 with cotangent weights of
 * the rest mesh (ACAP / ARAP deformation gradient), is split by polar decomposition T = Q S (Q proper rotation, S symmetric).
 * Outputs, row-major [Vm,3,3]: R = Q^T (the row-vector convention deform_gaussian expects: it uses
 * gaussian_deform_rot = blend(R)^T and transforms covariances by R^T S, edittool/__init__.py:118-129) and S.
 * state (optional, float [Vm,21]) receives V1 | R | S per vertex - the frame record gm_pack_mesh_state consumes.
 * R and S: both or neither; at least one of (R, S) / state.
 * gm_mesh_rs_packed writes, instead, the 96-byte-per-vertex gather table gm_pack_mesh_state would make of that record
 * (float [Vm,24], 16-byte aligned): what gm_deform_shade_packed / gm_forward_0_deformed_async read - one launch per frame. */
int gm_mesh_rs(int Vm, int nfaces, const float* V0, const float* V1, const int* faces, const int* adj_offsets, const int* adj_faces,
               float* R, float* S, float* state, void* stream);
int gm_mesh_rs_packed(int Vm, int nfaces, const float* V0, const float* V1, const int* faces, const int* adj_offsets, const int* adj_faces,
                      float* packed, void* stream);

/* Covariance -> (scale, rotation): replaces the per-frame eigh + host-side det sign + sqrt + matrix->quaternion of
 * SceneVisualTool.render_gaussian (edittool/__init__.py:204-207, 23-38).  cov float [N,3,3] (symmetric),
 * scales float [N,3] = sqrt of the eigenvalues in ascending order, rots float [N,4] = unit quaternion (w,x,y,z) of the
 * eigenvector matrix made right-handed, such that R(q) diag(scales^2) R(q)^T reproduces cov. */
int gm_cov_to_scale_rot(int N, const float* cov, float* scales, float* rots, void* stream);

/* Photometric loss of the training loop (train_mesh_gaussian.py:92-94): replaces utils/loss_utils.py:17-18 (l1_loss) and
 * :23-81 (ssim: 11x11 Gaussian window of sigma 1.5, zero padding 5, depthwise) and the autograd pass through them.
 * img1 (rendered) / img2 (ground truth): float [planes,H,W] (planes = channels, or batch x channels).
 * gm_ssim_fwd writes per 32x32-pixel workgroup {sum of the ssim map, sum of |img1-img2|} into partial
 * (float [gm_ssim_partials(planes,H,W)][2], plane-major then tile rows): ssim(...) = sum/(planes H W), per-image means
 * for size_average=False by summing a plane range.  dS_dmu1 / dS_dE11 / dS_dE12 (float [planes,H,W] each; all three or
 * all NULL) receive the per-pixel partial derivatives gm_ssim_bwd needs.
 * gm_ssim_bwd: dL_dimg1 = g_ssim[plane] * d(sum of ssim map)/d img1 + g_l1[0] * sign(img1 - img2); g_ssim (device float
 * [planes]) and g_l1 (device float [1], may be NULL) carry the upstream gradient times 1/count, so no host
 * synchronisation is needed between forward and backward.
 * gm_loss_combine: out[0] = offset + c_ssim * sum_i partial[i][0] + c_l1 * sum_i partial[i][1] (sums in double, one launch):
 * with c_ssim = -lambda/count, c_l1 = (1 - lambda)/count, offset = lambda this is the training loss
 * (1 - lambda) * l1_loss + lambda * (1 - ssim) of train_mesh_gaussian.py:92-94 as a device scalar. */
int64_t gm_ssim_partials(int planes, int H, int W);
int gm_ssim_fwd(const float* img1, const float* img2, int planes, int H, int W, float* dS_dmu1, float* dS_dE11, float* dS_dE12,
                float* partial, void* stream);
int gm_ssim_bwd(const float* img1, const float* img2, const float* dS_dmu1, const float* dS_dE11, const float* dS_dE12, int planes,
                int H, int W, const float* g_ssim, const float* g_l1, float* dL_dimg1, void* stream);
int gm_loss_combine(const float* partial, int64_t n_partials, double c_ssim, double c_l1, double offset, float* out, void* stream);

/* Training-loop fusions around the rasterizer ("MeshBasedGaussianModel.get_xyz ... fused into the op").
 * gm_mesh_activate_fwd: raw parameters -> rasterizer inputs in one pass, replacing the Jittor elementwise chains of
 *   get_xyz = softmax(bc).(v1,v2,v3) + alpha r (sigmoid(dist) - 0.5) normal   (scene/mesh_based_gaussian_model.py:138-152)
 *   get_scaling = exp(scaling), get_rotation = normalize(rotation) (:122-128), get_opacity = sigmoid(opacity) (:172-174).
 *   bc/scaling/v1/v2/v3/normal float [N,3]; dist/opacity/r float [N]; rotation float [N,4] (16-byte aligned).
 *   Outputs xyz [N,3], scales [N,3], rots [N,4] (16-byte aligned), opac [N].
 *   mr_partial (may be NULL; float [ceil(N/256)]): per-workgroup sums of the mesh-restrict loss term
 *   max(0, max_axis(scales) - mr_weight * sqrt(|AB x AC|)) (utils/loss_utils.py:86-108, train_mesh_gaussian.py:93); their
 *   sum is mesh_restrict_loss(scales, v1, v2, v3, mr_weight).
 * gm_mesh_activate_bwd: the adjoint (what Jittor's autograd derives op by op); d_xyz / d_scales / d_rots / d_opac may be
 *   NULL (= zero); d_mr (device float [1], may be NULL) is the upstream gradient of that loss term; writes d_bc, d_dist,
 *   d_scaling, d_rotation, d_opacity. */
int gm_mesh_activate_fwd(int N, float alpha, const float* bc, const float* dist, const float* scaling, const float* rotation,
                         const float* opacity, const float* v1, const float* v2, const float* v3, const float* normal, const float* r,
                         float* xyz, float* scales, float* rots, float* opac, float mr_weight, float* mr_partial, void* stream);
int gm_mesh_activate_bwd(int N, float alpha, const float* bc, const float* dist, const float* scaling, const float* rotation,
                         const float* opacity, const float* v1, const float* v2, const float* v3, const float* normal, const float* r,
                         const float* d_xyz, const float* d_scales, const float* d_rots, const float* d_opac, float* d_bc, float* d_dist,
                         float* d_scaling, float* d_rotation, float* d_opacity, float mr_weight, const float* d_mr, void* stream);

/* jittor.nn.Adam's update (the optimizer of training_setup, scene/mesh_based_gaussian_model.py:242-263) for up to 8
 * parameter tensors in one launch:  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 *   p -= lr sqrt(1-b2^step)/(1-b1^step) m / (sqrt(v) + eps).
 * The arrays are HOST arrays of `count` entries (device pointers, element counts, learning rates).  period/split/lr_rest
 * (may be NULL) give a tensor two rates: elements with (index % period) < split use lr, the others lr_rest - the SH
 * tensor [P,16,3] with period 48, split 3 is the reference's "f_dc" and "f_rest" groups without splitting the rows.
 * All tensors 16-byte aligned; gradients are read, not cleared.  beta1 / beta2 / eps are doubles: Jittor forms (1 - beta) in
 * python double precision before it meets the float32 tensors, and (1 - beta2) taken from a float32 0.999 is off by 1.3e-5. */
int gm_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 const uint64_t* sizes, const float* lr, const float* lr_rest, const uint32_t* period, const uint32_t* split,
                 double beta1, double beta2, double eps, int step, void* stream);
/* The same with `active` (host array, may be NULL; 0 = the whole tensor): of every period of tensor i only the elements with
 * (index % period) < active[i] - rounded up to a 16-byte granule - are read and written.  For the SH tensor while the model's
 * active degree D is below its maximum (train_mesh_gaussian.py:70-71 raises it every 1000 iterations): active = 3 (D+1)^2.
 * Coefficients above the active degree have g = m = v = 0, the update rule leaves them unchanged, so skipping them gives the
 * result of gm_adam_step bit for bit (the caller guarantees their gradients are zero: the rasterizer's backward writes zeros
 * there).  Needs sizes[i] % period[i] == 0. */
int gm_adam_step_active(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                        const uint64_t* sizes, const float* lr, const float* lr_rest, const uint32_t* period, const uint32_t* split,
                        const uint32_t* active, double beta1, double beta2, double eps, int step, void* stream);

/* Densification statistics of a training iteration in one pass (train_mesh_gaussian.py:119-126 and
 * scene/mesh_based_gaussian_model.py:587-589): for every Gaussian with radii[i] > 0 (render()'s visibility_filter)
 *   max_radii2D[i] = max(max_radii2D[i], radii[i]);  grad_accum[i] += |viewspace_grad[i, 0:2]|;  denom[i] += 1.
 * radii int32 [N]; viewspace_grad float [N,3] (the gradient of means2D); the three accumulators float [N]. */
int gm_densify_stats(int N, const int* radii, const float* viewspace_grad, float* max_radii2D, float* grad_accum, float* denom, void* stream);

/* Per-stage GPU timing (HIP events recorded on `stream` around each kernel group).  Off by default.
 * gm_profile_enable(1) starts collecting, gm_profile_read synchronises the recorded events and returns
 * accumulated milliseconds and launch count for a stage name ("preprocess","depth_sort","scan",
 * "duplicate","tile_sort","ranges","render","render_bwd","preprocess_bwd","deform","sh_colors","loss","loss_bwd");
 * gm_profile_reset clears the accumulators. */
void gm_profile_enable(int on);
void gm_profile_reset(void);
int gm_profile_read(const char* stage, double* total_ms, int64_t* launches);

/* Debugging aids of the repository's tools, never called by the package: while a device buffer is registered the forward
 * blend (tools/wave_trace.py: 8 x uint64 per wave - start / end clock, list length, entries evaluated ...) / the depth-bucket
 * sort and the two scatter kernels of the ordering (tools/bucket_stats.py, tools/pipeline_trace.py: 3 x uint64 per workgroup -
 * start / end clock, entries; 2048 records for the bucket sort, then 2048 for the depth partition's scatter, then 4096 for the tile
 * pass's: 3 * 8192 words) write per-wave / per-workgroup records into it; NULL switches the tracing off again.
 * Process-wide, not thread-safe.
 * One more measurement aid, read once from the environment: GM_DEBUG_STOP_AFTER = deform | depth | dup | tile makes every forward
 * stop launching after that stage (tools/stage_marginal.sh times the pipelined loop with it: what the remaining stages cost);
 * the frames are then garbage.  Unset in any real use. */
void gm_debug_render_trace(void* buffer);
void gm_debug_bucket_trace(void* buffer);

#ifdef __cplusplus
}
#endif
#endif
