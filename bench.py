#!/usr/bin/env python
"""bench.py -- headline benchmark of the GaussianMesh hot path on MI355X.

Workload (BASELINE.json configs[2] = "C3", the configuration the metric is quoted on: 1 M Gaussians @ 1080p):
1 M Gaussians bound to a 15k-face torus proxy mesh; one STEP = one frame of the edit-tool loop
(edittool/__init__.py:103-131, 400-475 of the reference):
    deform + view-dependent colour (gm_deform_shade, one fused pass: mesh state (V1,R,S) of frame t -> x', Sigma',
    rotated view direction, SH degree 3 -> colors_precomp)
  + rasterize forward (gm_forward_0/1 with colors_precomp + cov3D_precomp) at 1920x1080.
All inputs are resident in HBM before the timed region.  With --gpus N every rank renders its own camera of the 64-camera orbit
(views shard); rank 0 owns the mesh animation and broadcasts the deformed VERTEX POSITIONS of eight loop steps at a
time (90 KB per step) over RCCL, one batch ahead of their use, every rank derives the per-vertex (R, S) itself (gm_mesh_rs); the
static cloud is broadcast once before timing.  value = frames of all ranks per second (weak scaling).

One JSON line on stdout (rank 0).  Extra objects: "roofline" (dominant kernel of the frame, HIP-event timing from gm_profile_*),
"frame_roofline" (whole frame), "cpu_baseline" (oracle port on the host cores, N=1 only), "fwd_bwd" (ms/iter of forward + backward
on the same cloud through the autograd operator, with its stage times, "roofline" of the backward blend and "iteration_roofline";
also the iteration with the L1 + SSIM loss, a whole training iteration and BASELINE config C2; N=1 only), "c5" (a bounded leg of
BASELINE config C5: --c5-iters iterations of the 3 M-Gaussian 4K training loop; `--config c5` runs its 1000 iterations as a line
of its own), "c5_phases" / "c5_fixed" (the same loop at each SH degree / at degree 3 with the topology fixed), "batch" (per-frame stage times
and traffic of the K-frame launch chain the headline runs), "repeats" (the timed region four more times) and "single_stream" (latency of one
frame: with the instance count read back by the host, and enqueued sync-free in one go).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s measured achievable

STAGES = ["mesh_rs", "deform", "sh_colors", "preprocess", "depth_sort", "duplicate", "tile_sort", "ranges", "render"]
# stages that are ONE kernel launch per frame (gm_profile_* brackets it alone): the candidates for `roofline`, the dominant KERNEL
KERNEL_OF_STAGE = {"mesh_rs": "gm::mesh_rs_kernel", "deform": "gm::deform_shade_kernel<true,true,false>", "duplicate": "gm::duplicate_kernel",
                   "render": "gm::render_fwd_kernel", "sh_colors": "gm::sh_colors_kernel", "preprocess": "gm::preprocess_fwd_kernel"}


# The frame loop rotates over four HIP streams; the runtime multiplexes streams onto 4 hardware queues by default, and streams
# that share a queue serialise against each other (4 streams on 4 queues: 3460 frames/s, on 8 queues: 4300): GPU_MAX_HW_QUEUES=8.
# N > 1: RCCL sets up its xGMI peer buffers through HIP IPC handles, and the host driver of these nodes supports only the dmabuf
# flavour: with the legacy mode left on, the first collective fails with `hipIpcGetMemHandle: invalid argument`
# (HSA_ENABLE_IPC_MODE_LEGACY=0; the image exports it, set here as well so that a launch from a clean environment - the driver's
# torch.distributed.run, a test's subprocess - runs in the same mode; harmless at N = 1).  Both must be in the environment before
# the HIP runtime initialises; the package does not touch the environment on import, the integrator - here bench.py - calls
# configure_runtime() (GM_NO_RUNTIME_CONFIG=1 leaves the environment to the launcher).
import gaussianmesh_amd
RUNTIME = gaussianmesh_amd.configure_runtime(hw_queues=8, ipc_dmabuf=True)


class DistStage:
    """Failure surface of the N > 1 start-up (round 5): the first RCCL collectives this code ever issues are the driver's scaling
    run.  A stage that raises, or that is still running after `seconds` (a hang inside a collective never returns to Python: a timer
    thread watches it), ends the process with ONE parseable JSON line - {"error", "stage", "rank", "rank_env"} - on stdout (rank 0)
    or stderr (other ranks) and a non-zero exit code, instead of a traceback on one rank and a silent hang on the others."""

    def __init__(self, stage, rank, world, seconds, debug_file=None):
        self.stage, self.rank, self.world, self.seconds, self.debug_file = stage, rank, world, seconds, debug_file
        self.timer = None

    def report(self, err, code):
        tail = None
        try:
            if self.debug_file and os.path.exists(self.debug_file):
                tail = open(self.debug_file, errors="replace").read()[-1500:]
        except OSError:
            pass
        env = {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "NCCL_DEBUG_FILE", "MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE",
                                              "RANK", "LOCAL_RANK", "GM_BENCH_BACKEND", "GPU_MAX_HW_QUEUES", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")}
        line = json.dumps({"error": str(err)[-2000:], "stage": self.stage, "rank": self.rank, "n_gpus": self.world,
                           "rank_env": dict(env, nccl_debug_tail=tail), "metric": "frames/sec (fwd), 1M Gaussians @1080p, deform+render", "value": None})
        (sys.stdout if self.rank == 0 else sys.stderr).write(line + "\n")
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(code)                           # (not sys.exit: a rank stuck in a collective has threads that never join)

    def __enter__(self):
        import threading
        self.timer = threading.Timer(self.seconds, lambda: self.report("stage still running after %.0f s (hang)" % self.seconds, 3))
        self.timer.daemon = True
        self.timer.start()
        return self

    def __exit__(self, et, ev, tb):
        self.timer.cancel()
        if et is not None and not issubclass(et, (SystemExit, KeyboardInterrupt)):
            import traceback
            self.report("%s: %s | %s" % (et.__name__, ev, "".join(traceback.format_tb(tb)[-2:]).replace("\n", " ")), 2)
        return False


def build_scene(P, W, H, frames, seed=0):
    from gaussianmesh_amd import scenes
    verts, faces = scenes.torus_mesh(100, 75)
    cl = scenes.bind_cloud_to_mesh(P, verts, faces, seed=seed)
    # Memory layout: Gaussians stored grouped by the face they are bound to (as the reference's model creates them:
    # one per face, then face-splitting densification, scene/mesh_based_gaussian_model.py:183-241, 596-647), so the
    # per-vertex (dV,R,S) gathers of neighbouring Gaussians hit the same rows.
    perm = np.argsort(cl["fid"], kind="stable")
    for k in ("tri", "weights", "means", "opac", "shs", "scales", "rots", "fid"):
        cl[k] = np.ascontiguousarray(cl[k][perm])
    cov = scenes.cov3d_from_scale_rot(cl["scales"], cl["rots"]).astype(np.float32)
    mesh = np.zeros((frames, verts.shape[0], 21), np.float32)
    for t in range(frames):
        V1, Rv, Sv = scenes.twist_bend_frame(verts, t, period=frames)
        mesh[t, :, 0:3] = V1
        mesh[t, :, 3:12] = Rv.reshape(-1, 9)
        mesh[t, :, 12:21] = Sv.reshape(-1, 9)
    return dict(verts=verts.astype(np.float32), faces=faces.astype(np.int32), tri=cl["tri"], weights=cl["weights"], pos=cl["means"], cov=cov,
                opac=cl["opac"], shs=cl["shs"], scales=cl["scales"], rots=cl["rots"], mesh=mesh)


def algorithmic_bytes(stage, P, V, R, W, H, Vm, M=16, list_tiles=2040, cov_bytes=36, batch=1):
    """Bytes each stage has to move for THIS implementation's algorithm, precomputed colour/cov input mode."""
    hist = 2048 * 4                     # one histogram row per 4096 keys
    one_pass = list_tiles <= 2048
    return {
        "deform": P * (12 + 12 + 36 + 12 + 12 * M) + P * (12 + 24 + 12) + Vm * 84,     # fused deform + colour
        # deform + colour + forward preprocess in one kernel: the 48 B/Gaussian of intermediates disappear, the
        # preprocess outputs (splat record 36 B per visible Gaussian; radius, count, bin, depth key 28 B) and opacity appear
        "deform_pre": P * (12 + 12 + cov_bytes + 12 + 12 * M + 4) + Vm * 96 + V * 36 + P * 28,
        # the same pass for ONE frame of a batch of K (gm_forward_deformed_batch_async): the static cloud's bytes are read once per batch
        "deform_pre_batch": P * (12 + 12 + cov_bytes + 12 + 12 * M + 4) // max(batch, 1) + Vm * 96 + V * 36 + P * 28,
        # per-vertex (R, S) from the deformed mesh: rest + deformed positions, one-ring face ids (~6 faces x (4 + 12)), the 96-byte table row
        "mesh_rs": Vm * (24 + 96 + 96),
        "sh_colors": P * (12 + 36 + 12 * M) + P * 12,
        "preprocess": P * (12 + 24 + 4 + 12) + V * 36,
        # bucket partition (key read twice, (key, id) written once) + in-LDS bucket sort ((key, id) in; id, count out; count gather)
        # + the move of every visible Gaussian's 16-byte emission record into depth order (read + write): DESIGN.md section 3's 76 MB
        "depth_sort": P * 8 + V * 8 + V * (8 + 4 + 4 + 4) + V * 32 + (P // 4096 + 1) * hist * 4,
        # direct depth placement (DepthPlan): the fused pass writes (key, id) + the emission record once, into the bucket's slab,
        # instead of the per-Gaussian key and record arrays; the depth order is then the bucket counters and the in-LDS bucket sort
        # (slab in; id and record out) - no partition, no gather
        "deform_pre_direct": P * (12 + 12 + cov_bytes + 12 + 12 * M + 4) + Vm * 96 + V * 36 + P * 8 + V * 24,
        "depth_sort_direct": V * (8 + 16) + V * (4 + 16) + 2048 * (4 + 4) + 16384 * 4,
        "duplicate": V * (4 + 4) + V * 16 + R * 8,                                  # counts + ids in order, bin records, (key, id) out
        "tile_sort": (R * (4 + 8 + 8) + (R // 4096 + 1) * hist * 4) * (1 if one_pass else 2) + list_tiles * 8,
        "ranges": list_tiles * 12,                                                  # tile_order_kernel: ranges in, dispatch order out
        "render": R * 40 + W * H * 12,
    }[stage]


def measured_traffic(stage, P, W, H, batch=1):
    """HBM bytes per launch of a stage from the committed PMC passes (profiles/hbm_traffic.json: FETCH_SIZE / WRITE_SIZE
    cannot be collected from inside a timed run), or None when this workload was not the one profiled.
    batch > 1: the launches over `batch` frames of gm_forward_deformed_batch_async (profiles/hbm_traffic_batch.json, per launch)."""
    import json
    try:
        t = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json" if batch <= 1 else "hbm_traffic_batch.json")))
    except OSError:
        return None
    w = t["workload"]
    if (w["gaussians"], w["width"], w["height"]) != (P, W, H) or stage not in t["stages"] or (batch > 1 and t.get("frames_per_launch") != batch):
        return None
    e = t["stages"][stage]
    return int(1024 * (2 * e["fetch_kib"] + e["write_kib"]))


def measured_valu(stage, P, W, H):
    """Vector instructions per launch of a stage from the committed PMC pass (profiles/inst_mix.json), or None."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "inst_mix.json")))
    except OSError:
        return None
    w = t["workload"]
    if (w["gaussians"], w["width"], w["height"]) != (P, W, H) or stage not in t["stages"]:
        return None
    return t["stages"][stage].get("valu")


def valu_ceiling(stage, nv, avg_ms, frames=1):
    """Vector-ALU time of a kernel against its duration.  With SQ_ACTIVE_INST_VALU in profiles/inst_mix.json (round 5: the blend
    kernels) the busy time is measured; otherwise the instruction count is priced at the mix average of the blend kernels, 4.2 cycles
    (measured: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.24 forward, 4.17 backward - NOT the 2.3 cycles of a plain v_fma_f32: the mix is
    half compares, selects, min / max, packed and transcendental instructions, tools/valu_probe)."""
    quads = None
    try:
        quads = json.load(open(os.path.join(ROOT, "profiles", "inst_mix.json")))["stages"][stage].get("valu_active_quad_cycles")
    except (OSError, KeyError):
        pass
    if quads:
        quads *= frames                          # (a launch over `frames` frames: the committed counters are per single-frame launch)
    busy_ms = (4.0 * quads if quads else 4.2 * nv) / (1024 * 2.4e9) * 1e3
    return {"vector_instructions": nv, "valu_busy_ms": busy_ms, "frac_of_kernel_time": busy_ms / avg_ms,
            "cycles_per_instruction": (4.0 * quads / nv) if quads else 4.2,
            "source": "static: profiles/inst_mix.json (rocprofv3 --pmc SQ_INSTS_VALU" + (", SQ_ACTIVE_INST_VALU)" if quads else "; priced at 4.2 cycles per instruction)")}


# the "feature" rows of the C5 teacher that the student has misplaced (build_c5(teacher=True)): fraction of the rows, scale factor, opacity
# logit, displacement in log-barycentric units.  GM_C5_STUDENT="fraction,scale,logit,shift" overrides it (tools/c5_densify_sweep.py).
C5_STUDENT_BIG = (0.02, 4.0, 3.0, 2.0)


def build_c5(Nfg=2_000_000, Nbg=1_000_000, W=3840, H=2160, dev=None, sync_free=True, ncams=32, seed=0, teacher=False, scene="r05"):
    """BASELINE config C5 on this package's training harness: Nfg Gaussians bound to the 15 k-face torus + Nbg free, frozen
    "background" Gaussians in a shell of radius 6-12 that contains the cameras, W x H, a Trainer with FusedAdam on the six
    parameter groups, densification statistics, sync-free forward, and `ncams` orbit cameras.
    teacher=False: one fixed random target; returns (trainer, cameras, target, background colour) - also used by
    tests/test_gpu_fullsize.py.
    teacher=True: the cloud as generated is the TEACHER; every camera's target is its render at SH degree 3, kept as (colour over a
    black background, final transmittance) so that the loop can composite it over that iteration's random background exactly as
    train_mesh_gaussian.py:92-93 does with the dataset's mask; the STUDENT the trainer gets starts grey (all SH coefficients 0)
    and half transparent at SH degree 0, on the teacher's geometry.  Returns (trainer, cameras, (colour [ncams,3,H,W],
    transmittance [ncams,1,H,W]), None)."""
    import torch
    from types import SimpleNamespace
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.renderer import Camera, MeshBoundGaussians, render
    from gaussianmesh_amd.train import FrozenGaussians, Trainer
    dev = dev or torch.device("cuda", 0)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    verts, faces = scenes.torus_mesh(100, 75)
    cl = scenes.bind_cloud_to_mesh(Nfg, verts, faces, seed=seed)
    perm = np.argsort(cl["fid"], kind="stable")
    tri = cl["tri"][perm]
    v1, v2, v3 = (t(verts[tri[:, k]]) for k in range(3))
    nrm = torch.nn.functional.normalize(torch.cross(v2 - v1, v3 - v1, dim=1), dim=1)
    rad = (((v2 - v1).norm(dim=1) + (v3 - v2).norm(dim=1) + (v1 - v3).norm(dim=1)) / 3)[:, None]
    shs = t(cl["shs"][perm])
    model = MeshBoundGaussians(torch.log(t(cl["weights"][perm]).clamp_min(1e-6)), torch.zeros((Nfg, 1), device=dev), shs[:, :1].clone(),
                               shs[:, 1:].clone(), torch.log(t(cl["scales"][perm])), t(cl["rots"][perm]),
                               torch.logit(t(cl["opac"][perm]).reshape(-1, 1).clamp(1e-4, 1 - 1e-4)), v1, v2, v3, nrm, rad,
                               fid=torch.tensor(cl["fid"][perm].astype(np.int32), device=dev)).to(dev)
    b = scenes.make_cloud(Nbg, seed=seed + 1, extent=1.0)
    nb = np.linalg.norm(b["means"], axis=1, keepdims=True) + 1e-6
    bg = FrozenGaussians(t(b["means"] / nb * (6 + 6 * nb)), t(b["scales"]), torch.nn.functional.normalize(t(b["rots"])), t(b["opac"]).reshape(-1, 1),
                         t(b["shs"]))
    cams = [Camera(scenes.orbit_camera(k, ncams, W, H), dev) for k in range(ncams)]
    if not teacher:
        target = torch.rand((3, H, W), device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
        tr = Trainer(model, densify_stats=True, sync_free=sync_free, bg_gaussian=bg)
        return tr, cams, target, torch.zeros(3, device=dev)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    colour = torch.empty((ncams, 3, H, W), device=dev); trans = torch.empty((ncams, 1, H, W), device=dev)
    zero, one = torch.zeros(3, device=dev), torch.ones(3, device=dev)
    frac, bscale, blogit, bshift = C5_STUDENT_BIG
    if os.environ.get("GM_C5_STUDENT"):                            # tools/c5_densify_sweep.py
        frac, bscale, blogit, bshift = (float(x) for x in os.environ["GM_C5_STUDENT"].split(","))
    with torch.no_grad():
        # "features" of the TEACHER: one Gaussian in fifty is `bscale` times larger than its neighbours, nearly opaque and of a saturated
        # colour - structure at a scale of tens of pixels in every target.  The student below has these rows MISPLACED; a positional
        # gradient needs an error that changes sign across the splat, which a splat of the wrong colour or size over a smooth target
        # does not produce (round 4: one row over densify_grad_threshold in 2 M, see densify_note)
        g = torch.Generator(device=dev).manual_seed(seed + 7)
        big = torch.rand(model._bc.shape[0], device=dev, generator=g) < frac
        if scene == "r05":
            model._scaling[big] += math.log(bscale)
            model._opacity[big] = blogit
            model._features[big, 0] = 1.5 * torch.randn((int(big.sum()), 3), device=dev, generator=g)
        for k, c in enumerate(cams):
            colour[k] = render(c, model, pipe, zero, bg_gaussian=bg)["render"]
            trans[k] = (render(c, model, pipe, one, bg_gaussian=bg)["render"][:1] - colour[k][:1]).clamp(0.0, 1.0)   # C + T.1 - C
        # the student: the teacher's cloud dimmed (a quarter of its base colour, no view dependence), a quarter opaque, shrunk by a
        # fifth and pushed around inside its faces - so that colour, opacity, scale AND position have something to learn - at SH degree
        # 0, where the reference starts; the feature rows keep their look and are pushed `bshift` times further (log-barycentric
        # units): under-reconstructed regions whose mean view-space gradient crosses the reference's threshold, so that the topology
        # changes of iterations 600 / 800 / 1000 split one to two percent of the rows as they do on real data
        small = ~big if scene == "r05" else torch.ones_like(big)
        model._features[small, 0] *= 0.25
        model._features[:, 1:] = 0.0
        model._opacity[small] = -1.0
        model._scaling[small] += math.log(0.8)
        shift = torch.randn(model._bc.shape, device=dev, generator=g)
        model._bc += torch.where((big & ~small)[:, None], bshift * shift, 0.5 * shift)
        if scene != "r05":                       # the round-4 scene, kept for comparison across rounds (c5_r04_scene): no features in the teacher; one
            model._scaling[big] += math.log(4.0)  # student row in fifty four times too large and of a wrong colour - ONE row over the threshold in 2 M
            model._features[big, 0] = 1.5 * torch.randn((int(big.sum()), 3), device=dev, generator=g)
    model.active_sh_degree = 0
    tr = Trainer(model, densify_stats=True, sync_free=sync_free, bg_gaussian=bg)
    return tr, cams, (colour, trans), None


def c5_leg(steps, warm, Nfg=2_000_000, Nbg=1_000_000, W=3840, H=2160, sync_free=True, policy=None, work_hint=True, dev=None, as_reference=False,
           ncams=32, scene="r05"):
    """The "c5" object of the bench line.
    as_reference=False (the leg rounds 1-3 reported, kept for comparison): `steps` Trainer.step iterations after `warm` untimed ones at
    a fixed topology, SH degree 3, a zero background and one fixed random target.
    as_reference=True: iterations 1 .. `steps` of train_mesh_gaussian.py AS IT RUNS THEM (Trainer.train_iteration): SH degree 0
    (one up at iteration 1000, :70-71), a camera popped at random without replacement (:74-76), a random background per iteration
    because a background cloud exists (:85), the target composited over it (:92-93), densification statistics every iteration,
    densify_and_prune(0.0002, 0.005, extent, None, 5) at iterations 600 / 800 / 1000 with the optimizer step of those iterations
    skipped (:126-139) - all inside the timed region, no warm-up (the first iteration IS iteration 1)."""
    import random
    import torch
    from gaussianmesh_amd import rasterizer as Rz
    from gaussianmesh_amd.renderer import set_work_hints
    set_work_hints(work_hint)                    # (per-camera work hints of the forward blend's dispatch order)
    dev = dev or torch.device("cuda", 0)
    if policy is not None:
        Rz.set_default_emission_policy(policy)
    tr, cams, target, zero_bg = build_c5(Nfg, Nbg, W, H, dev, sync_free, ncams=ncams, teacher=as_reference, scene=scene)
    nc = len(cams)
    losses = []
    if not as_reference:
        for i in range(warm):
            losses.append(tr.step(cams[i % nc], target, zero_bg)[0])
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        redone0 = tr.redone
        t0 = time.perf_counter()
        for i in range(steps):
            loss, pkg = tr.step(cams[(warm + i) % nc], target, zero_bg)
            if i % 50 == 0 or i == steps - 1:
                losses.append(loss)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out = {"ms_per_iter": 1e3 * el / steps, "iters": steps, "warmup": warm, "peak_memory_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
               "iterations_redone": tr.redone - redone0, "gaussians": Nfg + Nbg, "trainable": Nfg, "width": W, "height": H,
               "visible": int((pkg["radii"] > 0).sum().item()), "emission_policy": Rz.get_default_emission_policy(W, H), "sync_free": bool(sync_free),
               "loss_first": float(losses[0]), "loss_last": float(losses[-1]), "sh_degree": 3, "sh_steps_inside_the_backward": tr.sh_steps_fused,
               "workload": "C5 (fixed topology): %d mesh-bound + %d frozen free Gaussians, %dx%d, render + L1/SSIM/mesh-restrict loss + backward + FusedAdam + "
                           "densification statistics, %d orbit cameras, SH degree 3, zero background, fixed random target" % (Nfg, Nbg, W, H, nc)}
        del tr, cams, target, pkg
        torch.cuda.empty_cache()
        return out
    colour, trans = target
    # the one-time costs of the FIRST topology change (the runtime loads the code objects of a dozen torch kernels: 60 ms) are paid
    # here, on a 256-Gaussian model, not inside iteration 600 of the timed loop
    tiny = build_c5(256, 64, 64, 48, dev, sync_free, ncams=2, teacher=True)
    for _ in range(2):
        tiny[0].step(tiny[1][0], torch.addcmul(tiny[2][0][0], tiny[2][1][0], torch.rand(3, device=dev).view(3, 1, 1)), torch.rand(3, device=dev))
    tiny[0].bc_gradient_accum[::2] = 1.0                             # every other row over the threshold: 128 x 5 new rows, 128 pruned
    tiny[0].densify_and_prune(0.0002, 0.005, None, None, 5)
    tiny[0].step(tiny[1][1], tiny[2][0][1], torch.zeros(3, device=dev))
    del tiny
    rng = random.Random(0)
    stack = []
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats(dev)
    densify_ms, rows_after, marks, curve, quant = [], [], {}, {}, []
    t0 = time.perf_counter()
    for it in range(1, steps + 1):
        if not stack:
            stack = list(range(nc))
        k = stack.pop(rng.randint(0, len(stack) - 1))                    # :74-76
        bgc = torch.rand(3, device=dev)                                  # :85 (device generator: no host round trip)
        gt = torch.addcmul(colour[k], trans[k], bgc.view(3, 1, 1))       # :92-93 gt * mask + bg * (1 - mask), with the teacher's transmittance
        plan = tr.schedule(it)
        if plan["densify"]:                                              # bracket the three topology changes (three extra host syncs in 1000 iterations)
            torch.cuda.synchronize()
            if not quant:                                                # what densify_and_prune will see the first time (diagnostic, outside the bracket)
                gq = (tr.bc_gradient_accum / tr.denom.clamp_min(1)).reshape(-1)
                ks = [int(f * (gq.numel() - 1)) for f in (0.5, 0.9, 0.99, 0.999)]
                srt = torch.sort(gq).values
                quant.extend([float(srt[k]) for k in ks] + [float(srt[-1]), float((gq >= tr.opt.densify_grad_threshold).float().mean())])
                del gq, srt
                torch.cuda.synchronize()
            td = time.perf_counter()
        loss, pkg, plan = tr.train_iteration(cams[k], gt, bgc)
        if plan["densify"]:
            torch.cuda.synchronize()
            densify_ms.append(1e3 * (time.perf_counter() - td)); rows_after.append(plan["rows"])
        if it in (1, 599, 600):
            torch.cuda.synchronize(); marks[it] = time.perf_counter()
        if it <= 5 or it > steps - 20:
            losses.append(loss)
        if it in (1, 2, 5, 10, 20, 50, 100, 200, 400, 599, 601, 700, 799, 801, 900, 999, steps):
            curve[it] = loss
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # What one topology change costs at this size, whatever the reference's threshold selected above (on this synthetic cloud - 2 M
    # small Gaussians on a 4K image - no row's mean view-space gradient reaches 0.0002, see viewspace_grad_at_first_densify): split
    # the 2 % of the rows with the largest statistic (N = 5, originals pruned: +8 % rows), timed on its own, then 20 more
    # iterations of the same loop at the new row count.  Outside the 1000 iterations that define ms_per_iter.
    forced = None
    if steps >= 600:
        # (iteration 1000 may just have split rows and reset the statistics: 60 more iterations of the loop give every row a statistic again)
        for it in range(60):
            if not stack:
                stack = list(range(nc))
            k = stack.pop(rng.randint(0, len(stack) - 1))
            bgc = torch.rand(3, device=dev)
            tr.step(cams[k], torch.addcmul(colour[k], trans[k], bgc.view(3, 1, 1)), bgc)
        gq = (tr.bc_gradient_accum / tr.denom.clamp_min(1)).reshape(-1)
        thr = float(torch.sort(gq).values[int(0.98 * (gq.numel() - 1))])
        del gq
        rows0 = int(tr.g._bc.shape[0])
        torch.cuda.synchronize(); td = time.perf_counter()
        tr.densify_and_prune(max(thr, 1e-30), 0.005, None, None, 5)
        torch.cuda.synchronize(); t_split = 1e3 * (time.perf_counter() - td)
        td = time.perf_counter()
        for it in range(20):
            if not stack:
                stack = list(range(nc))
            k = stack.pop(rng.randint(0, len(stack) - 1))
            bgc = torch.rand(3, device=dev)
            loss, pkg = tr.step(cams[k], torch.addcmul(colour[k], trans[k], bgc.view(3, 1, 1)), bgc)
        torch.cuda.synchronize()
        forced = {"rows_before": rows0, "rows_after": int(tr.g._bc.shape[0]), "densify_and_prune_ms": round(t_split, 3),
                  "ms_per_iter_after": 1e3 * (time.perf_counter() - td) / 20, "iterations_redone_after": tr.redone,
                  "selection": "rows whose mean view-space gradient is in the top 2 percent (threshold %.3g)" % thr}
    lf = float(torch.stack(losses[:5]).mean()); ll = float(torch.stack(losses[-20:]).mean())
    out = {"ms_per_iter": 1e3 * el / steps, "iters": steps, "warmup": 0, "peak_memory_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
           "iterations_redone": tr.redone, "gaussians": Nfg + Nbg, "trainable": Nfg, "trainable_at_end": int(tr.g._bc.shape[0]), "width": W, "height": H,
           "visible": int((pkg["radii"] > 0).sum().item()), "emission_policy": Rz.get_default_emission_policy(W, H), "sync_free": bool(sync_free),
           "loss_first": lf, "loss_last": ll, "loss_ratio": ll / lf, "sh_degree": "0, 1 from iteration 1000",
           "loss_first_note": "mean of iterations 1-5 / of the last 20 (a random camera each)", "loss_curve": {str(k): round(float(v), 5) for k, v in curve.items()},
           # iterations 2 .. 599: before the first topology change (iteration 1 carries the first allocations and the capacity seed)
           "ms_per_iter_before_first_densify": (1e3 * (marks[599] - marks[1]) / 598) if 599 in marks and 1 in marks else None,
           "densify_iterations_ms": [round(x, 3) for x in densify_ms],       # the WHOLE iteration that ends in densify_and_prune (no Adam step)
           "rows_after_densify": rows_after, "topology_changes": tr.resizes, "scene": scene,
           "densify_note": ("scene r05: 2 % of the teacher's rows are features (4 x larger, opaque, saturated) that the student has misplaced; at 4K with 2 M "
                            "Gaussians the reference's densify_grad_threshold = 0.0002 (a per-pixel loss weight of 1 / (3 H W)) is crossed by 0.04-0.05 % of the "
                            "rows - about a thousand rows split into five at each of 600 / 800 / 1000; a 2 % split is timed as forced_densify") if scene == "r05"
                           else "scene r04 (rounds 1-4): no features in the teacher; one row in 2 M crosses the threshold",
           "forced_densify": forced,
           "viewspace_grad_at_first_densify": dict(zip(("q50", "q90", "q99", "q999", "max", "fraction_over_threshold"), quant)),
           "workload": "C5 as train_mesh_gaussian.py runs its first %d iterations: %d mesh-bound + %d frozen free Gaussians, %dx%d, SH degree 0 "
                       "(1 at iteration 1000), random camera / random background per iteration, teacher-rendered targets composited over the "
                       "background, L1/SSIM/mesh-restrict loss, backward, FusedAdam, densification statistics, densify_and_prune(0.0002, N=5) at "
                       "600 / 800 / 1000 (optimizer step skipped there), %d orbit cameras" % (steps, Nfg, Nbg, W, H, nc)}
    del tr, cams, target, colour, trans, pkg
    torch.cuda.empty_cache()
    return out


def c5_phases(iters=200, warm=20, Nfg=2_000_000, Nbg=1_000_000, W=3840, H=2160, dev=None, ncams=32, degrees=(0, 1, 2, 3), trainer_kw=None):
    """ms per iteration of the C5 loop at EACH SH degree the reference's schedule passes through (train_mesh_gaussian.py:70-71 raises the
    degree every 1000 iterations: 1000 iterations at degree 0, 1000 at 1, 1000 at 2, 27 000 at 3), `iters` iterations each behind `warm`
    untimed ones, every phase from the SAME model state (the r05 student of build_c5): random camera without replacement, random
    background, teacher targets composited over it, statistics, Adam step - the loop of c5_leg(as_reference=True) without topology changes."""
    import random
    import torch
    from gaussianmesh_amd.train import Trainer
    dev = dev or torch.device("cuda", 0)
    tr, cams, (colour, trans), _ = build_c5(Nfg, Nbg, W, H, dev, True, ncams=ncams, teacher=True)
    model, bg = tr.g, tr.bg_gaussian
    model.end_dense_dc()
    names = ("_bc", "_distance", "_features", "_opacity", "_scaling", "_rotation")
    snap = {k: getattr(model, k).detach().clone() for k in names}
    del tr
    out = {"iters": iters, "warmup": warm, "gaussians": Nfg + Nbg, "width": W, "height": H}
    nc = len(cams)
    for d in degrees:
        with torch.no_grad():
            for k in names:
                getattr(model, k).copy_(snap[k])
        model.active_sh_degree = d
        tr = Trainer(model, densify_stats=True, sync_free=True, bg_gaussian=bg, **(trainer_kw or {}))
        rng, stack = random.Random(0), []
        t0 = None
        for it in range(warm + iters):
            if it == warm:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            if not stack:
                stack = list(range(nc))
            k = stack.pop(rng.randint(0, len(stack) - 1))
            bgc = torch.rand(3, device=dev)
            loss, pkg = tr.step(cams[k], torch.addcmul(colour[k], trans[k], bgc.view(3, 1, 1)), bgc)
        torch.cuda.synchronize()
        out["deg%d" % d] = round(1e3 * (time.perf_counter() - t0) / iters, 4)
        out["deg%d_redone" % d] = tr.redone
        out["deg%d_sh_operand" % d] = ("dense [N,%d,3] leaf" % tr.g._features_dc0.shape[1]) if getattr(tr.g, "_features_dc0", None) is not None else (
            "rows [N,16,3], Adam step inside the backward" if tr.sh_steps_fused else "rows [N,16,3]")
        if getattr(model, "_features_dc0", None) is not None:
            model.end_dense_dc()
        del tr, loss, pkg
    del snap, colour, trans, cams, model, bg
    torch.cuda.empty_cache()
    return out


def run_c5(args):
    """`bench.py --config c5`: the C5 training loop as the headline of its own JSON line (ms per iteration): the reference's first
    1000 iterations with their schedule (c5_leg(as_reference=True)); `--c5-fixed` times the fixed-topology SH-3 leg instead."""
    import torch
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(0)
    W, H = args.width if args.width != 1920 else 3840, args.height if args.height != 1080 else 2160
    Nfg, Nbg = ((2 * args.gaussians) // 3, args.gaussians // 3) if args.gaussians != 1_000_000 else (2_000_000, 1_000_000)
    steps = args.steps if args.steps != 300 else 1000
    c5 = c5_leg(steps, max(args.warmup, 5), Nfg, Nbg, W, H, sync_free=not args.exact_count, policy=args.policy, work_hint=not args.no_work_hint,
                as_reference=not args.c5_fixed, ncams=min(args.cameras, 32), scene=args.c5_scene)
    out = {"metric": "ms/iter (fwd+bwd+optimizer), 3M Gaussians @4K training loop", "value": c5["ms_per_iter"], "unit": "ms/iter", "n_gpus": 1,
           "steps": steps, "warmup": c5["warmup"], "ms_per_step": c5["ms_per_iter"], "higher_is_better": False, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": c5["workload"], "gaussians": Nfg + Nbg, "trainable": Nfg, "width": W, "height": H, "sh_degree": c5["sh_degree"],
                      "emission_policy": c5["emission_policy"], "sync_free": c5["sync_free"], "iterations_redone": c5["iterations_redone"]},
           "c5": c5}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=["c3", "c5"], help="c3 (default): the headline frames/s workload; c5: the "
                    "3 M-Gaussian 4K training loop (BASELINE config C5), reported as ms per iteration")
    ap.add_argument("--policy", type=int, default=None, help="emission policy 0..3 (default: the library's)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=4, help="extra timed regions of the same length after the one that defines `value` "
                    "(N=1 only): their frame rates and the median are reported beside it")
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--cameras", type=int, default=64)
    ap.add_argument("--unfused", action="store_true", help="deform+colour and the forward preprocess as two kernels (gm_deform_shade_packed, gm_forward_0_async)")
    ap.add_argument("--streams", type=int, default=4,
                    help="HIP streams the frame loop alternates over (frame i runs on stream i %% streams): consecutive frames are "
                         "independent, so frame i+1's per-Gaussian stages overlap the low-occupancy tail of frame i's blend")
    ap.add_argument("--frames-per-launch", type=int, default=4, help="K frames of the view stream per launch chain (gm_forward_deformed_batch_async): the "
                    "static cloud is read once per K frames and every stage is one launch over them; a batch goes to stream (batch index) %% streams.  "
                    "1 = a launch chain per frame (rounds 1-5); at most 8.  Measured (round 6, 300 / 20 steps): 1: 4960 / 4710 frames/s, 2: 5225 / 4980, 4: 5330-5400 / "
                    "5000-5170, 5: 5400 / 4930-5115, 6: 5480, 8: 5480 (on two streams 5510 / 5045): more frames per launch gain 3 %% in a long loop and nothing in a 20-frame burst")
    ap.add_argument("--begin-ahead", type=int, default=2, help="frames whose first half (deformation .. depth order) is issued before the "
                    "oldest of them is completed (emission .. blend): 2 fills the four streams sooner after the barrier that opens a timed "
                    "region than 1 (20-step regions, three runs each: 4389-4504 / 4540-4554 / 4039-4518 frames/s with 1 / 2 / 3; no "
                    "difference in steady state)")
    ap.add_argument("--status-lag", type=int, default=6, help="sync-free loop: the host reads a frame's status words this many frames "
                    "after completing it (how far the host may run ahead of the GPU).  With four frames in flight a lag of 3 still "
                    "makes the host wait for a frame that is running (4470 frames/s); 6 or more never does (4550-4570)")
    ap.add_argument("--exact-count", action="store_true", help="complete every frame with the instance count read back by the host "
                    "(gm_forward_1_geom's exact mode) instead of the sync-free mode")
    ap.add_argument("--check-dir", default=None, help="every rank saves the image of its last timed step (with the step, frame and "
                    "camera index) as <dir>/rank<r>.npz: lets a test verify that each rank rendered its own views")
    ap.add_argument("--analytic-rs", action="store_true", help="take the per-vertex (R, S) of every animation frame from the analytic "
                    "deformation (precomputed tables) instead of computing them from the deformed mesh inside the frame (gm_mesh_rs)")
    ap.add_argument("--no-cov6", action="store_true", help="edit loop: hand the rest covariances over as [N,3,3] even when they are bit-symmetric")
    ap.add_argument("--depth-plan", action="store_true", help="edit loop: direct depth placement over the view stream's DepthPlan (the fused pass "
                    "appends every Gaussian to its depth bucket; no bk_hist / bk_scan / bk_scatter, no record gather).  Measured: the ordering "
                    "stages drop from 0.153 to 0.125 ms, the fused pass grows by 0.016 ms and the pipelined loop is 2 %% slower (DESIGN.md "
                    "section 3) - off by default")
    ap.add_argument("--no-work-hint", action="store_true", help="dispatch the blend's tiles by list length instead of by what they cost "
                    "in recent frames (gm_forward_1_geom's work_hint)")
    ap.add_argument("--exchange-batch", type=int, default=None, help="loop steps whose mesh tables are produced (and at N > 1 "
                    "broadcast) together, one batch ahead of their use, on a stream of their own.  Default: 8 at N > 1 (1 = a "
                    "collective per frame), 0 at N = 1 (computed on the frame's own stream)")
    ap.add_argument("--backward-state", action="store_true", help="have the blend also write the per-pixel final transmittance / "
                    "contributor count (the state only a backward pass reads); the edit loop is forward-only and renders "
                    "with GM_FWD_IMAGE_ONLY by default")
    ap.add_argument("--no-discard-region", action="store_true", help="time the FIRST region of K steps behind the W warm-up steps (rounds 1-4) "
                    "instead of running one discarded region of K steps in front of the timed one")
    ap.add_argument("--no-variants", dest="variants", action="store_false", help="N = 1: leave out the two extra regions that time the loop with "
                    "the backward state written / with [N,3,3] rest covariances (`variants` of the bench line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fwd-bwd", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="leave out the C5 legs (the 3 M-Gaussian 4K training loop: the reference's first 1000 "
                    "iterations with their schedule, and --c5-iters iterations at a fixed topology)")
    ap.add_argument("--c5-iters", type=int, default=200)
    ap.add_argument("--c5-scene", default="r05", choices=["r05", "r04"], help="--config c5: r05 = the teacher has features the student misplaced (densify_and_prune "
                    "splits ~1000 rows at 600 / 800 / 1000), r04 = the scene of rounds 1-4 (one row)")
    ap.add_argument("--c5-fixed", action="store_true", help="--config c5: time the fixed-topology SH-degree-3 leg of rounds 1-3 instead of the "
                    "reference's first 1000 iterations with their schedule")
    args = ap.parse_args()
    if args.config == "c5":
        return run_c5(args)

    import torch
    import torch.distributed as dist
    from gaussianmesh_amd import _lib, multiview, scenes
    from gaussianmesh_amd import rasterizer as Rz
    from gaussianmesh_amd.deform import deform_shade_packed, mesh_rs_packed, mesh_rs_packed_batch, pack_mesh_state, vertex_face_adjacency

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    # one block of host cores per rank (before the HIP runtime and RCCL start their threads); GM_RANK_AFFINITY=0 leaves it to the OS
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    pinned = multiview.pin_rank_to_cores(local_rank, local_world)
    if pinned is not None:
        torch.set_num_threads(max(1, min(4, len(pinned))))
    image_only = not args.backward_state
    # GM_BENCH_SHARE_DEVICE=1 + GM_BENCH_BACKEND=gloo: functional smoke test of the N>1 path on a one-GPU box
    # (RCCL refuses two ranks on one device); never used for measurements.
    if os.environ.get("GM_BENCH_SHARE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("GM_BENCH_BACKEND", "nccl")
    have_gpu = torch.cuda.is_available()
    if have_gpu:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_timeout = float(os.environ.get("GM_BENCH_DIST_TIMEOUT", "120"))
    rccl_log = None
    if world > 1:
        from datetime import timedelta
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL's own warnings of this rank in a file of its own (its tail goes into the error line of a failed start-up)
        log_dir = args.check_dir or os.path.join("/tmp", "gm_bench_rccl_%s" % os.environ.get("MASTER_PORT", "0"))
        os.makedirs(log_dir, exist_ok=True)
        rccl_log = os.path.join(log_dir, "rccl_rank%d.log" % rank)
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("NCCL_DEBUG_FILE", rccl_log)
        with DistStage("rccl_init" if backend == "nccl" else backend + "_init", rank, world, dist_timeout + 30, rccl_log):
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=timedelta(seconds=dist_timeout))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world, timeout=timedelta(seconds=dist_timeout))
    assert have_gpu, "bench.py needs a GPU (the HIP path has no CPU fallback)"
    lib = _lib.lib()
    if args.policy is not None:
        Rz.set_default_emission_policy(args.policy)

    P, W, H, F = args.gaussians, args.width, args.height, args.cameras
    # ---- scene: rank 0 generates, everyone else receives it over RCCL (one-time broadcast of the shared cloud)
    shapes = dict(tri=((P, 3), torch.int32), weights=((P, 3), torch.float32), pos=((P, 3), torch.float32),
                  cov=((P, 3, 3), torch.float32), opac=((P, 1), torch.float32), shs=((P, 16, 3), torch.float32),
                  scales=((P, 3), torch.float32), rots=((P, 4), torch.float32), verts=((7500, 3), torch.float32),
                  faces=((15000, 3), torch.int32),
                  mesh=((F, 7500, 21), torch.float32))
    g = {}
    if rank == 0:
        host = build_scene(P, W, H, F)
        for k, (shp, dt) in shapes.items():
            g[k] = torch.tensor(host[k], dtype=dt, device=dev).reshape(shp).contiguous()
    else:
        for k, (shp, dt) in shapes.items():
            g[k] = torch.empty(shp, dtype=dt, device=dev)
    # the animation ("mesh") stays on rank 0; its frames are broadcast one at a time inside the timed loop
    n_ranks_seen = 1
    if world > 1:
        # the first collectives: the cloud (0.3 GB in eleven broadcasts) and an all-reduce of ones - proof in the bench line that the
        # backend saw all N ranks.  Completed (device synchronised) inside the guarded stage: RCCL reports asynchronously
        with DistStage("first_broadcast", rank, world, dist_timeout + 30, rccl_log):
            multiview.broadcast_cloud({k: v for k, v in g.items() if k != "mesh"}, src=0)
            n_ranks_seen = int(round(multiview.sum_over_ranks(1.0, dev)))
            torch.cuda.synchronize()
            if n_ranks_seen != world:
                raise RuntimeError("all-reduce of ones over %d ranks returned %d" % (world, n_ranks_seen))
    # the rest covariances as their six distinct entries where every matrix is symmetric bit for bit (one-time, like the mesh tables'
    # packing): the fused pass then reads 24 instead of 36 bytes per Gaussian and computes what it computes from [N,3,3]
    from gaussianmesh_amd.deform import pack_cov6
    g["cov_in"] = (None if args.no_cov6 else pack_cov6(g["cov"]))
    if g["cov_in"] is None:
        g["cov_in"] = g["cov"]
    Vm = g["verts"].shape[0]
    # per-frame ARAP-style deformation: (R, S) of every vertex come from the deformed mesh of that frame (gm_mesh_rs,
    # the device counterpart of pyACAP.GetRS), on the rank that owns the animation
    off, adj = vertex_face_adjacency(g["faces"], Vm)
    adjacency = (torch.tensor(off, device=dev), torch.tensor(adj, device=dev))
    v1_frames = g["mesh"][:, :, 0:3].contiguous()          # deformed vertex positions per animation frame
    cams = [scenes.orbit_camera(k, F, W, H) for k in range(F)]
    cam_t = [dict(view=torch.tensor(c["view"], device=dev), proj=torch.tensor(c["proj"], device=dev),
                  campos=torch.tensor(c["campos"], device=dev), tanx=c["tanx"], tany=c["tany"]) for c in cams]
    bg = torch.ones(3, device=dev)              # edit tool renders on white (edittool/__init__.py:410)
    stats = {}
    nstreams = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    lag = max(1, args.status_lag)
    ahead = max(1, args.begin_ahead)
    KB = max(1, min(args.frames_per_launch, 8))
    if args.unfused or args.depth_plan or args.exact_count or args.analytic_rs:
        KB = 1                                   # (those options are single-frame paths)
    nws = nstreams + ahead + lag                # frames i+1 .. i+ahead are begun before frame i is completed, and frame i's status is read `lag` frames later
    if KB > 1:
        lag = max(lag, 2 * KB)
        nws = KB * (nstreams + 1) + lag         # a workspace is reused only after its frame's status has been read
    workspaces = [Rz.RasterWorkspace(growth=1.5) for _ in range(nws)]
    torch.cuda.synchronize()

    # one work-hint buffer for the view stream of this rank: consecutive frames are neighbouring cameras of the orbit
    hint = None if args.no_work_hint else Rz.new_work_hint(W, H, dev)
    # ... and one DepthPlan: from the second frame on the fused pass places the Gaussians in their depth buckets itself
    dplan = Rz.new_depth_plan(dev) if (args.depth_plan and not args.unfused) else None
    pending = {}
    unchecked = []
    views_walked = []                            # camera index of every loop step this rank issued (--check-dir)
    stats["overflows"] = 0

    batching = [False]                           # batches only once the sizing passes have taught the workspaces their capacity
    batch_steps = []                             # loop steps collected for the next batch
    nbatch = [0]
    table_bufs = [torch.empty((KB, Vm, 24), dtype=torch.float32, device=dev) for _ in range(max(nstreams + 2, 4))] if KB > 1 else None

    one_stream = [None]                          # per-stage timing pass: every batch on this stream

    def launch_batch():
        """K collected loop steps as ONE launch chain on stream (batch index) % streams: gm_mesh_rs_packed_batch (K tables) +
        gm_forward_deformed_batch_async (arm, fused pass over the static cloud looping over the K (table, camera) pairs, depth order, emission,
        tile pass, blend: every stage one launch with grid z = frame)."""
        steps, b = list(batch_steps), nbatch[0]
        del batch_steps[:]
        nbatch[0] += 1
        torch.cuda.set_stream(one_stream[0] if one_stream[0] is not None else streams[b % nstreams])
        src = [pipe.frame(i) for i in steps] if pipe is not None else [v1_frames[i % F] for i in steps]
        tables = mesh_rs_packed_batch(g["verts"], src, g["faces"], adjacency, out=table_bufs[b % len(table_bufs)][:len(steps)])
        cams_k = []
        for i in steps:
            vi = multiview.view_for_step(i, F, rank, world)
            views_walked.append((i, vi))
            cams_k.append(cam_t[vi])
        hs = Rz.forward_deformed_batch(bg, g["tri"], g["weights"], tables, g["cov_in"], g["pos"], g["shs"], g["opac"], cams_k, H, W, 3,
                                       [workspaces[i % nws] for i in steps], image_only=image_only, work_hint=hint)
        if timeline is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(hs[0].stream)
            timeline.append((time.perf_counter(), ev))
        stats["last_image"] = hs[-1].color
        unchecked.extend(hs)
        while len(unchecked) > lag:
            verify(unchecked.pop(0))
        return hs[-1].color

    def step(i):
        """Issue frame i's first half (deform + colour + preprocess + depth order) on stream i % nstreams, THEN complete frame
        i - ahead (--begin-ahead, default 2).  Default: sync-free completion - the instance count stays on the device (binning
        buffer at the workspace's capacity, learned during warm-up), the host only reads the status words of frames completed
        `lag` steps ago, which landed long before.
        --exact-count: the host waits for the completed frame's count (one 4-byte read-back, hidden behind frame i's first half)."""
        if KB > 1 and batching[0]:
            batch_steps.append(i)
            return launch_batch() if len(batch_steps) == KB else None
        torch.cuda.set_stream(streams[i % nstreams])     # (not `with torch.cuda.stream(...)`: entering and leaving the context costs the
        if timeline is not None:
            t_a = time.perf_counter()
        pending[i] = step_on_stream(i, workspaces[i % nws], begin_only=True)   # host ~20 us per frame; drain() restores the default stream)
        if timeline is not None:
            host_split.append((time.perf_counter() - t_a, 0.0))
            t_a = time.perf_counter()
        prev = pending.pop(i - ahead, None)
        out = finish(prev) if prev is not None else None
        if timeline is not None:
            host_split[-1] = (host_split[-1][0], time.perf_counter() - t_a)
        return out

    def verify(h):
        ok, nr = h.check()
        if not ok:                              # instance count outgrew the binning capacity: render that frame again, exactly
            stats["overflows"] += 1
            nr = h.finish(image_only=image_only, work_hint=hint)[0]
        stats["R"] = nr
        stats["radii"] = h.radii

    timeline = [] if os.environ.get("GM_BENCH_TIMELINE") else None      # tools/region_timeline.py: an event behind every frame
    host_split = []                                                      # ... and the host time of each step's begin / finish halves

    def finish(h):
        out = h.finish(sync_free=not args.exact_count, image_only=image_only, work_hint=hint)
        if timeline is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(h.stream)
            timeline.append((time.perf_counter(), ev))
        stats["last_image"] = out[1]
        unchecked.append(h)
        while len(unchecked) > lag:
            verify(unchecked.pop(0))
        return out[1]

    default_stream = torch.cuda.default_stream(dev)

    def drain():
        if batch_steps:
            launch_batch()
        for k in sorted(pending):
            finish(pending.pop(k))
        torch.cuda.set_stream(default_stream)
        while unchecked:
            verify(unchecked.pop(0))

    def frame_table(t, out=None):                # per-vertex gather table [Vm,24] of animation frame t: dV | R | S
        if args.analytic_rs:
            tab = pack_mesh_state(g["mesh"][t], g["verts"])         # [Vm,21] frame state -> table (one small kernel)
            return tab if out is None else out.copy_(tab.view(out.shape))
        return mesh_rs_packed(g["verts"], v1_frames[t], g["faces"], adjacency, out=out)

    # N > 1, the real exchange step: rank 0 owns the animation.  What travels is the DEFORMED VERTEX POSITIONS of every loop step
    # (Vm x 12 B = 90 KB), --exchange-batch consecutive steps in one RCCL broadcast, one batch ahead of their use, on the pipe's
    # own stream (multiview.MeshStatePipe) - the render streams never wait on a collective - and EVERY rank turns them into the
    # per-vertex (R, S) gather table with gm_mesh_rs_packed on the frame's own stream, exactly as at N = 1: no rank runs a
    # kernel the others do not (round 2 produced the 720-KB tables on rank 0 only, which made rank 0 the slowest rank of a
    # max-over-ranks timing).  --analytic-rs (precomputed (R, S) that only rank 0 holds) still ships whole tables.
    # N = 1: no pipe; the positions are read where they lie.
    pipe = None
    batch = args.exchange_batch if args.exchange_batch is not None else (8 if world > 1 else 0)
    if world > 1 or batch > 0:
        if args.analytic_rs:
            pipe = multiview.MeshStatePipe(lambda i, out: frame_table(i % F, out), (Vm, 24), max(1, batch), dev, src=0,
                                           frames_in_flight=nws)
        else:
            def produce_positions(b, buf):           # one gather launch per batch on the pipe's stream (rank 0 only)
                idx = (torch.arange(buf.shape[0], device=dev) + b * buf.shape[0]) % F
                torch.index_select(v1_frames, 0, idx, out=buf)
            pipe = multiview.MeshStatePipe(None, (Vm, 3), max(1, batch), dev, src=0, frames_in_flight=nws,
                                           produce_batch=produce_positions)
    exchange_bytes = Vm * (96 if args.analytic_rs else 12)

    def table_for_step(i, exchange):
        if pipe is None or not exchange:
            return frame_table(i % F)
        got = pipe.frame(i)
        return got if args.analytic_rs else mesh_rs_packed(g["verts"], got, g["faces"], adjacency)

    def step_on_stream(i, workspace, exchange=True, begin_only=False):
        t = i % F
        packed = table_for_step(i, exchange)
        vi = multiview.view_for_step(i, F, rank, world)
        if begin_only:
            views_walked.append((i, vi))
        c = cam_t[vi]
        if begin_only and not args.unfused:      # one enqueue: deform + colour + preprocess + depth sort + instance count
            return Rz.forward_deformed_begin(bg, g["tri"], g["weights"], packed, g["cov_in"], g["pos"], g["shs"], g["opac"], c["view"],
                                             c["proj"], c["tanx"], c["tany"], H, W, 3, c["campos"], False, workspace=workspace,
                                             want_count=args.exact_count, depth_plan=dplan)
        if not args.unfused:                     # same path, completed at once (per-stage timing pass)
            nr, color, radii, _, _, _ = Rz.forward_deformed_begin(bg, g["tri"], g["weights"], packed, g["cov_in"], g["pos"], g["shs"], g["opac"],
                                                                  c["view"], c["proj"], c["tanx"], c["tany"], H, W, 3, c["campos"], False,
                                                                  workspace=workspace, depth_plan=dplan).finish(image_only=image_only, work_hint=hint)
            stats["R"] = nr
            stats["radii"] = radii
            return color
        pos, cov6, rgb = deform_shade_packed(g["tri"], g["weights"], packed, g["cov"], g["pos"], g["shs"], c["campos"], deg=3)
        if begin_only:
            return Rz.rasterize_forward_begin(bg, pos, rgb, g["opac"], None, None, 1.0, cov6, c["view"], c["proj"], c["tanx"],
                                              c["tany"], H, W, None, 3, c["campos"], False, False, workspace=workspace)
        nr, color, radii, _, _, _ = Rz.rasterize_forward(bg, pos, rgb, g["opac"], None, None, 1.0, cov6, c["view"], c["proj"],
                                                         c["tanx"], c["tany"], H, W, None, 3, c["campos"], False, False,
                                                         workspace=workspace)
        stats["R"] = nr
        stats["radii"] = radii
        return color

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0 and not args.unfused and not os.environ.get("GM_DEBUG_STOP_AFTER"):   # (tools/stage_marginal.sh: truncated frames)
        # the timed path (fused deform + colour + preprocess, async halves) must render what the unfused chain
        # (gm_deform_shade_packed -> gm_forward_0/1) renders for the same frame: bit-identical image and radii
        c = cam_t[multiview.view_for_step(0, F, rank, world)]
        pk = pack_mesh_state(g["mesh"][0], g["verts"])
        nr_f, col_f, rad_f, *_ = Rz.forward_deformed_begin(bg, g["tri"], g["weights"], pk, g["cov_in"], g["pos"], g["shs"], g["opac"], c["view"],
                                                           c["proj"], c["tanx"], c["tany"], H, W, 3, c["campos"], False).finish(image_only=image_only)
        pos_u, cov6_u, rgb_u = deform_shade_packed(g["tri"], g["weights"], pk, g["cov"], g["pos"], g["shs"], c["campos"], deg=3)
        nr_u, col_u, rad_u, *_ = Rz.rasterize_forward(bg, pos_u, rgb_u, g["opac"], None, None, 1.0, cov6_u, c["view"], c["proj"], c["tanx"],
                                                      c["tany"], H, W, None, 3, c["campos"], False, False)
        torch.cuda.synchronize()
        assert nr_f == nr_u and torch.equal(rad_f, rad_u) and torch.equal(col_f, col_u), "fused frame differs from the unfused chain"
        del pos_u, cov6_u, rgb_u, col_u, col_f
    # setup, not part of W: passes over the camera orbit size every workspace's binning buffer for the largest instance count
    # of the trajectory (the sync-free forward renders into a buffer of fixed capacity; a frame that outgrows it is redone).
    # Three passes (~45 ms of device work) rather than one: a timed region that starts a few milliseconds after the device
    # leaves idle runs ~3 % slower than the same frames a little later (20-step regions, same cameras: 4010 frames/s after
    # one pass, 4130 after three, no further gain from eight).
    # the cyclic garbage collector stays out of the timed regions (a generation-2 pass over the scene's objects is tens of
    # milliseconds - 10 % of a 300-step region - whenever it happens to fall into one); reference counting still frees every
    # frame.  Collected BEFORE the sizing passes: the device must not sit idle right in front of the timed region.
    import gc
    gc.collect()
    gc.disable()
    for i in range(-3 * F, 0):
        step(i)
    if KB > 1:                                   # every workspace at the stream's largest capacity (a batch's binning buffers share one layout), then batches
        drain()
        cap = max(ws_.capacity for ws_ in workspaces)
        for ws_ in workspaces:
            ws_.capacity = cap
        batching[0] = True
    for i in range(args.warmup):
        step(i)
    drain()
    def timed_region(first):
        """K steps bracketed by barrier + synchronize on both sides; seconds (this rank)."""
        barrier()
        if timeline is not None:
            del timeline[:]
            del host_split[:]
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(streams[first % nstreams])
        t = time.perf_counter()
        for i in range(args.steps):
            step(first + i)
        drain()                                  # the last frame of the timed region is completed inside it
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - t
        if timeline is not None:
            sys.stderr.write("region from step %d: %.3f ms; frame completions (device ms after the region opened | host ms when issued): %s\n" % (
                first, 1e3 * el, " ".join("%.2f|%.2f" % (ev0.elapsed_time(e), 1e3 * (th - t)) for th, e in timeline)))
            sys.stderr.write("   host us per step (begin / finish+verify): %s\n" % " ".join("%.0f/%.0f" % (1e6 * a, 1e6 * b) for a, b in host_split))
        return el

    # One DISCARDED region of the same K steps in front of the timed one (round 5, asked for by the round-4 review): the first
    # region after the long device-bound setup runs 3-6 % slower than every region behind it (profiles/r04_region_timeline.txt: the
    # host needs 100-170 us per frame instead of 70 for its first ten to fifteen frames, and steps W .. W+K-1 are the orbit's
    # heaviest stretch), and that first region was the one the driver scored.  It is warm-up, bracketed by the same barriers, and
    # reported (config.discarded_region_frames_per_s); --no-discard-region restores the old behaviour.
    first = args.warmup
    discarded_fps = None
    if not args.no_discard_region:
        discarded_fps = world * args.steps / multiview.max_over_ranks(timed_region(first), dev)
        first += args.steps
    elapsed = timed_region(first)
    per_rank_s = multiview.gather_over_ranks(elapsed, dev)       # every rank's own clock around the same barriers
    elapsed = multiview.max_over_ranks(elapsed, dev)
    fps = world * args.steps / elapsed
    repeats = []
    variants = {}
    single_stream_ms = single_stream_sf_ms = None
    if world == 1:
        nxt = first + args.steps
        for _ in range(max(0, args.repeats)):    # the same region again: run-to-run spread of the pipelined loop
            repeats.append(args.steps / timed_region(nxt))
            nxt += args.steps
        # the same loop in the two configurations the headline does not use (both disclosed in `config`): with the per-pixel backward
        # state written as the reference's forward always does (forward.cu:369-370), and with the rest covariances handed over as
        # [N,3,3] (the reference's deform_gaussian signature) instead of the six distinct entries
        if args.variants and not args.unfused and dplan is None:
            keep_io, keep_cov = image_only, g["cov_in"]
            if image_only:
                image_only = False
                timed_region(nxt); nxt += args.steps                    # (the workspaces' image buffers grow: one untimed pass)
                variants["backward_state"] = args.steps / timed_region(nxt); nxt += args.steps
                image_only = keep_io
            if g["cov_in"] is not g["cov"]:
                g["cov_in"] = g["cov"]
                timed_region(nxt); nxt += args.steps
                variants["cov9"] = args.steps / timed_region(nxt); nxt += args.steps
                g["cov_in"] = keep_cov
            timed_region(nxt); nxt += args.steps
        # latency of one frame: the same frames one after the other on one stream, each completed before the next begins
        nlat = min(args.steps, 100)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        with torch.cuda.stream(streams[0]):
            for i in range(nlat):
                step_on_stream(nxt + i, workspaces[0], exchange=False)
                streams[0].synchronize()
        torch.cuda.synchronize()
        single_stream_ms = 1e3 * (time.perf_counter() - t1) / nlat
        # the same with the frame enqueued in ONE go (capacity known from the frames before, count left on the device, status words read
        # after the image): what an interactive caller that renders frame after frame of one scene waits for
        single_stream_sf_ms = None
        if not args.unfused and dplan is None:
            ws0 = workspaces[0]
            t1 = time.perf_counter()
            with torch.cuda.stream(streams[0]):
                for i in range(nlat):
                    k = nxt + i
                    c = cam_t[multiview.view_for_step(k, F, rank, world)]
                    h = Rz.forward_deformed_begin(bg, g["tri"], g["weights"], table_for_step(k, False), g["cov_in"], g["pos"], g["shs"], g["opac"], c["view"],
                                                  c["proj"], c["tanx"], c["tany"], H, W, 3, c["campos"], False, workspace=ws0, want_count=False)
                    h.finish(sync_free=True, image_only=image_only, work_hint=hint)
                    if not h.check()[0]:
                        h.finish(image_only=image_only)
                    streams[0].synchronize()
            torch.cuda.synchronize()
            single_stream_sf_ms = 1e3 * (time.perf_counter() - t1) / nlat
    gc.enable()
    if args.check_dir:
        last = first + args.steps - 1
        os.makedirs(args.check_dir, exist_ok=True)
        np.savez(os.path.join(args.check_dir, "rank%d.npz" % rank), step=last, frame=last % F, view=multiview.view_for_step(last, F, rank, world),
                 image=stats["last_image"].cpu().numpy(), overflows=stats["overflows"],
                 views=np.array([v for (k, v) in views_walked if first <= k < first + args.steps], np.int64))   # timed steps only

    out = {
        "metric": "frames/sec (fwd), 1M Gaussians @1080p, deform+render", "value": fps, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C3: %d Gaussians bound to 15k-face torus, per-frame mesh deform + SH colour + forward "
                               "render %dx%d, %d-camera orbit, views sharded by rank" % (P, W, H, F),
                   "gaussians": P, "width": W, "height": H, "sh_degree": 3, "views_per_step_per_gpu": 1, "frames_per_launch": KB,
                   "vertex_rs": "analytic tables" if args.analytic_rs else "gm_mesh_rs per frame", "hip_streams": nstreams,
                   "exchange": None if pipe is None else {"n_ranks_seen": n_ranks_seen, "backend": backend if world > 1 else None, "steps_per_broadcast": pipe.batch, "bytes_per_step": exchange_bytes, "broadcasts": pipe.broadcasts,
                                                          "payload": "per-vertex (R, S) tables" if args.analytic_rs else "deformed vertex positions; (R, S) by gm_mesh_rs on every rank",
                                                          # a slow rank is visible here: each rank's own time for the region / steps
                                                          "ms_per_step_by_rank": [round(1e3 * t / args.steps, 4) for t in per_rank_s],
                                                          "slowest_over_fastest_rank": round(max(per_rank_s) / max(min(per_rank_s), 1e-12), 4),
                                                          "cores_per_rank": None if pinned is None else len(pinned)},
                   "emission_policy": Rz.get_default_emission_policy(W, H), "image_only": image_only, "work_hint": hint is not None,
                   "cov6": bool(g["cov_in"].dim() == 2 and g["cov_in"].shape[-1] == 6), "depth_plan": dplan is not None, "frames_refused_by_direct_placement": None if dplan is None else dplan.refused,
                   "frames_redone": stats["overflows"],          # sync-free frames that outgrew their binning buffer (rendered again, exactly)
                   "parallelism": "views x%d" % world},
    }
    out["config"]["discarded_region_frames_per_s"] = None if discarded_fps is None else round(discarded_fps, 1)
    out["config"]["runtime"] = RUNTIME
    if variants:
        out["variants"] = {k: round(v, 1) for k, v in variants.items()}
        out["variants"]["note"] = ("frames/s of the same pipelined loop with the per-pixel final_T / n_contrib written (backward_state) and with "
                                   "[N,3,3] rest covariances instead of the packed [N,6] (cov9); `value` is image-only + cov6")
    if repeats:
        out["repeats"] = {"frames_per_s": [round(x, 1) for x in repeats], "median": float(np.median(repeats + [fps]))}
    if single_stream_ms is not None:
        out["single_stream"] = {"ms_per_frame": single_stream_ms, "frames_per_s": 1e3 / single_stream_ms,
                                "note": "frame latency: one frame at a time on one stream, instance count read back by the host, stream drained after every frame"}
        if single_stream_sf_ms is not None:
            out["single_stream"]["sync_free_ms_per_frame"] = single_stream_sf_ms
            out["single_stream"]["sync_free_note"] = ("the same frames enqueued in one go each (binning capacity known from the frames before, instance count left on "
                                                      "the device, status words read after the image), stream drained after every frame")

    if rank == 0:
        # ---- per-stage HIP-event timing over a second pass of the same steps (events perturb the pipelining a
        # little, so they are kept out of the region that defines `value`)
        # ... and on ONE stream, so a stage's events bracket only its own kernels
        lib.gm_profile_reset(); lib.gm_profile_enable(1)
        nprof = min(args.steps, 50)
        gx16, gy16 = (W + 15) // 16, (H + 15) // 16
        psh = max(Rz.get_default_emission_policy(W, H) - 1, 0)                   # lists per 2^psh x 2^psh tiles
        list_tiles = ((gx16 + (1 << psh) - 1) >> psh) * ((gy16 + (1 << psh) - 1) >> psh)
        with torch.cuda.stream(streams[0]):
            for i in range(nprof):
                step_on_stream(args.warmup + i, workspaces[0], exchange=False)   # rank 0 only: no collective here
        torch.cuda.synchronize()
        lib.gm_profile_enable(0)
        import ctypes as C
        Rn = int(stats["R"]); V = int((stats["radii"] > 0).sum().item())
        per = {}
        for s in STAGES:
            ms = C.c_double(0); n = C.c_int64(0)
            lib.gm_profile_read(s.encode(), C.byref(ms), C.byref(n))
            if n.value:
                per[s] = ms.value / nprof               # ms per frame (a stage may be several launches)
        direct = dplan is not None
        bytes_key = lambda st: (("deform_pre_direct" if direct else "deform_pre") if (st == "deform" and not args.unfused) else
                                ("depth_sort_direct" if (st == "depth_sort" and direct) else st))
        stage_bytes = lambda st: algorithmic_bytes(bytes_key(st), P, V, Rn, W, H, Vm, list_tiles=list_tiles,
                                                   cov_bytes=24 if g["cov_in"].shape[-1] == 6 and g["cov_in"].dim() == 2 else 36)
        # `roofline` = the dominant KERNEL of the frame: the longest of the stages that are one launch each (the ordering stages are
        # 4 + 3 launches of at most 30 us each; they are under stage_roofline)
        dom = max((st for st in per if st in KERNEL_OF_STAGE), key=per.get)
        ab = stage_bytes(dom)
        ach = ab / (per[dom] * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": dom, "kernel_name": KERNEL_OF_STAGE[dom], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": measured_traffic(dom, P, W, H),
                           "traffic_source": "static: profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this "
                                             "workload; counters cannot be read inside the timed run)",
                           "algorithmic_bytes": ab, "avg_ms": per[dom],
                           "note": ("the forward blend: bound by the vector ALU, not by HBM (DESIGN.md section 4: 3.4e7 vector instructions per launch "
                                    "keep the chip's 1024 SIMDs busy for 59 us, SQ_ACTIVE_INST_VALU); the HBM figure is reported because the contract asks for it")
                                   if dom == "render" else
                                   ("the fused deformation / SH colour / preprocess kernel: a streaming kernel bound by HBM") if dom == "deform" else
                                   "longest single kernel of the frame"}
        # the ceiling the dominant kernel actually runs against when it is the blend: the vector ALU.  Round 5 measured what a wave64
        # vector instruction costs its SIMD by class (tools/valu_probe: 2.3 cycles for fma / mul / add / logic, 4.1-4.3 for min / max /
        # compare / select / packed f32 / SGPR operands, 8.1 for exp / rcp) and the kernel's own SQ_ACTIVE_INST_VALU (quad-cycles the
        # vector ALU of a SIMD is occupied, summed over the chip): busy time = 4 x that / (1024 SIMDs x 2.4 GHz).  Static, from the PMC
        # pass of the same workload (profiles/inst_mix.json <- profiles/r05_blend_sq_pmc.txt), like `traffic`
        nv = measured_valu(dom, P, W, H)
        if nv:
            out["roofline"]["valu_issue"] = valu_ceiling(dom, nv, per[dom])
        # every stage, same definition (algorithmic bytes of the stage / its HIP-event time); "deform" is the fused kernel alone
        out["stage_roofline"] = {st: round(stage_bytes(st) / (per[st] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                 for st in per if st in ("mesh_rs", "deform", "depth_sort", "duplicate", "tile_sort", "render")}
        out["stage_ms"] = {k: round(v, 4) for k, v in per.items()}
        out["scene"] = {"P": P, "V": V, "R": Rn}
        frame_valu = [measured_valu(s, P, W, H) for s in per]
        tot_bytes = sum(stage_bytes(s) for s in per)
        out["frame_roofline"] = {"algorithmic_bytes": tot_bytes, "achieved": tot_bytes / (elapsed / args.steps) / 1e9,
                                 "frac": tot_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s",
                                 "single_stream_ms_per_frame": sum(per.values())}
        if KB > 1:
            # the batched loop's own stage times (HIP events around each launch of K frames, every batch on ONE stream so that a stage's events
            # bracket only its kernels) and bytes: the fused pass reads the static cloud once per K frames
            lib.gm_profile_reset(); lib.gm_profile_enable(1)
            one_stream[0] = streams[0]
            nb = 10
            pipe_keep, pipe = pipe, None                                          # rank 0 only: no collective here
            for i in range(nb * KB):
                step(args.warmup + i)
            drain()
            torch.cuda.synchronize()
            pipe = pipe_keep
            one_stream[0] = None
            lib.gm_profile_enable(0)
            perb = {}
            for st in STAGES:
                ms = C.c_double(0); n = C.c_int64(0)
                lib.gm_profile_read(st.encode(), C.byref(ms), C.byref(n))
                if n.value:
                    perb[st] = ms.value / (nb * KB)                                 # ms per FRAME
            bbytes = lambda st: algorithmic_bytes("deform_pre_batch" if st == "deform" else bytes_key(st), P, V, Rn, W, H, Vm, list_tiles=list_tiles, batch=KB,
                                                  cov_bytes=24 if g["cov_in"].shape[-1] == 6 and g["cov_in"].dim() == 2 else 36)
            tot_b = sum(bbytes(st) for st in perb)
            # `roofline` = the dominant kernel AS THE TIMED LOOP LAUNCHES IT: one launch over KB frames (the single-frame launch stays beside it)
            domb = max((st for st in perb if st in KERNEL_OF_STAGE), key=perb.get)
            single = out["roofline"]
            abb = bbytes(domb) * KB
            achb = abb / (perb[domb] * KB * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": domb,
                               "kernel_name": ("gm::deform_shade_pre_batch_kernel" if domb == "deform" else KERNEL_OF_STAGE[domb]) + " (grid z = %d: %d frames per launch)" % (KB, KB),
                               "achieved": achb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achb / HBM_PEAK_GBS, "traffic": measured_traffic(domb, P, W, H, batch=KB),
                               "traffic_source": "static: profiles/hbm_traffic_batch.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the batched loop, per launch)",
                               "algorithmic_bytes": abb, "avg_ms": perb[domb] * KB, "frames_per_launch": KB, "note": single.get("note"),
                               "single_frame_launch": {k: single[k] for k in ("kernel", "kernel_name", "achieved", "frac", "traffic", "algorithmic_bytes", "avg_ms") if k in single}}
            nvb_ = measured_valu(domb, P, W, H)
            if nvb_ and domb != "deform":           # (the per-frame instruction count of the blend does not depend on how its frames are launched)
                out["roofline"]["valu_issue"] = valu_ceiling(domb, nvb_ * KB, perb[domb] * KB, frames=KB)
            out["batch"] = {"frames_per_launch": KB, "launches_per_batch": 12,
                            "stage_ms_per_frame": {k: round(v, 4) for k, v in perb.items()}, "kernel_ms_per_frame_one_stream": round(sum(perb.values()), 4),
                            "stage_roofline": {st: round(bbytes(st) / (perb[st] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for st in perb if st in ("mesh_rs", "deform", "depth_sort", "duplicate", "tile_sort", "render")},
                            "algorithmic_bytes_per_frame": tot_b, "fused_pass_bytes_per_frame": bbytes("deform"),
                            "frame_roofline_frac": tot_b / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                            "note": "frame_roofline.frac prices the loop against the 677 MB a frame moves when rendered alone; this fraction against what a frame of a "
                                    "batch moves (the static cloud once per %d frames) - frames/s is the metric, the second fraction falls by construction" % KB}
        if all(frame_valu):                      # the frame's other ceiling: its vector instructions at the blend's measured mix average (4.2 cycles; the streaming kernels' mix is cheaper: an upper bound)
            out["frame_roofline"]["valu_issue_ms"] = sum(frame_valu) * 4.2 / (1024 * 2.4e9) * 1e3
            out["frame_roofline"]["valu_issue_frac"] = out["frame_roofline"]["valu_issue_ms"] / (1e3 * elapsed / args.steps)

    if rank == 0 and world == 1 and not args.no_fwd_bwd:
        # ---- forward + backward through the autograd operator (train-time input mode: SH + scale/rot)
        from gaussianmesh_amd import GaussianRasterizer, GaussianRasterizationSettings
        c = cam_t[0]
        rs = GaussianRasterizationSettings(H, W, c["tanx"], c["tany"], torch.zeros(3, device=dev), 1.0, c["view"], c["proj"], 3,
                                           c["campos"], False, False, None if args.no_work_hint else Rz.new_work_hint(W, H, dev))
        leaves = [g[k].clone().requires_grad_(True) for k in ("pos", "opac", "shs", "scales", "rots")]
        m2d = torch.zeros_like(leaves[0], requires_grad=True)
        wgt = torch.randn((3, H, W), device=dev)
        rast = GaussianRasterizer(rs)

        def it():
            for l in leaves + [m2d]:
                l.grad = None
            color, _ = rast(leaves[0], m2d, leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
            (color * wgt).sum().backward()
        for _ in range(10):
            it()
        torch.cuda.synchronize()
        nit = 100
        t1 = time.perf_counter()
        for _ in range(nit):
            it()
        torch.cuda.synchronize()
        out["fwd_bwd"] = {"ms_per_iter": 1e3 * (time.perf_counter() - t1) / nit, "iters": nit,
                          "config": "%d Gaussians, %dx%d, SH3 + scale/rot inputs, loss = sum(w*image)" % (P, W, H)}
        # roofline of the other half of the metric: per-stage HIP events (torch's current stream IS the stream the operator
        # launches on) over a second pass of the same iteration; algorithmic bytes of THIS algorithm for forward (training
        # input mode, backward state written) + backward (SURVEY.md 8d A_fwd(training) + A_bwd restated for the record layout
        # of DESIGN.md section 3); dominant kernel = the backward blend
        lib.gm_profile_reset(); lib.gm_profile_enable(1)
        nprof_fb = 20
        for _ in range(nprof_fb):
            it()
        torch.cuda.synchronize()
        lib.gm_profile_enable(0)
        per_fb = {}
        for st_name in ("preprocess", "depth_sort", "duplicate", "tile_sort", "ranges", "render", "render_bwd", "preprocess_bwd"):
            ms = C.c_double(0); n = C.c_int64(0)
            lib.gm_profile_read(st_name.encode(), C.byref(ms), C.byref(n))
            if n.value:
                per_fb[st_name] = ms.value / nprof_fb
        st_img = torch.zeros((4,), dtype=torch.int32).pin_memory()
        with torch.no_grad():
            nr_fb, _, rad_fb, *_ = Rz.rasterize_forward(rs.bg, g["pos"], None, g["opac"], g["scales"], g["rots"], 1.0, None, c["view"], c["proj"],
                                                        c["tanx"], c["tany"], H, W, g["shs"], 3, c["campos"], False, False)
        Vf = int((rad_fb > 0).sum().item()); Rf = int(nr_fb)
        fb_bytes = {
            "preprocess": P * (12 + 12 + 16 + 4) + Vf * 12 * 16 + Vf * 36 + P * 28 + Vf * 27,     # inputs, SH rows, splat record, radius/count/bin/key, cov3D + clamped
            "depth_sort": algorithmic_bytes("depth_sort", P, Vf, Rf, W, H, Vm, list_tiles=list_tiles),
            "duplicate": algorithmic_bytes("duplicate", P, Vf, Rf, W, H, Vm, list_tiles=list_tiles),
            "tile_sort": algorithmic_bytes("tile_sort", P, Vf, Rf, W, H, Vm, list_tiles=list_tiles),
            "ranges": algorithmic_bytes("ranges", P, Vf, Rf, W, H, Vm, list_tiles=list_tiles),
            "render": Rf * 40 + W * H * (12 + 8),                                                  # + final_T, n_contrib
            "render_bwd": Rf * 40 + W * H * 20 + Vf * 48,                                           # lists + records, dL/dpixel + T + n_contrib, grad_acc record
            "preprocess_bwd": Vf * 507,                                                              # 303 read + 256 written per visible Gaussian, minus the 52 bytes of intermediates (dL/dconic, dL/dcolour, dL/dcov3D) the operator declines
        }
        tot_fb = sum(fb_bytes[k] for k in per_fb)
        sec = out["fwd_bwd"]["ms_per_iter"] * 1e-3
        ach_b = fb_bytes["render_bwd"] / (per_fb["render_bwd"] * 1e-3) / 1e9
        out["fwd_bwd"]["stage_ms"] = {k: round(v, 4) for k, v in per_fb.items()}
        out["fwd_bwd"]["scene"] = {"P": P, "V": Vf, "R": Rf}
        out["fwd_bwd"]["roofline"] = {"bound": "hbm", "kernel": "render_bwd", "achieved": ach_b, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": ach_b / HBM_PEAK_GBS, "algorithmic_bytes": fb_bytes["render_bwd"], "avg_ms": per_fb["render_bwd"],
                                      "traffic": None,
                                      "note": "the backward blend is bound by the vector ALU (DESIGN.md section 4), not by HBM"}
        nvb = measured_valu("render_bwd", P, W, H)
        if nvb:
            out["fwd_bwd"]["roofline"]["valu_issue"] = valu_ceiling("render_bwd", nvb, per_fb["render_bwd"])
        out["fwd_bwd"]["iteration_roofline"] = {"algorithmic_bytes": tot_fb, "achieved": tot_fb / sec / 1e9, "frac": tot_fb / sec / 1e9 / HBM_PEAK_GBS,
                                                "unit": "GB/s", "sum_of_stage_ms": sum(per_fb.values())}
        # the same iteration with the training loop's photometric loss (L1 + SSIM, gm_ssim_fwd/bwd) on the rendered image
        from gaussianmesh_amd.loss import photometric_loss
        gt = torch.rand((3, H, W), device=dev)

        def it_loss():
            for l in leaves + [m2d]:
                l.grad = None
            color, _ = rast(leaves[0], m2d, leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
            photometric_loss(color, gt, 0.2).backward()
        for _ in range(10):
            it_loss()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nit):
            it_loss()
        torch.cuda.synchronize()
        out["fwd_bwd"]["ms_per_iter_with_l1_ssim_loss"] = 1e3 * (time.perf_counter() - t1) / nit
        # a whole optimisation iteration of the mesh-Gaussian training loop on the same cloud (train_mesh_gaussian.py:80-148:
        # mesh-bound positions and activations, render, L1 + SSIM + mesh-restrict loss, backward, Adam on all 7 groups)
        from gaussianmesh_amd.renderer import Camera, MeshBoundGaussians
        from gaussianmesh_amd.train import Trainer
        del leaves, m2d
        tri_np = g["tri"].cpu().numpy(); vn = g["verts"].cpu().numpy()
        v1, v2, v3 = (torch.tensor(vn[tri_np[:, k]], device=dev) for k in range(3))
        nrm = torch.nn.functional.normalize(torch.cross(v2 - v1, v3 - v1, dim=1), dim=1)
        rad = (((v2 - v1).norm(dim=1) + (v3 - v2).norm(dim=1) + (v1 - v3).norm(dim=1)) / 3)[:, None]
        model = MeshBoundGaussians(torch.log(g["weights"].clamp_min(1e-6)), torch.zeros((P, 1), device=dev), g["shs"][:, :1].clone(),
                                   g["shs"][:, 1:].clone(), torch.log(g["scales"]), g["rots"].clone(),
                                   torch.logit(g["opac"].reshape(-1, 1).clamp(1e-4, 1 - 1e-4)), v1, v2, v3, nrm, rad).to(dev)
        tr = Trainer(model)
        cam0 = Camera(cams[0], dev)
        zero_bg = torch.zeros(3, device=dev)
        for _ in range(10):
            tr.step(cam0, gt, zero_bg)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nit):
            tr.step(cam0, gt, zero_bg)
        torch.cuda.synchronize()
        out["fwd_bwd"]["ms_per_training_iteration"] = 1e3 * (time.perf_counter() - t1) / nit
        # the same iteration with the sync-free forward (no instance-count read-back: the host never waits inside an iteration), as C5 runs
        del tr
        tr = Trainer(model, sync_free=True)
        for _ in range(10):
            tr.step(cam0, gt, zero_bg)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nit):
            tr.step(cam0, gt, zero_bg)
        torch.cuda.synchronize()
        out["fwd_bwd"]["ms_per_training_iteration_sync_free"] = 1e3 * (time.perf_counter() - t1) / nit
        # BASELINE.json config C2: 500k Gaussians (the free-standing synthetic cloud of SURVEY.md 8d), 1080p, SH3, forward+backward
        del model, tr
        from gaussianmesh_amd import scenes as _sc
        c2 = _sc.make_cloud(500_000, seed=0)
        cam2 = _sc.orbit_camera(3, 64, W, H)
        t2 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
        rs2 = GaussianRasterizationSettings(H, W, cam2["tanx"], cam2["tany"], torch.zeros(3, device=dev), 1.0, t2(cam2["view"]), t2(cam2["proj"]),
                                            3, t2(cam2["campos"]), False, False, None if args.no_work_hint else Rz.new_work_hint(W, H, dev))
        rast2 = GaussianRasterizer(rs2)
        lv = [t2(c2[k]).requires_grad_(True) for k in ("means", "opac", "shs", "scales", "rots")]
        m2 = torch.zeros_like(lv[0], requires_grad=True)

        def it_c2():
            for l in lv + [m2]:
                l.grad = None
            color, _ = rast2(lv[0], m2, lv[1], shs=lv[2], scales=lv[3], rotations=lv[4])
            (color * wgt).sum().backward()
        for _ in range(10):
            it_c2()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(nit):
            it_c2()
        torch.cuda.synchronize()
        out["fwd_bwd"]["c2_500k_ms_per_iter"] = 1e3 * (time.perf_counter() - t1) / nit

    if rank == 0 and world == 1 and not args.no_fwd_bwd and not args.no_c5:
        # ---- bounded C5 leg: BASELINE config C5 (3 M Gaussians, 3840x2160, the train_mesh_gaussian.py iteration) on the same box,
        # --c5-iters iterations of Trainer.step; `python bench.py --config c5` runs the 1000 iterations of the config as a line of its own
        host_keep = {k: g[k].cpu() for k in ("tri", "weights", "pos", "cov", "opac", "shs", "verts")} if not args.no_cpu_baseline else None
        mesh_keep = g["mesh"][:24].cpu() if not args.no_cpu_baseline else None
        g.clear(); workspaces.clear(); pending.clear()
        del lv, m2, rast2, rs2, wgt, gt, rast, rs, v1_frames, hint, adjacency, stats      # C3 / C2 state: not part of C5's peak memory
        torch.cuda.empty_cache()
        # "c5": the reference's first 1000 iterations with their schedule (SH ramp, random background, densify_and_prune at 600 / 800 /
        # 1000); "c5_fixed": the fixed-topology SH-3 leg rounds 1-3 reported (--c5-iters iterations), for comparison across rounds
        out["c5"] = c5_leg(1000, 0, policy=args.policy, work_hint=not args.no_work_hint, dev=dev, as_reference=True)
        # the same loop on the scene rounds 1-4 timed (no features in the teacher: densify_and_prune selected one row in 2 M): what this
        # round's CODE does to the number the earlier rounds reported, separate from what the new scene does to it
        out["c5_r04_scene"] = {k: v for k, v in c5_leg(1000, 0, policy=args.policy, work_hint=not args.no_work_hint, dev=dev, as_reference=True, scene="r04").items()
                               if k in ("ms_per_iter", "ms_per_iter_before_first_densify", "densify_iterations_ms", "rows_after_densify", "loss_first", "loss_last",
                                        "iterations_redone", "visible", "peak_memory_gb")}
        out["c5_fixed"] = c5_leg(max(1, args.c5_iters), 10, policy=args.policy, work_hint=not args.no_work_hint, dev=dev)
        # the loop at each SH degree of the reference's ramp (27 000 of its 30 000 iterations run at degree 3), 200 iterations each from equal state
        out["c5_phases"] = c5_phases(max(1, args.c5_iters), 20, dev=dev)
    else:
        host_keep = mesh_keep = None

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline: the oracle port (plain C + OpenMP) on the host cores, bounded sample of the same workload
        from oracle import oracle as orc
        nfr_max, budget_s = 24, 10.0                # bounded sample: at least 2 frames, then until ~10 s of host work
        hp = {k: (host_keep[k] if host_keep is not None else g[k].cpu()).numpy() for k in ("tri", "weights", "pos", "cov", "opac", "shs", "verts")}
        mesh0 = (mesh_keep if mesh_keep is not None else g["mesh"][:nfr_max].cpu()).numpy()
        tc = time.perf_counter()
        nfr = 0
        for t in range(min(nfr_max, mesh0.shape[0])):
            if t >= 2 and time.perf_counter() - tc > budget_s:
                break
            nfr += 1
            ms = mesh0[t]
            dV = ms[:, 0:3] - hp["verts"]
            p2, c2, r2 = orc.deform(hp["tri"], hp["weights"], dV, ms[:, 3:12].reshape(-1, 3, 3), ms[:, 12:21].reshape(-1, 3, 3),
                                    hp["cov"], hp["pos"])
            rgb = orc.sh_colors_rotated(p2, cams[t]["campos"], r2, hp["shs"], deg=3)
            sc = dict(means=p2, opac=hp["opac"], colors_precomp=rgb, cov3D_precomp=scenes.strip_symmetric(c2))
            orc.forward_fast(sc, cams[t], np.ones(3, np.float32), D=3, use_precomp_cov=True, use_precomp_color=True)
        cpu_s = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": nfr / cpu_s, "unit": "frames/s", "cores": orc.num_threads(), "kind": "port",
                               "sample": "%d frames of the same C3 workload (deform + SH colour + forward) through "
                                         "oracle/libgm_oracle.so with OpenMP" % nfr}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()                           # rank 0 finishes its (collective-free) profiling pass first
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
