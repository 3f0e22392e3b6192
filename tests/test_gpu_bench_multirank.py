"""-m gpu: bench.py's N>1 branch executed for real - two ranks launched by torch.distributed.run on the one GPU of the test
box (GM_BENCH_SHARE_DEVICE=1, gloo instead of RCCL, which refuses two ranks on one device) on the HIP path: scene broadcast,
mesh-state broadcasts (batched, one batch ahead: multiview.MeshStatePipe) inside the multi-stream loop, view sharding, max-over-ranks timing, one JSON line from rank 0.
Each rank must have rendered ITS OWN view of the frame: the images equal a single-process render of those views bit for bit."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _launch(tmp_path, nproc, env_extra, P, W, H, F, steps, warm, batch=None, extra=()):
    """`python -m torch.distributed.run --nproc-per-node nproc bench.py --gpus nproc ...` exactly as the driver launches it;
    returns bench.py's JSON line."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", str(steps), "--warmup", str(warm),
           "--gaussians", str(P), "--width", str(W), "--height", str(H), "--cameras", str(F), "--check-dir", str(tmp_path),
           "--no-cpu-baseline", "--no-fwd-bwd"] + ([] if batch is None else ["--exchange-batch", str(batch)]) + list(extra)
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


def test_bench_eight_ranks_walk_the_c4_view_split(tmp_path):
    """BASELINE config C4's split - 64-camera trajectory, 8 views per rank - as bench.py shards it: `--gpus 8 --cameras 64`, eight
    ranks on the one GPU of the test box (gloo; a dry run of the launch, the exchange and the sharding, not a measurement): over
    eight loop steps rank r renders cameras 8r .. 8r+7 and nothing else; one broadcast carries eight steps of vertex positions."""
    P, W, H, F, steps, warm = 2000, 96, 64, 64, 8, 0
    out = _launch(tmp_path, 8, dict(GM_BENCH_SHARE_DEVICE="1", GM_BENCH_BACKEND="gloo"), P, W, H, F, steps, warm)
    assert out["n_gpus"] == 8 and out["config"]["parallelism"] == "views x8" and out["value"] > 0
    assert out["config"]["exchange"]["steps_per_broadcast"] == 8 and out["config"]["exchange"]["bytes_per_step"] == 7500 * 12
    seen = set()
    for r in range(8):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert d["views"].tolist() == list(range(8 * r, 8 * r + 8)), (r, d["views"].tolist())
        assert int(d["overflows"]) == 0 and np.isfinite(d["image"]).all()
        seen.update(d["views"].tolist())
    assert seen == set(range(64))


def test_c4_at_full_size_eight_ranks_on_one_gpu(tmp_path):
    """BASELINE config C4 AT ITS SIZE - the 1 M-Gaussian deformed scene, 1920x1080, 64-camera trajectory, 8 views per rank - with
    the eight ranks of `bench.py --gpus 8 --cameras 64` sharing the one GPU of the test box (gloo for the exchange; the images do
    not depend on the transport).  Not a measurement: what it shows is that the full-size multi-rank configuration runs end to end
    (one-time broadcast of the 0.3-GB cloud, batched vertex-position broadcasts, every rank deriving (R, S) itself, sync-free
    four-stream loop) and that a rank renders exactly what a single process renders for its view: rank r walks cameras 8r .. 8r+7,
    the 64 views are covered once, no frame is lost to an overflow, and the last images of two ranks are bit-identical to
    single-process renders of those (frame, camera) pairs."""
    P, W, H, F, steps, warm = 1_000_000, 1920, 1080, 64, 8, 0
    out = _launch(tmp_path, 8, dict(GM_BENCH_SHARE_DEVICE="1", GM_BENCH_BACKEND="gloo"), P, W, H, F, steps, warm)
    assert out["n_gpus"] == 8 and out["config"]["gaussians"] == P and out["config"]["width"] == W and out["value"] > 0
    ex = out["config"]["exchange"]
    assert ex["steps_per_broadcast"] == 8 and ex["bytes_per_step"] == 7500 * 12 and len(ex["ms_per_step_by_rank"]) == 8
    seen = set()
    for r in range(8):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert d["views"].tolist() == list(range(8 * r, 8 * r + 8)) and int(d["overflows"]) == 0 and np.isfinite(d["image"]).all()
        seen.update(d["views"].tolist())
    assert seen == set(range(64))
    sys.path.insert(0, ROOT)
    import bench
    from gpu_utils import T
    from gaussianmesh_amd import multiview, rasterizer as Rz, scenes
    from gaussianmesh_amd.deform import mesh_rs, pack_mesh_state
    host = bench.build_scene(P, W, H, F)
    g = {k: T(host[k]) for k in ("weights", "pos", "cov", "opac", "shs", "verts")}
    g["tri"] = T(host["tri"], dtype=torch.int32)
    last = warm + 2 * steps - 1                  # (bench.py runs one discarded region of `steps` in front of the timed one)
    for r in (0, 5):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        v = multiview.view_for_step(last, F, r, 8)
        assert int(d["view"]) == v == 8 * r + 7 and int(d["frame"]) == last % F
        cam = scenes.orbit_camera(v, F, W, H)
        ct = {n: T(cam[n]) for n in ("view", "proj", "campos")}
        state = mesh_rs(g["verts"], T(host["mesh"][last % F][:, 0:3]), T(host["faces"], dtype=torch.int32), want_state=True)[2]
        _, color, *_ = Rz.forward_deformed_begin(torch.ones(3, device="cuda"), g["tri"], g["weights"], pack_mesh_state(state, g["verts"]), g["cov"], g["pos"],
                                                 g["shs"], g["opac"], ct["view"], ct["proj"], cam["tanx"], cam["tany"], H, W, 3, ct["campos"]).finish()
        assert np.array_equal(d["image"], color.cpu().numpy()), "rank %d did not render view %d of frame %d" % (r, v, last % F)


@pytest.mark.parametrize("batch", [None, 1, 3])          # None: bench.py's own default at N > 1 (8)
def test_bench_two_ranks_render_their_own_views(tmp_path, batch):
    _two_ranks(tmp_path, batch, dict(GM_BENCH_SHARE_DEVICE="1", GM_BENCH_BACKEND="gloo"))


def _two_ranks(tmp_path, batch, env_extra):
    P, W, H, F, steps, warm = 20000, 320, 200, 8, 4, 2
    out = _launch(tmp_path, 2, env_extra, P, W, H, F, steps, warm, batch)
    batch = 8 if batch is None else batch
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["scaling"] == "weak" and out["value"] > 0
    ex = out["config"]["exchange"]                                      # vertex positions: `batch` loop steps per broadcast, one batch ahead
    assert ex["steps_per_broadcast"] == batch and ex["broadcasts"] >= (3 * F + warm + steps) // batch and ex["bytes_per_step"] == 7500 * 12
    assert abs(out["value"] - 2 * steps / (out["ms_per_step"] * 1e-3 * steps)) < 1e-6 * out["value"]        # frames of both ranks / max time
    by_rank = ex["ms_per_step_by_rank"]                                 # every rank's own clock; the line's time is the slowest rank's
    assert len(by_rank) == 2 and abs(max(by_rank) - out["ms_per_step"]) <= 1e-3 and ex["slowest_over_fastest_rank"] >= 1.0
    # single-process render of the same frame for each rank's view
    sys.path.insert(0, ROOT)
    import bench
    from gpu_utils import T
    from gaussianmesh_amd import multiview, rasterizer as Rz, scenes
    from gaussianmesh_amd.deform import mesh_rs, pack_mesh_state
    host = bench.build_scene(P, W, H, F)
    g = {k: T(host[k]) for k in ("weights", "pos", "cov", "opac", "shs", "verts")}
    g["tri"] = T(host["tri"], dtype=torch.int32)
    last = warm + 2 * steps - 1                  # (bench.py runs one discarded region of `steps` in front of the timed one)
    views = set()
    for r in range(2):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        v = multiview.view_for_step(last, F, r, 2)
        assert int(d["step"]) == last and int(d["view"]) == v and int(d["frame"]) == last % F
        views.add(v)
        cam = scenes.orbit_camera(v, F, W, H)
        ct = {n: T(cam[n]) for n in ("view", "proj", "campos")}
        state = mesh_rs(g["verts"], T(host["mesh"][last % F][:, 0:3]), T(host["faces"], dtype=torch.int32), want_state=True)[2]
        packed = pack_mesh_state(state, g["verts"])
        _, color, *_ = Rz.forward_deformed_begin(torch.ones(3, device="cuda"), g["tri"], g["weights"], packed, g["cov"], g["pos"], g["shs"],
                                                 g["opac"], ct["view"], ct["proj"], cam["tanx"], cam["tany"], H, W, 3, ct["campos"]).finish()
        assert np.array_equal(d["image"], color.cpu().numpy()), "rank %d did not render view %d of frame %d" % (r, v, last % F)
    assert len(views) == 2                                               # the two ranks rendered different cameras
