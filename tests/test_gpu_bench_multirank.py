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


@pytest.mark.parametrize("batch", [None, 1, 3])          # None: bench.py's own default at N > 1 (8)
def test_bench_two_ranks_render_their_own_views(tmp_path, batch):
    P, W, H, F, steps, warm = 20000, 320, 200, 8, 4, 2
    env = dict(os.environ, GM_BENCH_SHARE_DEVICE="1", GM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warm),
           "--gaussians", str(P), "--width", str(W), "--height", str(H), "--cameras", str(F), "--check-dir", str(tmp_path),
           "--no-cpu-baseline", "--no-fwd-bwd"] + ([] if batch is None else ["--exchange-batch", str(batch)])
    batch = 8 if batch is None else batch
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["scaling"] == "weak" and out["value"] > 0
    ex = out["config"]["exchange"]                                      # mesh tables: `batch` loop steps per broadcast, one batch ahead
    assert ex["steps_per_broadcast"] == batch and ex["broadcasts"] >= (3 * F + warm + steps) // batch
    assert abs(out["value"] - 2 * steps / (out["ms_per_step"] * 1e-3 * steps)) < 1e-6 * out["value"]        # frames of both ranks / max time
    # single-process render of the same frame for each rank's view
    sys.path.insert(0, ROOT)
    import bench
    from gpu_utils import T
    from gaussianmesh_amd import multiview, rasterizer as Rz, scenes
    from gaussianmesh_amd.deform import mesh_rs, pack_mesh_state
    host = bench.build_scene(P, W, H, F)
    g = {k: T(host[k]) for k in ("weights", "pos", "cov", "opac", "shs", "verts")}
    g["tri"] = T(host["tri"], dtype=torch.int32)
    last = warm + steps - 1
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
