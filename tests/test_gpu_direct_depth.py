"""Direct depth placement (gm_forward_0_deformed_stream_async, DepthPlan): frames of one view stream place their Gaussians in
depth buckets looked up in a table an EARLIER frame left behind.  The (depth, id) order - and so every list and image - must not
depend on that table; a frame the placement cannot order must say so and come out right when it is begun again.
Reference behaviour: RAST/rasterizer_impl.cu:407-489 (one stable sort of the instance keys per frame)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(P, W, H, F, seed=0):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from gpu_utils import T
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.deform import mesh_rs_packed, vertex_face_adjacency
    host = bench.build_scene(P, W, H, F, seed=seed)
    g = {k: T(host[k]) for k in ("weights", "pos", "cov", "opac", "shs", "verts")}
    g["tri"] = T(host["tri"], dtype=torch.int32)
    faces = T(host["faces"], dtype=torch.int32)
    off, adj = vertex_face_adjacency(host["faces"], host["verts"].shape[0])
    adjacency = (torch.tensor(off, device="cuda"), torch.tensor(adj, device="cuda"))
    g["packed"] = [mesh_rs_packed(g["verts"], T(np.ascontiguousarray(host["mesh"][t][:, 0:3])), faces, adjacency) for t in range(F)]
    cams = []
    for k in range(F):
        cam = scenes.orbit_camera(k, F, W, H)
        cams.append((T(cam["view"]), T(cam["proj"]), cam["tanx"], cam["tany"], T(cam["campos"])))
    return g, cams


def _lists(h, out, P, W, H):
    """order of the visible Gaussians and the tile-sorted instance stream of a finished frame"""
    from gaussianmesh_amd import _lib
    from gpu_utils import _view
    lib = _lib.lib()
    nr, _, _, geom, binning, _ = out
    torch.cuda.synchronize()
    gp = lambda n: lib.gm_geom_field(geom.data_ptr(), P, n.encode())
    V = int(_view(geom, gp("bucket_start"), 2049, torch.int32)[2048])
    order = _view(geom, gp("order"), P, torch.int32)[:V].copy()
    pp = lib.gm_binning_field(binning.data_ptr(), nr, W, H, h.policy, b"pairs")
    pairs = _view(binning, pp, 2 * nr, torch.int32).reshape(nr, 2).copy()
    return order, pairs


def _frame(g, cams, t, k, W, H, bg, plan=None, ws=None, **kw):
    from gaussianmesh_amd import rasterizer as Rz
    c = cams[k]
    return Rz.forward_deformed_begin(bg, g["tri"], g["weights"], g["packed"][t], g["cov"], g["pos"], g["shs"], g["opac"], c[0], c[1], c[2],
                                     c[3], H, W, 3, c[4], False, workspace=ws, depth_plan=plan, **kw)


@pytest.mark.parametrize("P,W,H", [(70, 64, 48), (3000, 160, 96), (60000, 480, 320)])
def test_lists_do_not_depend_on_the_depth_path(P, W, H):
    """every frame of an orbit, twice: partition path and direct placement (table of the frame before).  Visible order, instance
    stream (keys and ids) and image bit-identical; the exact completion path notices a refused frame by itself."""
    from gaussianmesh_amd import rasterizer as Rz
    F = 12
    g, cams = _scene(P, W, H, F)
    bg = torch.zeros(3, device="cuda")
    plan = Rz.new_depth_plan(bg.device)
    ws = Rz.RasterWorkspace()
    direct_frames = 0
    for i in range(2 * F):
        t, k = i % F, i % F
        h0 = _frame(g, cams, t, k, W, H, bg)
        o0 = h0.finish()
        ord0, pairs0 = _lists(h0, o0, P, W, H)
        img0 = o0[1].clone()
        h1 = _frame(g, cams, t, k, W, H, bg, plan=plan, ws=ws)
        was_direct = h1.direct
        o1 = h1.finish()
        ord1, pairs1 = _lists(h1, o1, P, W, H)
        assert o1[0] == o0[0]
        assert np.array_equal(ord0, ord1), "frame %d: visible order differs (direct=%s)" % (i, was_direct)
        assert np.array_equal(pairs0, pairs1), "frame %d: instance stream differs" % i
        assert torch.equal(img0, o1[1])
        direct_frames += int(was_direct)
    assert direct_frames == 2 * F - 1                    # all but the stream's first frame were begun direct
    print("P=%d: %d direct frames, %d refused and rendered again" % (P, direct_frames, plan.refused))
    assert plan.refused <= F // 2, "neighbouring cameras of a 12-view orbit should rarely need the partition path"


def test_refused_frames_come_out_right():
    """what the direct placement cannot order: (a) a table made for another scene scale (every key lands in one bucket: above the
    slab), (b) thousands of equal depths (a pile the one-word sort does not take).  Status 2, and the frame begun again matches."""
    from gaussianmesh_amd import rasterizer as Rz
    P, W, H, F = 40000, 320, 200, 4
    g, cams = _scene(P, W, H, F)
    bg = torch.zeros(3, device="cuda")
    ref = _frame(g, cams, 0, 0, W, H, bg).finish()[1].clone()
    # (a) prime the plan with a view from far away (all depths in other coarse bins), then render the near view direct
    far = []
    for c in cams:
        v = c[0].clone(); v[3, 2] += 400.0                # row-vector convention: translation in the last row; push the scene 400 units away
        far.append((v, c[1], c[2], c[3], c[4]))
    plan = Rz.new_depth_plan(bg.device)
    ws = Rz.RasterWorkspace()
    _frame(g, far, 0, 0, W, H, bg, plan=plan, ws=ws).finish()
    h = _frame(g, cams, 0, 0, W, H, bg, plan=plan, ws=ws, want_count=False)
    assert h.direct
    ws.capacity = 4_000_000
    out = h.finish(sync_free=True)
    ok, _ = h.check()
    assert not ok and h.refusal == 2
    img = h.finish()[1]
    assert torch.equal(img, ref)
    assert plan.refused == 1
    # the frame after it finds the table the repeated frame left: direct, not refused
    h = _frame(g, cams, 0, 1, W, H, bg, plan=plan, ws=ws, want_count=False)
    out = h.finish(sync_free=True)
    ok, _ = h.check()
    assert ok and h.direct
    assert torch.equal(out[1], _frame(g, cams, 0, 1, W, H, bg).finish()[1])
    # (b) 300 Gaussians moved onto a sheet facing the camera, at (nearly) one view depth: they fit their bucket's slab, but the
    # in-LDS sort meets a pile of equal keys; (c) the whole cloud on the sheet: far above the slab
    cam = cams[0]
    view = cam[0].cpu().numpy().astype(np.float64)
    Rm, tr = view[:3, :3], view[3, :3]                     # p_view = p @ Rm + tr
    from gaussianmesh_amd.deform import pack_mesh_state
    ident = torch.zeros((g["verts"].shape[0], 21), device="cuda")
    ident[:, 0:3] = g["verts"]
    for j in (3, 7, 11, 12, 16, 20):
        ident[:, j] = 1.0
    for n_sheet in (300, P):
        rng = np.random.default_rng(5)
        pv = np.stack([rng.uniform(-1.5, 1.5, n_sheet), rng.uniform(-1.0, 1.0, n_sheet), np.full(n_sheet, 5.0)], 1)
        pw = (pv - tr) @ np.linalg.inv(Rm)
        g2 = dict(g)
        pos = g["pos"].clone()
        pos[:n_sheet] = torch.tensor(pw, dtype=torch.float32, device="cuda")
        g2["pos"] = pos
        g2["packed"] = [pack_mesh_state(ident, g["verts"])]
        ref2 = _frame(g2, cams, 0, 0, W, H, bg).finish()[1].clone()
        plan2 = Rz.new_depth_plan(bg.device)
        _frame(g2, cams, 0, 0, W, H, bg, plan=plan2, ws=ws).finish()
        h = _frame(g2, cams, 0, 0, W, H, bg, plan=plan2, ws=ws)
        assert h.direct
        img2 = h.finish()[1]                                # exact completion: looks at the status itself
        assert torch.equal(img2, ref2)
        assert plan2.refused == 1, "a pile of %d equal depths must be refused" % n_sheet


def test_pipelined_loop_with_a_depth_plan_every_frame():
    """the four-stream loop of test_gpu_parity with a DepthPlan: (i) cameras 135 degrees apart from frame to frame - tables that
    fit badly, frames in flight publishing tables while others read them; (ii) neighbouring cameras.  Every frame bit-identical
    to the synchronous render; in (ii) nothing is refused once the stream is under way."""
    from test_gpu_parity import pipelined_deformed_loop
    plan = pipelined_deformed_loop(20000, 320, 200, 8, 320, plan=True)
    print("135-degree jumps: %d of 320 frames refused" % plan.refused)
    plan = pipelined_deformed_loop(60000, 480, 320, 64, 256, plan=True, cam_stride=1)
    print("neighbouring cameras: %d of 256 frames refused" % plan.refused)
    assert plan.refused <= 8
