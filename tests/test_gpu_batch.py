"""gm_forward_deformed_batch_async (round 6): K frames of one view stream from ONE pass over the static cloud, every later stage one
launch over the K frames.  The contract is equivalence, frame by frame and bit for bit, with the single-frame calls
(gm_forward_0_deformed_stream_async + gm_forward_1_geom): radii, emission records, depth keys, the sorted instance lists, tile ranges,
the image, the status words - asserted here for K = 1 .. 4, both covariance layouts, image-only and with the backward state written,
refused (overflowing) frames included; plus the batched gm_mesh_rs_packed_batch against gm_mesh_rs_packed."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(P, W, H, F):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from gpu_utils import T
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.deform import vertex_face_adjacency
    host = bench.build_scene(P, W, H, F)
    g = {k: T(host[k]) for k in ("weights", "pos", "cov", "opac", "shs", "verts")}
    g["tri"] = T(host["tri"], dtype=torch.int32)
    g["faces"] = T(host["faces"], dtype=torch.int32)
    off, adj = vertex_face_adjacency(host["faces"], host["verts"].shape[0])
    g["adjacency"] = (torch.tensor(off, device="cuda"), torch.tensor(adj, device="cuda"))
    g["v1"] = [T(np.ascontiguousarray(host["mesh"][t][:, 0:3])) for t in range(F)]
    cams = []
    for k in range(F):
        cam = scenes.orbit_camera(k, F, W, H)
        cams.append(dict(view=T(cam["view"]), proj=T(cam["proj"]), campos=T(cam["campos"]), tanx=cam["tanx"], tany=cam["tany"]))
    return g, cams


def _state(h, P, W, H, policy, nr):
    """everything a frame leaves behind that later stages or the caller read, as numpy"""
    from gpu_utils import _view
    from gaussianmesh_amd import _lib
    lib = _lib.lib()
    geom, binning, img = h.geom, h.binning if getattr(h, "binning", None) is not None else h.result[4], h.img
    gp = lambda n: lib.gm_geom_field(geom.data_ptr(), P, n.encode())
    out = {"radii": h.radii.cpu().numpy(), "color": h.color.cpu().numpy()}
    SF = lib.gm_splat_floats()
    vis = out["radii"] > 0
    out["splat"] = _view(geom, gp("splat"), P * SF, torch.float32).reshape(P, SF)[vis]
    out["depth_key"] = _view(geom, gp("depth_key"), P, torch.int32)
    V = int(_view(geom, gp("bucket_start"), 2049, torch.int32)[2048])
    out["order"] = _view(geom, gp("order"), P, torch.int32)[:V]
    out["counters"] = _view(geom, gp("counters"), 8, torch.int32)[[0, 2, 3, 4]]            # rendered, policy, refused, visible
    sh = max(policy - 1, 0)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tiles = ((gx + (1 << sh) - 1) >> sh) * ((gy + (1 << sh) - 1) >> sh)
    out["ranges"] = _view(img, lib.gm_image_field(img.data_ptr(), W, H, b"ranges"), tiles * 2, torch.int32)
    if nr > 0:
        cap = h.workspace.capacity
        out["pairs"] = _view(binning, lib.gm_binning_field(binning.data_ptr(), cap, W, H, policy, b"pairs"), 2 * nr, torch.int32)
    return out


@pytest.mark.parametrize("K,cov6,image_only", [(1, True, True), (2, False, True), (3, True, False), (4, True, True), (4, False, False), (8, True, True)])
def test_batched_frames_equal_the_single_frame_calls(K, cov6, image_only):
    """Each of the K frames of a batch against forward_deformed_begin(...).finish(sync_free=True) of the same (mesh frame, camera) on
    buffers of the same capacity: every byte a later stage or the caller reads is the same."""
    from gaussianmesh_amd import rasterizer as Rz
    from gaussianmesh_amd.deform import mesh_rs_packed, mesh_rs_packed_batch, pack_cov6
    P, W, H, F = (30000 if K < 8 else 29989), 480, 270, 8                       # (K = GM_BATCH_MAX on a row count that fills no wave)
    g, cams = _scene(P, W, H, F)
    policy = Rz.get_default_emission_policy(W, H)
    bg = torch.tensor([0.2, 0.5, 0.7], device="cuda")
    cov = pack_cov6(g["cov"]) if cov6 else g["cov"]
    assert cov is not None
    hint = Rz.new_work_hint(W, H, bg.device)
    pairs = [(1, 5), (4, 2), (6, 7), (3, 0), (0, 1), (2, 3), (5, 4), (7, 6)][:K]   # (mesh frame, camera) of the batch's frames
    # reference: the single-frame path, one workspace per frame; the first (exact) pass learns the capacity
    ref_ws = [Rz.RasterWorkspace() for _ in range(K)]
    ref = []
    for (t, c), ws in zip(pairs, ref_ws):
        packed = mesh_rs_packed(g["verts"], g["v1"][t], g["faces"], g["adjacency"])
        cm = cams[c]
        begin = lambda: Rz.forward_deformed_begin(bg, g["tri"], g["weights"], packed, cov, g["pos"], g["shs"], g["opac"], cm["view"], cm["proj"],
                                                  cm["tanx"], cm["tany"], H, W, 3, cm["campos"], False, workspace=ws, want_count=True)
        begin().finish(image_only=image_only)                                   # exact pass: sizes the binning buffer
    cap = max(ws.capacity for ws in ref_ws)
    for (t, c), ws in zip(pairs, ref_ws):
        ws.capacity = cap
        packed = mesh_rs_packed(g["verts"], g["v1"][t], g["faces"], g["adjacency"])
        cm = cams[c]
        h = Rz.forward_deformed_begin(bg, g["tri"], g["weights"], packed, cov, g["pos"], g["shs"], g["opac"], cm["view"], cm["proj"],
                                      cm["tanx"], cm["tany"], H, W, 3, cm["campos"], False, workspace=ws, want_count=False)
        h.finish(sync_free=True, image_only=image_only, work_hint=hint)
        ok, nr = h.check()
        assert ok and nr > 0
        ref.append((_state(h, P, W, H, policy, nr), nr, packed.cpu().numpy()))
    # the batch
    ws = [Rz.RasterWorkspace() for _ in range(K)]
    for w_ in ws:
        w_.capacity = cap
    tables = mesh_rs_packed_batch(g["verts"], [g["v1"][t] for t, _ in pairs], g["faces"], g["adjacency"])
    hs = Rz.forward_deformed_batch(bg, g["tri"], g["weights"], tables, cov, g["pos"], g["shs"], g["opac"], [cams[c] for _, c in pairs], H, W, 3, ws,
                                   image_only=image_only, work_hint=hint)
    torch.cuda.synchronize()
    for k, h in enumerate(hs):
        ok, nr = h.check()
        assert ok and nr == ref[k][1], (k, ok, nr, ref[k][1])
        assert np.array_equal(tables[k].cpu().numpy(), ref[k][2]), "gm_mesh_rs_packed_batch table of frame %d" % k
        st = _state(h, P, W, H, policy, nr)
        for name, a in ref[k][0].items():
            assert np.array_equal(st[name], a), "frame %d of a batch of %d: %s differs from the single-frame call" % (k, K, name)
        if not image_only:                                                       # the per-pixel backward state too
            from gpu_utils import _view
            from gaussianmesh_amd import _lib
            for fld, dt in (("final_T", torch.float32), ("n_contrib", torch.int32)):
                a = _view(h.img, _lib.lib().gm_image_field(h.img.data_ptr(), W, H, fld.encode()), W * H, dt)
                b = _view(ref_ws[k]._bufs["img"], _lib.lib().gm_image_field(ref_ws[k]._bufs["img"].data_ptr(), W, H, fld.encode()), W * H, dt)
                assert np.array_equal(a, b), (k, fld)


def test_a_frame_that_outgrows_the_batch_capacity_is_refused_alone_and_redone():
    """Sync-free contract inside a batch: the capacity is per frame.  With a capacity between the instance counts of two frames the
    lighter frame renders, the heavier one is refused in ITS status words (image = background) and finish() renders it again, exactly,
    through the single-frame second half - from the geometry state the batch left."""
    from gaussianmesh_amd import rasterizer as Rz
    from gaussianmesh_amd.deform import mesh_rs_packed_batch
    P, W, H, F = 30000, 480, 270, 8
    g, cams = _scene(P, W, H, F)
    bg = torch.tensor([0.9, 0.1, 0.3], device="cuda")
    # the same mesh frame seen from the orbit and from three times as far away: very different instance counts
    from gaussianmesh_amd import scenes
    from gpu_utils import T
    near = cams[2]
    cam_far = scenes.orbit_camera(2, F, W, H, radius=18.0)
    far = dict(view=T(cam_far["view"]), proj=T(cam_far["proj"]), campos=T(cam_far["campos"]), tanx=cam_far["tanx"], tany=cam_far["tany"])
    tables = mesh_rs_packed_batch(g["verts"], [g["v1"][1], g["v1"][1]], g["faces"], g["adjacency"])
    counts, images = [], []
    for cm, tab in ((near, tables[0]), (far, tables[1])):
        nr, color, *_ = Rz.forward_deformed_begin(bg, g["tri"], g["weights"], tab, g["cov"], g["pos"], g["shs"], g["opac"], cm["view"], cm["proj"], cm["tanx"],
                                                  cm["tany"], H, W, 3, cm["campos"], False).finish(image_only=True)
        counts.append(nr); images.append(color.clone())
    assert counts[0] > 1.1 * counts[1] > 0, counts
    ws = [Rz.RasterWorkspace() for _ in range(2)]
    for w_ in ws:
        w_.capacity = (counts[0] + counts[1]) // 2
    hs = Rz.forward_deformed_batch(bg, g["tri"], g["weights"], tables, g["cov"], g["pos"], g["shs"], g["opac"], [near, far], H, W, 3, ws, image_only=True)
    ok0, nr0 = hs[0].check()
    ok1, nr1 = hs[1].check()
    assert (ok0, nr0) == (False, counts[0]) and (ok1, nr1) == (True, counts[1])
    assert torch.equal(hs[1].color, images[1])
    assert torch.equal(hs[0].color, bg.reshape(3, 1, 1).expand(3, H, W))         # refused: the background
    nr, color, *_ = hs[0].finish(image_only=True)
    torch.cuda.synchronize()
    assert nr == counts[0] and torch.equal(color, images[0])
    assert ws[0].in_flight is None and ws[0].capacity > counts[0]


def test_a_frame_that_sees_nothing_inside_a_batch():
    """An empty frame between two ordinary ones (a camera that looks away from the cloud: every Gaussian culled).  Its launches of the
    batched chain have nothing to order and nothing to blend: zero instances in its status words, no radius set, the image the background -
    and its neighbours in the batch are, byte for byte, what they are alone."""
    from gaussianmesh_amd import rasterizer as Rz, scenes
    from gaussianmesh_amd.deform import mesh_rs_packed_batch
    from gpu_utils import T
    P, W, H, F = 20000, 320, 200, 4
    g, cams = _scene(P, W, H, F)
    bg = torch.tensor([0.3, 0.6, 0.1], device="cuda")
    away = scenes.look_at_camera((8.0, 1.5, 0.0), (16.0, 1.5, 0.0), W, H, 60.0)                    # on the orbit, back to the torus
    blind = dict(view=T(away["view"]), proj=T(away["proj"]), campos=T(away["campos"]), tanx=away["tanx"], tany=away["tany"])
    frames = [(1, cams[1]), (2, blind), (3, cams[3])]
    tables = mesh_rs_packed_batch(g["verts"], [g["v1"][t] for t, _ in frames], g["faces"], g["adjacency"])
    alone = []
    for (t, cm), tab in zip(frames, tables):
        nr, color, radii, *_ = Rz.forward_deformed_begin(bg, g["tri"], g["weights"], tab, g["cov"], g["pos"], g["shs"], g["opac"], cm["view"], cm["proj"],
                                                         cm["tanx"], cm["tany"], H, W, 3, cm["campos"], False).finish(image_only=True)
        alone.append((nr, color.clone(), radii.clone()))
    assert alone[1][0] == 0 and alone[0][0] > 0 and alone[2][0] > 0, [a[0] for a in alone]
    ws = [Rz.RasterWorkspace() for _ in frames]
    for w_ in ws:
        w_.capacity = 2 * max(a[0] for a in alone)
    hs = Rz.forward_deformed_batch(bg, g["tri"], g["weights"], tables, g["cov"], g["pos"], g["shs"], g["opac"], [cm for _, cm in frames], H, W, 3, ws, image_only=True)
    for k, h in enumerate(hs):
        ok, nr = h.check()
        assert ok and nr == alone[k][0], (k, ok, nr)
        assert torch.equal(h.color, alone[k][1]) and torch.equal(h.radii, alone[k][2]), k
    assert int((hs[1].radii != 0).sum()) == 0
    assert torch.equal(hs[1].color, bg.reshape(3, 1, 1).expand(3, H, W))


def test_batch_argument_errors():
    from gaussianmesh_amd import _lib, rasterizer as Rz
    from gaussianmesh_amd.deform import mesh_rs_packed_batch
    P, W, H, F = 2000, 96, 64, 4
    g, cams = _scene(P, W, H, F)
    bg = torch.zeros(3, device="cuda")
    tables = mesh_rs_packed_batch(g["verts"], [g["v1"][0], g["v1"][1]], g["faces"], g["adjacency"])
    ws = [Rz.RasterWorkspace() for _ in range(2)]
    with pytest.raises(_lib.GmeshError, match="capacity"):                      # no frame of the stream completed yet
        Rz.forward_deformed_batch(bg, g["tri"], g["weights"], tables, g["cov"], g["pos"], g["shs"], g["opac"], cams[:2], H, W, 3, ws)
    assert all(w_.in_flight is None for w_ in ws)
    for w_ in ws:
        w_.capacity = 100000
    with pytest.raises(ValueError):                                               # one workspace for two frames
        Rz.forward_deformed_batch(bg, g["tri"], g["weights"], tables, g["cov"], g["pos"], g["shs"], g["opac"], cams[:2], H, W, 3, [ws[0], ws[0]])
    with pytest.raises(_lib.GmeshError, match="list tiles"):                    # policy 0 at this size: 24 tiles - fine; 4K under policy 2 is not
        Rz.forward_deformed_batch(bg, g["tri"], g["weights"], tables, g["cov"], g["pos"], g["shs"], g["opac"], cams[:2], 2160, 3840, 3, ws, emission_policy=2)
    assert all(w_.in_flight is None for w_ in ws)
    hs = Rz.forward_deformed_batch(bg, g["tri"], g["weights"], tables, g["cov"], g["pos"], g["shs"], g["opac"], cams[:2], H, W, 3, ws)
    assert all(h.check()[0] for h in hs)
    many = _lib.GM_BATCH_MAX + 1                                                  # more frames than one launch chain carries
    ws9 = [Rz.RasterWorkspace() for _ in range(many)]
    for w_ in ws9:
        w_.capacity = 100000
    with pytest.raises((ValueError, _lib.GmeshError)):
        Rz.forward_deformed_batch(bg, g["tri"], g["weights"], [tables[0]] * many, g["cov"], g["pos"], g["shs"], g["opac"], [cams[0]] * many, H, W, 3, ws9)
    assert all(w_.in_flight is None for w_ in ws9)
