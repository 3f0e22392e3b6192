"""-m gpu tests at BASELINE.json's full sizes (C1, C2, C3, C5 shapes) through size-independent properties, plus one
mid-size oracle comparison.  The oracle is only used where it finishes in seconds."""
import numpy as np
import pytest
import torch

from helpers import assert_forward_gate, assert_grads_elementwise

pytestmark = pytest.mark.gpu


def _check_all_grads(g, bw, rtol):
    """All eight gradient tensors of the operator (SH + scale/rot input mode) against the oracle's: max-abs difference
    relative to the tensor's max magnitude (float atomics: summation order is not reproducible), and relative L2."""
    from test_gpu_parity import _rel
    pairs = [("means", bw["dmean3D"]), ("m2d", bw["dmean2D"]), ("opac", bw["dopacity"]), ("shs", bw["dsh"]),
             ("scales", bw["dscale"]), ("rots", bw["drot"])]
    for name, ref in pairs:
        got = np.asarray(g[name], np.float64).reshape(np.shape(ref)) if name != "m2d" else np.asarray(g[name], np.float64)
        ref = np.asarray(ref, np.float64)
        if name == "m2d":
            got, ref = got[:, :2], ref[:, :2]
        assert _rel(got, ref) <= rtol, (name, _rel(got, ref))
        l2 = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)
        assert l2 <= rtol, (name, "L2", l2)
        n, bad, worst = assert_grads_elementwise(got, ref, name)            # every entry >= 1e-3 x max to 1e-2 of ITSELF
        print("gradient %-7s max-norm %.2e  L2 %.2e  element-wise gate: %d entries, %d above 1e-2 (worst %.2e)" % (name, _rel(got, ref), l2, n, bad, worst))


def _forward(sc, cam, bg, **kw):
    from gpu_utils import forward_state
    return forward_state(sc, cam, bg, D=3, **kw)


def _check_list_invariants(st, cam, mode=0):
    """Structural invariants of the binning output, any size (lists are per parent tile of 16 << shift pixels)."""
    shift = max(mode - 1, 0); ts = 16 << shift
    T = st["ranges"].shape[0]
    r = st["ranges"].astype(np.int64)
    nonempty = r[:, 1] > r[:, 0]
    assert (r[~nonempty] == 0).all()
    # ranges of non-empty tiles, in tile order, partition [0, R)
    seg = r[nonempty]
    assert seg[0, 0] == 0 and seg[-1, 1] == st["R"] and (seg[1:, 0] == seg[:-1, 1]).all()
    assert st["R"] == int(st["tiles"].astype(np.int64).sum())
    keys = st["tile_keys"].astype(np.int64)
    assert (np.diff(keys) >= 0).all() and keys.max() < T
    # inside a tile: depth non-decreasing, ties broken by ascending Gaussian id (the reference's stable sort)
    depth_bits = st["depth_key"].astype(np.int64)[st["point_list"]]
    same = keys[1:] == keys[:-1]
    dd = np.diff(depth_bits)
    assert (dd[same] >= 0).all()
    tie = same & (dd == 0)
    assert (np.diff(st["point_list"].astype(np.int64))[tie] > 0).all()
    # every emitted instance lies inside its Gaussian's tile rectangle
    gx = (((cam["W"] + 15) // 16) + (1 << shift) - 1) >> shift
    g = st["point_list"]
    x, y, rad = st["splat"][g, 0], st["splat"][g, 1], st["radii"][g]
    tx, ty = keys % gx, keys // gx
    assert ((tx * ts <= x + rad + 15) & (tx * ts + ts - 1 >= x - rad - 15) & (ty * ts <= y + rad + 15) & (ty * ts + ts - 1 >= y - rad - 15)).all()
    # mask bits: policy 2 (32-px parents) carries one bit per 8x8 quadrant of the parent (16), policy 3 one per 16-px child (16)
    assert (st["child_mask"] < (1 << (16 if mode >= 2 else 1))).all()


def test_c1_plumbing_case(oracle):
    """C1: 10k Gaussians, 256x256, SH degree 0 - oracle forward is the CPU reference; GPU must match it."""
    from gaussianmesh_amd import scenes
    sc = scenes.make_cloud(10000, seed=0, D=0); sc["D"] = 0
    cam = scenes.orbit_camera(0, 1, 256, 256)
    for bg in (np.zeros(3, np.float32), np.ones(3, np.float32)):
        fw = oracle.forward_full(sc, cam, bg, D=0)
        from gpu_utils import forward_state
        st = forward_state(sc, cam, bg, D=0)
        assert np.array_equal(st["radii"], fw["geo"]["radii"]) and np.array_equal(st["point_list"], fw["bins"]["point_list"])
        # C1: at most a pixel or two go through the flip exemption (round 2's pixel-relative exponent: none; the matrix-core
        # polynomial of round 3 carries ~1e-5 of absolute error in the exponent and lands on the other side of a threshold that
        # lies within the entry's own rounding distance slightly more often)
        assert assert_forward_gate(fw, st["color"], 256, 256, 1e-4, "C1", plain_tol=5e-5) <= 2


def test_mid_size_oracle_parity(oracle):
    """100k Gaussians at 640x360 against the oracle (forward, lists bit-exact, backward)."""
    from gaussianmesh_amd import scenes
    from test_gpu_parity import _grads_gpu, _rel
    sc = scenes.make_cloud(100000, seed=3, scale_lo=0.008, scale_hi=0.08)
    cam = scenes.orbit_camera(5, 64, 640, 360)
    bg = np.zeros(3, np.float32)
    fw = oracle.forward_full(sc, cam, bg, D=3)
    st = _forward(sc, cam, bg)
    assert np.array_equal(st["radii"], fw["geo"]["radii"]) and st["R"] == fw["bins"]["R"]
    assert np.array_equal(st["point_list"], fw["bins"]["point_list"]) and np.array_equal(st["ranges"], fw["bins"]["ranges"])
    assert_forward_gate(fw, st["color"], 640, 360, 1e-4, "100k", plain_tol=5e-5)
    dpix = np.random.default_rng(1).normal(size=(3, 360, 640)).astype(np.float32)
    bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=3)
    _, _, g = _grads_gpu(sc, cam, bg, dpix, 3, False, False)
    _check_all_grads(g, bw, 1e-3)


@pytest.mark.parametrize("P,W,H", [(500_000, 1920, 1080), (1_000_000, 1920, 1080)])
def test_full_size_forward_properties(P, W, H):
    """C2 / C3 sizes: structural invariants, emission policies agree bit for bit, determinism, affinity in background."""
    from gaussianmesh_amd import scenes
    sc = scenes.make_cloud(P, seed=0)
    cam = scenes.orbit_camera(3, 64, W, H)
    z = np.zeros(3, np.float32)
    ex = _forward(sc, cam, z, tile_cull=0)
    _check_list_invariants(ex, cam)
    prev_R = ex["R"]
    for mode in (1, 2, 3):
        cu = _forward(sc, cam, z, tile_cull=mode)
        _check_list_invariants(cu, cam, mode)
        assert cu["R"] < prev_R and np.array_equal(cu["radii"], ex["radii"])
        prev_R = cu["R"]
        assert np.array_equal(cu["color"], ex["color"]) and np.array_equal(cu["final_T"], ex["final_T"])
        again = _forward(sc, cam, z, tile_cull=mode)
        assert np.array_equal(again["color"], cu["color"]) and np.array_equal(again["point_list"], cu["point_list"])   # deterministic
        one = _forward(sc, cam, np.ones(3, np.float32), tile_cull=mode)
        assert np.allclose(one["color"] - cu["color"], cu["final_T"].reshape(1, H, W), atol=1e-6)       # C + T*bg
        assert cu["color"].min() >= 0 and np.isfinite(cu["color"]).all() and (cu["final_T"] <= 1).all() and (cu["final_T"] >= 0).all()
        # n_contrib indexes into the parent tile's list
        nc = cu["n_contrib"].reshape(H, W)
        sh = mode - 1
        pgx = (((W + 15) // 16) + (1 << sh) - 1) >> sh
        lens = (cu["ranges"][:, 1] - cu["ranges"][:, 0]).astype(np.int64)
        ty, tx = np.mgrid[0:H, 0:W]
        assert (nc <= lens[(ty // (16 << sh)) * pgx + tx // (16 << sh)]).all()


def test_c2_backward_properties():
    """C2: 500k / 1080p / SH3 forward+backward: gradient identities that hold for any size."""
    from gpu_utils import T, settings
    from gaussianmesh_amd import GaussianRasterizer, scenes
    P, W, H = 500_000, 1920, 1080
    sc = scenes.make_cloud(P, seed=1)
    cam = scenes.orbit_camera(7, 64, W, H)
    bg = np.array([0.2, 0.4, 0.6], np.float32)
    rast = GaussianRasterizer(settings(cam, bg, 3))
    leaves = [T(sc[k], True) for k in ("means", "opac", "shs", "scales", "rots")]
    m2d = torch.zeros((P, 3), device="cuda", requires_grad=True)
    color, radii = rast(leaves[0], m2d, leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
    wgt = torch.randn_like(color)
    (color * wgt).sum().backward()
    g = [l.grad for l in leaves] + [m2d.grad]
    assert all(torch.isfinite(x).all() for x in g)
    vis = radii > 0
    assert all((x[~vis] == 0).all() for x in g)                      # culled Gaussians get exactly zero gradient
    assert (m2d.grad[:, 2] == 0).all()
    # linearity of the backward in dL/dimage: grad(2w) = 2 grad(w)
    for l in leaves:
        l.grad = None
    m2d.grad = None
    color2, _ = rast(leaves[0], m2d, leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
    assert torch.equal(color2, color)
    (color2 * (2 * wgt)).sum().backward()
    for a, b in zip(g[:5], [l.grad for l in leaves]):
        assert (b - 2 * a).abs().max() <= 2e-3 * a.abs().max()      # float atomics: summation order differs between runs
    # directional derivative along sign(dL/dopacity) (central difference on the scalar loss; a random direction drowns in
    # the noise of the discrete alpha / transmittance thresholds flipping on a few thousand pixels)
    with torch.no_grad():
        d = torch.sign(g[1]) * 1e-3
        lp = (rast(leaves[0], m2d, leaves[1] + d, shs=leaves[2], scales=leaves[3], rotations=leaves[4])[0].double() * wgt.double()).sum()
        lm = (rast(leaves[0], m2d, leaves[1] - d, shs=leaves[2], scales=leaves[3], rotations=leaves[4])[0].double() * wgt.double()).sum()
    fd = float(lp - lm) / 2
    an = float((g[1].double() * d.double()).sum())
    assert abs(fd - an) <= 2e-2 * max(abs(an), 1e-3), (fd, an)


def test_c5_scale_smoke():
    """C5 as bench.py times it (bench.build_c5): 2 M mesh-bound + 1 M frozen background Gaussians, 3840x2160, through
    Trainer(sync_free=True, densify_stats=True, bg_gaussian=...) - fused activations, render with the shared SH storage,
    L1 + SSIM + mesh-restrict loss, backward, FusedAdam, densification statistics.  Twelve iterations alternating over two
    cameras: finite, the loss of each camera goes down, nothing is lost to an overflowing binning buffer, statistics fill."""
    import bench
    tr, cams, target, zero = bench.build_c5(ncams=32)
    N = tr.g.get_number
    losses = []
    for it in range(12):
        loss, pkg = tr.step(cams[(it % 2) * 5], target, zero)
        losses.append(loss)
    losses = [float(l) for l in losses]
    assert all(np.isfinite(losses)), losses
    assert losses[10] < losses[0] and losses[11] < losses[1], losses              # per camera: later visits are cheaper
    assert pkg["render"].shape == (3, 2160, 3840) and torch.isfinite(pkg["render"]).all()
    assert pkg["radii"].shape[0] == 3_000_000 and int((pkg["radii"] > 0).sum()) > 1_000_000 and pkg["scale"].shape[0] == N
    assert tr.optimizer.n_step == 12 and tr.redone <= 2                            # a redone iteration is repeated, never skipped
    assert float(tr.denom.max()) == 12 and float((tr.denom > 0).float().mean()) > 0.3 and float(tr.max_radii2D.max()) > 0
    assert torch.isfinite(tr.bc_gradient_accum).all()


def test_c2_full_size_gradients_vs_oracle(oracle):
    """BASELINE config C2 at its size: 500k Gaussians, 1920x1080, SH degree 3, dL/dimage = N(0,1) seed 1:
    radii, instance count and lists under the reference emission policy are bit-exact, the image passes the strict
    forward gate and all gradient tensors agree with the oracle to <= 1e-3 (reference: RAST/forward.cu:306-373,
    RAST/backward.cu:441-556)."""
    from gaussianmesh_amd import _lib, scenes
    from test_gpu_parity import _grads_gpu
    P, W, H = 500_000, 1920, 1080
    sc = scenes.make_cloud(P, seed=0)
    cam = scenes.orbit_camera(3, 64, W, H)
    bg = np.zeros(3, np.float32)
    fw = oracle.forward_full(sc, cam, bg, D=3)
    ex = _forward(sc, cam, bg, tile_cull=0)
    assert np.array_equal(ex["radii"], fw["geo"]["radii"]) and ex["R"] == fw["bins"]["R"]
    assert np.array_equal(ex["point_list"], fw["bins"]["point_list"]) and np.array_equal(ex["ranges"], fw["bins"]["ranges"])
    dpix = np.random.default_rng(1).normal(size=(3, H, W)).astype(np.float32)
    bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=3)
    color, radii, g = _grads_gpu(sc, cam, bg, dpix, 3, False, False)          # the product's default emission policy
    assert np.array_equal(radii, fw["geo"]["radii"])
    # the operator's forward of a step that needs a gradient runs with the backward's exponent (GM_FWD_EXACT_EXPONENT): same build under
    # the reference policy -> policies agree bit for bit; both builds pass the strict gate
    import ctypes as C
    lib = C.CDLL(_lib.lib()._name)
    lib.gm_debug_forward_exact_exponent(1)
    try:
        ex_exact = _forward(sc, cam, bg, tile_cull=0)
    finally:
        lib.gm_debug_forward_exact_exponent(0)
    assert np.array_equal(color, ex_exact["color"])
    assert_forward_gate(fw, color, W, H, 1e-4, "C2 (training forward)", plain_tol=2.5e-5)
    assert_forward_gate(fw, ex["color"], W, H, 1e-4, "C2", plain_tol=5e-5)
    _check_all_grads(g, bw, 1e-3)


def test_c3_bench_path_full_size_vs_oracle(oracle):
    """The exact path bench.py times (BASELINE config C3): bench.build_scene's 1 M-Gaussian torus cloud, 1920x1080,
    per-vertex (R, S) of the frame from gm_mesh_rs (vs oracle/mesh_oracle.py), gm_forward_0_deformed_async +
    gm_forward_1_geom, two frames / cameras.  Deformed cloud and colours vs the oracle's
    deform -> rotated SH colour (<= 1e-5 relative); then the oracle rasterizes the SAME deformed cloud: radii equal,
    num_rendered under the reference emission policy equal, sorted lists equal, strict image gate; the default policy's
    image is bit-identical to the reference policy's."""
    import bench
    from gpu_utils import T, set_policy
    from gaussianmesh_amd import rasterizer as Rz, scenes
    from gaussianmesh_amd.deform import mesh_rs, pack_mesh_state
    from oracle import mesh_oracle
    P, W, H, F = 1_000_000, 1920, 1080, 64
    host = bench.build_scene(P, W, H, F)
    faces_t = T(host["faces"], dtype=torch.int32)
    g = {k: T(host[k]) for k in ("weights", "pos", "cov", "opac", "shs", "verts")}
    g["tri"] = T(host["tri"], dtype=torch.int32)
    bg = np.ones(3, np.float32)
    for t, k in ((3, 3), (40, 17)):
        cam = scenes.orbit_camera(k, F, W, H)
        ct = {n: T(cam[n]) for n in ("view", "proj", "campos")}
        state = mesh_rs(g["verts"], T(host["mesh"][t][:, 0:3]), faces_t, want_state=True)[2]      # as bench.py's frame_state()
        ms = state.cpu().numpy()
        Ro, So = mesh_oracle.mesh_rs(host["verts"], host["mesh"][t][:, 0:3], host["faces"])
        assert np.abs(ms[:, 3:12].reshape(-1, 3, 3) - Ro).max() <= 5e-5 and np.abs(ms[:, 12:21].reshape(-1, 3, 3) - So).max() <= 5e-5
        assert np.abs(ms[:, 3:21] - host["mesh"][t][:, 3:21]).max() <= 0.25       # sanity: first-order (edge length) approximation of the analytic Jacobian's factors
        packed = pack_mesh_state(state, g["verts"])
        out = {}
        for mode in (0, 2):
            set_policy(mode)
            h = Rz.forward_deformed_begin(T(bg), g["tri"], g["weights"], packed, g["cov"], g["pos"], g["shs"], g["opac"], ct["view"],
                                          ct["proj"], cam["tanx"], cam["tany"], H, W, 3, ct["campos"], want_deformed=True)
            nr, color, radii, geom, binning, img = h.finish()
            torch.cuda.synchronize()
            out[mode] = (nr, color.cpu().numpy(), radii.cpu().numpy(), [x.cpu().numpy() for x in h.deformed])
            if mode == 0:
                from gpu_utils import _view
                from gaussianmesh_amd import _lib
                plist = _view(binning, _lib.lib().gm_binning_field(binning.data_ptr(), nr, W, H, 0, b"pairs"), 2 * nr, torch.int32).astype(np.uint32)[1::2]
        set_policy(2)
        assert np.array_equal(out[0][1], out[2][1]) and np.array_equal(out[0][2], out[2][2]) and out[2][0] < out[0][0]
        pos_d, cov6_d, rgb_d = out[2][3]
        # a20 / a21 against the oracle (float64 algebra, see gm_oracle.c orc_deform)
        dV = ms[:, 0:3] - host["verts"]
        p_ref, c_ref, r_ref = oracle.deform(host["tri"], host["weights"], dV, ms[:, 3:12].reshape(-1, 3, 3), ms[:, 12:21].reshape(-1, 3, 3),
                                            host["cov"], host["pos"])
        rgb_ref = oracle.sh_colors_rotated(p_ref, cam["campos"], r_ref, host["shs"], deg=3)
        assert np.abs(pos_d - p_ref).max() <= 1e-5 * np.abs(p_ref).max()
        assert np.abs(cov6_d - scenes.strip_symmetric(c_ref)).max() <= 1e-5 * np.abs(c_ref).max()
        assert np.abs(rgb_d - rgb_ref).max() <= 2e-5
        # a2-a13 on the same deformed cloud
        sc = dict(means=pos_d, opac=host["opac"], colors_precomp=rgb_d, cov3D_precomp=cov6_d)
        fw = oracle.forward_full(sc, cam, bg, D=3, use_precomp_cov=True, use_precomp_color=True)
        assert np.array_equal(out[0][2], fw["geo"]["radii"])
        assert out[0][0] == fw["bins"]["R"] and np.array_equal(plist, fw["bins"]["point_list"])
        assert_forward_gate(fw, out[2][1], W, H, 1e-4, "C3 frame %d" % t, plain_tol=5e-5)


def test_c3_batch_of_four_full_size_equals_the_single_frame_path():
    """The launch chain bench.py's headline runs since round 6 - gm_mesh_rs_packed_batch + gm_forward_deformed_batch_async, four frames
    per chain, packed [N,6] covariances, image-only, work hint - at BASELINE's size (1 M Gaussians, 1920x1080): every frame's image, radii
    and instance count are, bit for bit, the single-frame path's (gm_mesh_rs_packed + gm_forward_0_deformed_async + gm_forward_1_geom),
    which test_c3_bench_path_full_size_vs_oracle holds against the oracle for two of these four (mesh frame, camera) pairs."""
    import bench
    from gpu_utils import T
    from gaussianmesh_amd import rasterizer as Rz, scenes
    from gaussianmesh_amd.deform import mesh_rs_packed, mesh_rs_packed_batch, pack_cov6, vertex_face_adjacency
    P, W, H, F = 1_000_000, 1920, 1080, 64
    host = bench.build_scene(P, W, H, F)
    g = {k: T(host[k]) for k in ("weights", "pos", "cov", "opac", "shs", "verts")}
    g["tri"] = T(host["tri"], dtype=torch.int32)
    faces = T(host["faces"], dtype=torch.int32)
    off, adj = vertex_face_adjacency(host["faces"], host["verts"].shape[0])
    adjacency = (torch.tensor(off, device="cuda"), torch.tensor(adj, device="cuda"))
    cov6 = pack_cov6(g["cov"])
    bg = T(np.ones(3, np.float32))
    pairs = [(3, 3), (40, 17), (10, 50), (63, 0)]
    cams = []
    for _, k in pairs:
        cam = scenes.orbit_camera(k, F, W, H)
        cams.append(dict(view=T(cam["view"]), proj=T(cam["proj"]), campos=T(cam["campos"]), tanx=cam["tanx"], tany=cam["tany"]))
    v1 = [T(np.ascontiguousarray(host["mesh"][t][:, 0:3])) for t, _ in pairs]
    single = []
    for (t, _), cm, v in zip(pairs, cams, v1):
        packed = mesh_rs_packed(g["verts"], v, faces, adjacency)
        nr, color, radii, *_ = Rz.forward_deformed_begin(bg, g["tri"], g["weights"], packed, cov6, g["pos"], g["shs"], g["opac"], cm["view"], cm["proj"],
                                                         cm["tanx"], cm["tany"], H, W, 3, cm["campos"], False).finish(image_only=True)
        single.append((nr, color.clone(), radii.clone(), packed.clone()))
    ws = [Rz.RasterWorkspace() for _ in pairs]
    for w_ in ws:
        w_.capacity = int(1.05 * max(s[0] for s in single))
    hint = Rz.new_work_hint(W, H, bg.device)
    for rep in range(2):                                   # the second chain runs with the work hint the first one left
        tables = mesh_rs_packed_batch(g["verts"], v1, faces, adjacency)
        hs = Rz.forward_deformed_batch(bg, g["tri"], g["weights"], tables, cov6, g["pos"], g["shs"], g["opac"], cams, H, W, 3, ws, image_only=True, work_hint=hint)
        for k, h in enumerate(hs):
            ok, nr = h.check()
            assert ok and nr == single[k][0], (rep, k, ok, nr, single[k][0])
            assert torch.equal(tables[k], single[k][3]), (rep, k)
            assert torch.equal(h.radii, single[k][2]) and torch.equal(h.color, single[k][1]), (rep, k)


def test_4k_policy3_lists_image_and_gradients_vs_oracle(oracle):
    """BASELINE config C5's list geometry against the ORACLE, not only through properties (review round 5, item 5c): 200 k Gaussians on
    a 3840x2160 grid, where the product's default emission policy is 3 - instances per 64-px parent tile, 2040 list tiles, one 11-bit
    tile pass, keys carrying a 4 x 4 mask of 16-px children.
      * reference policy at 4K: radii, instance count, sorted list and ranges bit-identical to the oracle's (32 400 tiles: the two-pass
        tile sort + tile_ranges_kernel);
      * policy 3: every instance SOME pixel accepts according to the oracle (orc_instance_needed over all 6 M reference instances) is
        present with its child bit, nothing outside the reference's instance set is, parent lists are (depth, id) ordered; image and
        final_T bit-identical to the reference-policy run; strict image gate against the oracle;
      * all gradient tensors of the operator (which runs under policy 3 here) within 1e-3 of the oracle's + the element-wise gate."""
    from gaussianmesh_amd import rasterizer as Rz, scenes
    from test_gpu_parity import _grads_gpu
    P, W, H = 200_000, 3840, 2160
    assert Rz.auto_emission_policy(W, H) == 3
    sc = scenes.make_cloud(P, seed=2)
    cam = scenes.orbit_camera(11, 64, W, H)
    bg = np.array([0.1, 0.3, 0.2], np.float32)
    fw = oracle.forward_full(sc, cam, bg, D=3)
    ex = _forward(sc, cam, bg, tile_cull=0)
    assert np.array_equal(ex["radii"], fw["geo"]["radii"]) and ex["R"] == fw["bins"]["R"]
    assert np.array_equal(ex["point_list"], fw["bins"]["point_list"]) and np.array_equal(ex["ranges"], fw["bins"]["ranges"])
    cu = _forward(sc, cam, bg, tile_cull=3)
    _check_list_invariants(cu, cam, 3)
    assert np.array_equal(cu["radii"], ex["radii"]) and cu["R"] < 0.5 * ex["R"]
    assert np.array_equal(cu["color"], ex["color"]) and np.array_equal(cu["final_T"], ex["final_T"])
    # list geometry: (16-px tile, Gaussian) pairs the policy-3 keys stand for, against the reference instance set and the needed subset
    gx = (W + 15) // 16
    pgx = (gx + 3) >> 2
    tile_ref = (fw["bins"]["keys"] >> np.uint64(32)).astype(np.int64)
    ref_pairs = tile_ref * P + fw["bins"]["point_list"].astype(np.int64)
    needed = oracle.instance_needed(W, H, fw["bins"], fw["geo"]).astype(bool)
    par = cu["tile_keys"].astype(np.int64); gid = cu["point_list"].astype(np.int64); mask = cu["child_mask"].astype(np.int64)
    px, py = par % pgx, par // pgx
    got = []
    for c in range(16):
        sel = (mask >> c) & 1 == 1
        t16 = ((py[sel] << 2) + (c >> 2)) * gx + (px[sel] << 2) + (c & 3)
        got.append(t16 * P + gid[sel])
    got = np.concatenate(got)
    assert len(np.unique(got)) == len(got)                                    # no (tile, Gaussian) pair twice
    assert np.isin(got, ref_pairs).all(), "policy 3 emitted an instance outside the reference's rectangle set"
    missing = ~np.isin(ref_pairs[needed], got)
    assert not missing.any(), "%d instances some pixel accepts are missing under policy 3" % int(missing.sum())
    print("4K policy 3: %d instances for %d reference instances (%d of them needed by some pixel); %d (tile, Gaussian) pairs covered" % (
        cu["R"], ex["R"], int(needed.sum()), len(got)))
    assert_forward_gate(fw, cu["color"], W, H, 1e-4, "4K policy 3", plain_tol=5e-5)
    dpix = np.random.default_rng(1).normal(size=(3, H, W)).astype(np.float32)
    bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=3)
    color, radii, g = _grads_gpu(sc, cam, bg, dpix, 3, False, False)            # the operator: auto policy = 3 at this size
    assert np.array_equal(radii, fw["geo"]["radii"])
    assert_forward_gate(fw, color, W, H, 1e-4, "4K policy 3 (training forward)", plain_tol=2.5e-5)
    _check_all_grads(g, bw, 1e-3)


def _front_T_after_backward(run_backward, H, W):
    """the transmittance in front of every pixel's first entry as the backward walk reconstructs it (final_T divided by 1 - alpha of each
    entry it takes): exactly 1, up to the rounding of a few hundred reciprocals, when it took the entries the forward blended; off by
    >= 0.39 % per entry the halves disagree about.  The plane comes out of the backward kernel itself (gm_debug_backward_front_T)."""
    import ctypes as C
    from gaussianmesh_amd import _lib
    lib = C.CDLL(_lib.lib()._name)
    front = torch.ones((H, W), dtype=torch.float32, device="cuda")
    lib.gm_debug_backward_front_T(C.c_void_p(front.data_ptr()))
    try:
        run_backward()
        torch.cuda.synchronize()
    finally:
        lib.gm_debug_backward_front_T(None)
    assert torch.isfinite(front).all()
    return (front - 1.0).abs()


def test_training_step_halves_take_the_same_entries_c2_size():
    """In the reference backward.cu repeats forward.cu's expression, so the two halves of a training step take the same alpha >= 1/255
    decision for every (entry, pixel).  Here the forward of a step that needs a gradient runs with GM_FWD_EXACT_EXPONENT - the blend
    evaluates its exponents with the backward kernel's per-pixel expression instead of the matrix-core polynomial - and the halves agree
    EXACTLY: at BASELINE config C2's size (500 k Gaussians, 1920x1080, SH 3) no covered pixel's walk deviates beyond reciprocal rounding."""
    from gpu_utils import T, settings
    from gaussianmesh_amd import GaussianRasterizer, scenes
    sc = scenes.make_cloud(500_000, seed=0)
    W, H = 1920, 1080
    cam = scenes.orbit_camera(3, 64, W, H)
    bg = np.zeros(3, np.float32)
    means = T(sc["means"], True); m2d = torch.zeros_like(means, requires_grad=True)
    rast = GaussianRasterizer(settings(cam, bg, 3))
    color, radii = rast(means, m2d, T(sc["opac"], True), shs=T(sc["shs"], True), scales=T(sc["scales"], True), rotations=T(sc["rots"], True))
    dev = _front_T_after_backward(lambda: (color * torch.randn_like(color)).sum().backward(), H, W)
    covered = int((color.detach().sum(0) > 0).sum())
    print("training step (exact-exponent forward), C2 size: largest deviation of the reconstructed front transmittance %.3g over %d covered "
          "pixels, median %.2g" % (float(dev.max()), covered, float(dev.median())))
    assert covered > 0.2 * W * H
    assert float(dev.max()) <= 1e-3 and float(dev.median()) <= 2e-5            # rounding of the reciprocals only: no entry taken by one half alone


def test_backward_after_a_matrix_core_forward_c2_size():
    """The low-level pair rasterize_forward (no flag: exponents from the polynomial on the matrix core, ~1e-5) + rasterize_backward
    (per pixel, ~5e-7): the halves CAN disagree about an entry that sits on the alpha = 1/255 threshold.  MEASURED at C2's size:
    6 of 1.3 M covered pixels, each a 0.4 % weight - why the operator sets GM_FWD_EXACT_EXPONENT when a gradient is required."""
    from gpu_utils import T
    from gaussianmesh_amd import rasterizer as Rz, scenes
    sc = scenes.make_cloud(500_000, seed=0)
    W, H = 1920, 1080
    cam = scenes.orbit_camera(3, 64, W, H)
    bg = T(np.zeros(3, np.float32))
    a = (bg, T(sc["means"]), None, T(sc["opac"]), T(sc["scales"]), T(sc["rots"]), 1.0, None, T(cam["view"]), T(cam["proj"]), cam["tanx"], cam["tany"])
    nr, color, radii, geom, binning, img = Rz.rasterize_forward(*a, H, W, T(sc["shs"]), 3, T(cam["campos"]), False, False)
    g = torch.randn_like(color)
    dev = _front_T_after_backward(lambda: Rz.rasterize_backward(bg, a[1], radii, None, a[4], a[5], 1.0, None, a[8], a[9], cam["tanx"], cam["tany"], g,
                                                                T(sc["shs"]), 3, T(cam["campos"]), geom, nr, binning, img, False), H, W)
    covered = int((color.sum(0) > 0).sum())
    off = int((dev > 1e-3).sum())
    print("front-of-list transmittance after the backward walk behind a matrix-core forward (C2 size): %d of %d covered pixels off by more "
          "than 1e-3 (largest %.3g); median deviation of the others %.2g" % (off, covered, float(dev.max()), float(dev[dev <= 1e-3].median())))
    assert covered > 0.2 * W * H
    assert float(dev[dev <= 1e-3].max()) <= 1e-3 and float(dev[dev <= 1e-3].median()) <= 2e-5         # rounding of the reciprocals only
    assert off <= 64, off                                  # pixels with an entry on the threshold (measured: 6 of 1.3 M covered pixels), each a 0.4 % weight
    assert float(dev.max()) <= 0.05                        # never more than a handful of threshold entries on one pixel
