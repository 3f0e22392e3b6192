"""-m gpu parity tests: the HIP path (through the python operator API and the C ABI) against the CPU oracle.

Bars (BASELINE.json north_star): bit-exact for integer / index work (radii, tile counts, instance lists, tile
ranges, depth order) and for the per-Gaussian geometry under the shared arithmetic contract; forward colour
<= 1e-4 max-abs per pixel; gradients <= 1e-3 relative (to the tensor's max magnitude).
"""
import numpy as np
import pytest
import torch

from helpers import assert_contributor_counts, assert_forward_gate, assert_grads_elementwise, small_scene

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-4
GRAD_RTOL = 1e-3


@pytest.fixture(autouse=True)
def _default_emission_policy():
    from gaussianmesh_amd import rasterizer
    rasterizer.set_default_emission_policy(2)
    yield
    rasterizer.set_default_emission_policy(2)

MODES = [(False, False), (True, False), (False, True), (True, True)]


def _check_geometry(orc, st, fw, scene, use_precomp_color):
    g = fw["geo"]
    vis = g["radii"] > 0
    assert np.array_equal(st["radii"], g["radii"])
    assert np.array_equal(st["tiles"], g["tiles"])
    assert st["R"] == fw["bins"]["R"]
    sp = st["splat"]
    assert np.array_equal(sp[vis, 0:2].view(np.uint32), g["xy"][vis].view(np.uint32))
    con = np.concatenate([sp[:, 2:4], sp[:, 4:5]], 1)
    assert np.array_equal(con[vis].view(np.uint32), g["conic_op"][vis, :3].view(np.uint32))
    assert np.array_equal(sp[vis, 5], g["conic_op"][vis, 3])
    assert np.array_equal(st["depth_key"][vis], g["depths"][vis].view(np.uint32)) and (st["depth_key"][~vis] == 0xFFFFFFFF).all()
    rgb = np.concatenate([sp[:, 6:8], sp[:, 8:9]], 1)
    assert np.array_equal(rgb[vis].view(np.uint32), g["rgb"][vis].view(np.uint32))
    if not use_precomp_color:
        cl = np.stack([(st["clamped"] >> c) & 1 for c in range(3)], 1)
        assert np.array_equal(cl[vis], g["clamped"][vis])
    assert np.array_equal(st["point_list"], fw["bins"]["point_list"])
    assert np.array_equal(st["tile_keys"], (fw["bins"]["keys"] >> np.uint64(32)).astype(np.uint32))
    assert np.array_equal(st["ranges"], fw["bins"]["ranges"])


@pytest.mark.parametrize("use_precomp_cov,use_precomp_color", MODES)
@pytest.mark.parametrize("D", [0, 3])
def test_forward_small(oracle, use_precomp_cov, use_precomp_color, D):
    from gpu_utils import forward_state
    sc, cam = small_scene(P=600, W=70, H=50, seed=3, D=D)
    bg = np.array([0.1, 0.4, 0.8], np.float32)
    fw = oracle.forward_full(sc, cam, bg, D=D, use_precomp_cov=use_precomp_cov, use_precomp_color=use_precomp_color)
    st = forward_state(sc, cam, bg, D=D, use_precomp_cov=use_precomp_cov, use_precomp_color=use_precomp_color)
    _check_geometry(oracle, st, fw, sc, use_precomp_color)
    if not use_precomp_cov:
        vis = fw["geo"]["radii"] > 0
        assert np.array_equal(st["cov3D"][vis].view(np.uint32), fw["geo"]["cov3D"][vis].view(np.uint32))
    assert np.abs(st["color"] - fw["color"]).max() <= FWD_TOL
    assert np.abs(st["final_T"] - fw["final_T"]).max() <= FWD_TOL
    assert_contributor_counts(fw, st["n_contrib"], st["color"], cam["W"], cam["H"], "small D=%d" % D)     # every differing pixel accounted for


@pytest.mark.parametrize("D", [1, 2])
def test_forward_sh_degrees(oracle, D):
    from gpu_utils import forward_state
    sc, cam = small_scene(P=400, W=64, H=64, seed=5, D=D)
    bg = np.zeros(3, np.float32)
    fw = oracle.forward_full(sc, cam, bg, D=D)
    st = forward_state(sc, cam, bg, D=D)
    _check_geometry(oracle, st, fw, sc, False)
    assert np.abs(st["color"] - fw["color"]).max() <= FWD_TOL


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_forward_medium_ragged(oracle, mode):
    """C1-like case (10k Gaussians) at a size that is not a multiple of the tile, long lists (multi-batch), under every
    emission policy (0 = reference lists, 1 = culled 16-px lists, 2 / 3 = culled lists per 32- / 64-px parent tile)."""
    from gpu_utils import forward_state
    from gaussianmesh_amd import scenes
    sc = scenes.make_cloud(10000, seed=0, scale_lo=0.02, scale_hi=0.25)
    cam = scenes.orbit_camera(2, 9, 250, 130, radius=7.0)
    bg = np.array([1, 1, 1], np.float32)
    fw = oracle.forward_full(sc, cam, bg, D=3)
    st = forward_state(sc, cam, bg, D=3, tile_cull=mode)
    if mode == 0:
        _check_geometry(oracle, st, fw, sc, False)
    else:
        assert np.array_equal(st["radii"], fw["geo"]["radii"])
    # strict gate: every pixel <= 1e-4, except pixels where an entry provably sits on a decision threshold
    # (alpha = 1/255, T(1-alpha) = 1e-4, power = 0) within its own rounding distance (helpers.account_outlier_pixels)
    assert_forward_gate(fw, st["color"], cam["W"], cam["H"], FWD_TOL, "policy %d" % mode)


def _grads_gpu(sc, cam, bg, dpix, D, use_precomp_cov, use_precomp_color, mod=1.0):
    from gpu_utils import T, settings
    from gaussianmesh_amd import GaussianRasterizer
    means = T(sc["means"], True); opac = T(sc["opac"], True)
    m2d = torch.zeros_like(means, requires_grad=True)
    kw = {}
    leaves = dict(means=means, opac=opac, m2d=m2d)
    if use_precomp_color:
        kw["colors_precomp"] = leaves["colors"] = T(sc["colors_precomp"], True)
    else:
        kw["shs"] = leaves["shs"] = T(sc["shs"], True)
    if use_precomp_cov:
        kw["cov3D_precomp"] = leaves["cov"] = T(sc["cov3D_precomp"], True)
    else:
        kw["scales"] = leaves["scales"] = T(sc["scales"], True)
        kw["rotations"] = leaves["rots"] = T(sc["rots"], True)
    rast = GaussianRasterizer(settings(cam, bg, D, mod))
    color, radii = rast(means, m2d, opac, **kw)
    (color * T(dpix)).sum().backward()
    torch.cuda.synchronize()
    return color.detach().cpu().numpy(), radii.cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in leaves.items()}


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _grad_gate(got, ref, what=""):
    """max-norm gate (north_star: <= 1e-3 relative to the tensor's magnitude) AND the element-wise one: every entry of at least
    1e-3 x max agrees to 1e-2 of itself"""
    assert _rel(got, ref) <= GRAD_RTOL, (what, _rel(got, ref))
    assert_grads_elementwise(got, ref, what)


@pytest.mark.parametrize("mode", [0, 1, 3])
def test_backward_under_every_emission_policy(oracle, mode):
    from gaussianmesh_amd import rasterizer
    rasterizer.set_default_emission_policy(mode)
    test_backward_medium(oracle)


@pytest.mark.parametrize("D", [0, 1, 2])
def test_backward_below_the_full_sh_degree(oracle, D):
    """SH input at active degree 0..2 (the first 3000 training iterations, train_mesh_gaussian.py:70-71): both preprocess kernels
    fetch only the leading (D+1)^2 coefficients of every 16-coefficient row; image, radii and every gradient against the oracle,
    and dL/dSH of the coefficients above the degree is exactly zero (FusedAdam's `active` relies on it)."""
    from gaussianmesh_amd import scenes
    sc = scenes.make_cloud(3000, seed=21 + D, scale_lo=0.02, scale_hi=0.25, D=3)
    cam = scenes.orbit_camera(2, 7, 144, 96, radius=7.0)
    bg = np.array([0.2, 0.6, 0.1], np.float32)
    dpix = np.random.default_rng(D).normal(size=(3, cam["H"], cam["W"])).astype(np.float32)
    fw = oracle.forward_full(sc, cam, bg, D=D)
    bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=D)
    color, radii, g = _grads_gpu(sc, cam, bg, dpix, D, False, False)
    assert np.array_equal(radii, fw["geo"]["radii"])
    assert_forward_gate(fw, color, cam["W"], cam["H"], FWD_TOL, "SH degree %d" % D)
    nc = (D + 1) ** 2
    assert np.abs(g["shs"][:, nc:]).max() == 0.0 and np.abs(g["shs"][:, :nc]).max() > 0
    for name, ref in (("means", bw["dmean3D"]), ("opac", bw["dopacity"]), ("shs", bw["dsh"]), ("scales", bw["dscale"]), ("rots", bw["drot"])):
        _grad_gate(g[name].reshape(ref.shape), ref, "D=%d d/d%s" % (D, name))


@pytest.mark.parametrize("use_precomp_cov,use_precomp_color", MODES)
def test_backward_small(oracle, use_precomp_cov, use_precomp_color):
    D = 3
    sc, cam = small_scene(P=500, W=70, H=50, seed=7, D=D)
    bg = np.array([0.3, 0.2, 0.7], np.float32)
    rng = np.random.default_rng(1)
    dpix = rng.normal(size=(3, cam["H"], cam["W"])).astype(np.float32)
    fw = oracle.forward_full(sc, cam, bg, D=D, use_precomp_cov=use_precomp_cov, use_precomp_color=use_precomp_color)
    bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=D, use_precomp_cov=use_precomp_cov, use_precomp_color=use_precomp_color)
    color, radii, g = _grads_gpu(sc, cam, bg, dpix, D, use_precomp_cov, use_precomp_color)
    assert np.array_equal(radii, fw["geo"]["radii"])
    assert np.abs(color - fw["color"]).max() <= FWD_TOL
    _grad_gate(g["means"], bw["dmean3D"], 'g["means"]')
    _grad_gate(g["m2d"][:, :2], bw["dmean2D"][:, :2], 'g["m2d"][:, :2]')
    _grad_gate(g["opac"].reshape(-1), bw["dopacity"], 'g["opac"].reshape(-1)')
    if use_precomp_color:
        _grad_gate(g["colors"], bw["dcolor"], 'g["colors"]')
    else:
        _grad_gate(g["shs"], bw["dsh"], 'g["shs"]')
    if use_precomp_cov:
        _grad_gate(g["cov"], bw["dcov3D"], 'g["cov"]')
    else:
        _grad_gate(g["scales"], bw["dscale"], 'g["scales"]')
        _grad_gate(g["rots"], bw["drot"], 'g["rots"]')


def test_backward_medium(oracle):
    from gaussianmesh_amd import scenes
    D = 2
    sc = scenes.make_cloud(6000, seed=11, scale_lo=0.02, scale_hi=0.2, D=D)
    cam = scenes.orbit_camera(1, 5, 200, 120, radius=7.0)
    bg = np.zeros(3, np.float32)
    dpix = np.random.default_rng(2).normal(size=(3, cam["H"], cam["W"])).astype(np.float32)
    fw = oracle.forward_full(sc, cam, bg, D=D)
    bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=D)
    color, radii, g = _grads_gpu(sc, cam, bg, dpix, D, False, False)
    assert np.array_equal(radii, fw["geo"]["radii"])
    for name, ref in [("means", bw["dmean3D"]), ("shs", bw["dsh"]), ("scales", bw["dscale"]), ("rots", bw["drot"])]:
        _grad_gate(g[name], ref, name)
    _grad_gate(g["opac"].reshape(-1), bw["dopacity"], 'g["opac"].reshape(-1)')


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("case", ["small", "medium", "huge_splats"])
def test_tile_culling_is_exact(oracle, case, mode):
    """Culled emission policies vs the reference policy: per 16-px tile the entries the blend kernel takes (the
    parent tile's list filtered by the tile's bit(s) in the child mask: one per tile, under policy 2 one per 8x8
    quadrant) are the reference list minus instances no pixel of the tile accepts; images are bit-identical."""
    from gpu_utils import forward_state
    from gaussianmesh_amd import scenes
    if case == "small":
        sc, cam = small_scene(P=600, W=70, H=50, seed=3)
    elif case == "medium":
        sc = scenes.make_cloud(10000, seed=0, scale_lo=0.02, scale_hi=0.25); cam = scenes.orbit_camera(2, 9, 250, 130, radius=7.0)
    else:                                       # rectangles of more than 64 tiles (chunked path) and needle-shaped splats
        sc = scenes.make_cloud(300, seed=5, scale_lo=0.02, scale_hi=2.5); cam = scenes.orbit_camera(1, 5, 320, 200, radius=7.0)
    bg = np.array([0.3, 0.6, 0.1], np.float32)
    fw = oracle.forward_full(sc, cam, bg, D=3)
    ex = forward_state(sc, cam, bg, D=3, tile_cull=0)
    cu = forward_state(sc, cam, bg, D=3, tile_cull=mode)
    assert np.array_equal(ex["point_list"], fw["bins"]["point_list"])
    assert np.array_equal(cu["radii"], ex["radii"])
    assert cu["R"] <= ex["R"]
    # images: same per-pixel arithmetic on the same accepted entries -> identical bits
    assert np.array_equal(cu["color"], ex["color"])
    assert np.array_equal(cu["final_T"], ex["final_T"])
    # parent lists: keys sorted, ranges delimit them
    sh = max(mode - 1, 0); m = (1 << sh) - 1
    gx, gy = (cam["W"] + 15) // 16, (cam["H"] + 15) // 16
    pgx = (gx + m) >> sh
    assert np.all(np.diff(cu["tile_keys"].astype(np.int64)) >= 0)
    for p in range(cu["ranges"].shape[0]):
        b0, b1 = cu["ranges"][p]
        assert np.all(cu["tile_keys"][b0:b1] == p)
    assert sum(int(b1 - b0) for b0, b1 in cu["ranges"]) == cu["R"]
    # list structure: per 16-px tile, the taken entries are an order-preserving subsequence of the reference list ...
    needed = oracle.instance_needed(cam["W"], cam["H"], fw["bins"], fw["geo"]).astype(bool)
    kept = np.zeros(ex["R"], bool)
    for t in range(gx * gy):
        tx, ty = t % gx, t // gx
        p = (ty >> sh) * pgx + (tx >> sh); c = ((ty & m) << sh) | (tx & m)
        a0, a1 = ex["ranges"][t]; b0, b1 = cu["ranges"][p]
        ref = ex["point_list"][a0:a1]
        if mode == 2:                            # policy 2: one bit per 8x8 quadrant of the parent, qy * 4 + qx; the tile's four quadrants
            cx, cy = tx & 1, ty & 1
            tbits = 0x33 << (8 * cy + 2 * cx)
        else:
            tbits = 1 << c
        sub = cu["point_list"][b0:b1][(cu["child_mask"][b0:b1] & tbits) != 0]
        j = 0
        for g in sub:                            # two-pointer subsequence check (ids are unique within a tile)
            while j < len(ref) and ref[j] != g:
                j += 1
            assert j < len(ref), "taken entries are not a subsequence of the reference list in tile %d" % t
            kept[a0 + j] = True
            j += 1
    # ... that contains every instance some pixel accepts (conservative), and drops a good share of the others
    assert not (needed & ~kept).any()
    if mode == 1:
        tile_of = (fw["bins"]["keys"] >> np.uint64(32)).astype(np.uint32)
        assert np.array_equal(cu["tile_keys"], tile_of[kept]) and np.all(cu["child_mask"] == 1)
    if case != "small":
        assert cu["R"] < (0.8, 0.5, 0.35)[mode - 1] * ex["R"]


def test_tile_culling_gradients_match_reference_policy(oracle):
    from gaussianmesh_amd import rasterizer
    D = 3
    sc, cam = small_scene(P=500, W=70, H=50, seed=7, D=D)
    bg = np.array([0.3, 0.2, 0.7], np.float32)
    dpix = np.random.default_rng(1).normal(size=(3, cam["H"], cam["W"])).astype(np.float32)
    rasterizer.set_default_emission_policy(0)
    c0, r0, g0 = _grads_gpu(sc, cam, bg, dpix, D, False, False)
    for mode in (1, 2, 3):
        rasterizer.set_default_emission_policy(mode)
        c1, r1, g1 = _grads_gpu(sc, cam, bg, dpix, D, False, False)
        assert np.array_equal(c0, c1) and np.array_equal(r0, r1)
        for k in g0:
            assert _rel(g1[k], g0[k]) <= 1e-5, (mode, k)       # same terms, only the float-atomic summation order differs


def test_edge_cases(oracle):
    from gpu_utils import forward_state, T, settings
    from gaussianmesh_amd import GaussianRasterizer
    bg = np.array([0.25, 0.5, 0.75], np.float32)
    # all Gaussians behind the camera -> image == background, R == 0
    sc, cam = small_scene(P=64, W=33, H=17, seed=1, behind=False)
    sc["means"] = (np.asarray(cam["campos"])[None, :] * 1.5 + 0.01 * sc["means"]).astype(np.float32)
    st = forward_state(sc, cam, bg, D=3)
    assert st["R"] == 0 and (st["radii"] == 0).all()
    assert np.allclose(st["color"], bg[:, None, None])
    # P == 0
    rast = GaussianRasterizer(settings(cam, bg, 0))
    e = torch.zeros((0, 3), device="cuda")
    color, radii = rast(e, e, torch.zeros((0, 1), device="cuda"), colors_precomp=e, cov3D_precomp=torch.zeros((0, 6), device="cuda"))
    assert radii.numel() == 0 and np.allclose(color.cpu().numpy(), bg[:, None, None])
    # argument validation as in the reference module (__init__.py:146-150)
    m = T(sc["means"])
    with pytest.raises(Exception):
        rast(m, m, T(sc["opac"]), shs=None, colors_precomp=None, scales=T(sc["scales"]), rotations=T(sc["rots"]))
    with pytest.raises(Exception):
        rast(m, m, T(sc["opac"]), shs=T(sc["shs"]), scales=T(sc["scales"]), rotations=T(sc["rots"]), cov3D_precomp=T(sc["cov3D_precomp"]))


def test_mark_visible(oracle):
    from gpu_utils import T, settings
    from gaussianmesh_amd import GaussianRasterizer
    sc, cam = small_scene(P=500, seed=2)
    vis = GaussianRasterizer(settings(cam, np.zeros(3), 0)).markVisible(T(sc["means"])).cpu().numpy()
    assert np.array_equal(vis, oracle.mark_visible(sc["means"], cam["view"], cam["proj"]))


def test_large_splats_match_the_oracle(oracle):
    """Splats whose tile rectangle holds more than 64 tiles take the wave-wide emission path (gm_binning.hip, (2)): instance
    count and image against the oracle under the reference policy, images identical under the others."""
    from gpu_utils import T
    from gaussianmesh_amd import rasterizer as Rz
    W, H = 400, 240
    sc, cam = small_scene(P=600, W=W, H=H, seed=12, scale_lo=0.05, scale_hi=0.6, behind=False)
    sc["scales"][:120] *= 8.0                                    # 120 splats that cover a good part of the screen
    bg = np.array([0.2, 0.1, 0.3], np.float32)
    ct = {n: T(cam[n]) for n in ("view", "proj", "campos")}
    args = (T(bg), T(sc["means"]), None, T(sc["opac"]), T(sc["scales"]), T(sc["rots"]), 1.0, None, ct["view"], ct["proj"], cam["tanx"], cam["tany"],
            H, W, T(sc["shs"]), 3, ct["campos"])
    fw = oracle.forward_full(sc, cam, bg, D=3)
    assert (fw["geo"]["radii"] > 56).sum() >= 100
    ref = Rz.rasterize_forward(*args, False, False, emission_policy=0)
    assert ref[0] == fw["bins"]["R"] and np.array_equal(ref[2].cpu().numpy(), fw["geo"]["radii"])
    assert_forward_gate(fw, ref[1].cpu().numpy(), W, H, FWD_TOL, "large splats")
    for pol in (1, 2, 3):
        out = Rz.rasterize_forward(*args, False, False, emission_policy=pol)
        assert torch.equal(out[1], ref[1]) and out[0] < ref[0]


def test_prefiltered_contract(oracle):
    """prefiltered=True promises that no Gaussian is frustum-culled; the reference traps the kernel when one is
    (RAST/auxiliary.h:153-159).  Here the violation is an error with the reference's message, and a cloud that keeps the promise
    renders exactly as with prefiltered=False."""
    from gpu_utils import T
    from gaussianmesh_amd import GaussianRasterizationSettings, GaussianRasterizer
    from gaussianmesh_amd._lib import GmeshError
    W, H = 48, 40
    sc, cam = small_scene(P=400, W=W, H=H, behind=True)
    def run(means, prefiltered):
        rs = GaussianRasterizationSettings(H, W, cam["tanx"], cam["tany"], T(np.zeros(3)), 1.0, T(cam["view"]), T(cam["proj"]), 3,
                                           T(cam["campos"]), prefiltered, False)
        n = means.shape[0]
        return GaussianRasterizer(rs)(T(means), torch.zeros((n, 3), device="cuda"), T(sc["opac"][:n]), shs=T(sc["shs"][:n]), scales=T(sc["scales"][:n]),
                                      rotations=T(sc["rots"][:n]))
    with pytest.raises(GmeshError, match="filtered although prefiltered"):
        run(sc["means"], True)                                 # small_scene(behind=True) puts Gaussians behind the camera
    vz = (np.c_[sc["means"], np.ones(len(sc["means"]))] @ cam["view"])[:, 2]
    keep = np.nonzero(vz > 0.2)[0]
    sc = {k: sc[k][keep] for k in ("means", "opac", "shs", "scales", "rots")}
    a, b = run(sc["means"], True), run(sc["means"], False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_scale_modifier_and_debug(oracle):
    from gpu_utils import forward_state
    sc, cam = small_scene(P=300, W=48, H=48, seed=9)
    bg = np.zeros(3, np.float32)
    fw = oracle.forward_full(sc, cam, bg, D=3, mod=1.7)
    st = forward_state(sc, cam, bg, D=3, mod=1.7, debug=True)
    _check_geometry(oracle, st, fw, sc, False)
    assert np.abs(st["color"] - fw["color"]).max() <= FWD_TOL


@pytest.mark.parametrize("P", [1, 5, 1000, 4097, 20000])
def test_knn(oracle, P):
    from gpu_utils import T
    from gaussianmesh_amd import distCUDA2
    rng = np.random.default_rng(P)
    pts = rng.normal(size=(P, 3)).astype(np.float32) * np.array([3, 1, 0.2], np.float32)
    if P >= 1000:
        pts[10] = pts[11]                      # duplicate -> distance 0 counted (simple_knn.cu:158,177 only skip i==idx)
    out = distCUDA2(T(pts)).cpu().numpy()
    ref = oracle.knn_mean_dist2(pts)
    if P >= 4:
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    else:                                      # fewer than 3 neighbours: FLT_MAX sentinels overflow identically
        assert np.array_equal(np.isfinite(out), np.isfinite(ref))


def test_deform_and_sh_colors(oracle):
    from gpu_utils import T
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.deform import deform_tensors, sh_colors
    verts, faces = scenes.torus_mesh(40, 30)
    N = 5000
    cl = scenes.bind_cloud_to_mesh(N, verts, faces, seed=4)
    V1, Rv, Sv = scenes.twist_bend_frame(verts, t=9)
    cov = scenes.cov3d_from_scale_rot(cl["scales"], cl["rots"]).astype(np.float32)
    dV = (V1 - verts).astype(np.float32)
    p_ref, c_ref, r_ref = oracle.deform(cl["tri"], cl["weights"], dV, Rv, Sv, cov, cl["means"])
    p, c, r, c6 = deform_tensors(T(cl["tri"], dtype=torch.int32), T(cl["weights"]), T(dV), T(Rv), T(Sv), T(cov), T(cl["means"]))
    p, c, r, c6 = (x.cpu().numpy() for x in (p, c, r, c6))
    assert np.abs(p - p_ref).max() <= 1e-5 * np.abs(p_ref).max()
    assert np.abs(c - c_ref).max() <= 1e-5 * np.abs(c_ref).max()
    assert np.abs(r - r_ref).max() <= 1e-5
    assert np.array_equal(c6, c[:, [0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]])
    campos = np.array([4.0, 1.0, -3.0], np.float32)
    rgb_ref = oracle.sh_colors_rotated(p_ref, campos, r_ref, cl["shs"], deg=3)
    rgb = sh_colors(T(p_ref), T(campos), T(cl["shs"]), rot=T(r_ref), deg=3).cpu().numpy()
    assert np.abs(rgb - rgb_ref).max() <= 1e-5
    # fused kernel == the two separate kernels, bit for bit (same arithmetic, different data movement)
    from gaussianmesh_amd.deform import deform_shade
    args = (T(cl["tri"], dtype=torch.int32), T(cl["weights"]), T(dV), T(Rv), T(Sv), T(cov), T(cl["means"]), T(cl["shs"]), T(campos))
    fp, fc6, frgb, fcov, frot = (x.cpu().numpy() for x in deform_shade(*args, deg=3, want_cov_rot=True))
    rgb_sep = sh_colors(T(p), T(campos), T(cl["shs"]), rot=T(r), deg=3).cpu().numpy()
    assert np.array_equal(fp, p) and np.array_equal(fc6, c6) and np.array_equal(fcov, c) and np.array_equal(frot, r)
    assert np.array_equal(frgb, rgb_sep)
    fp2, fc62, frgb2 = (x.cpu().numpy() for x in deform_shade(*args, deg=3))
    assert np.array_equal(fp2, p) and np.array_equal(fc62, c6) and np.array_equal(frgb2, rgb_sep)
    # packed per-vertex table (what the render loop uses on the broadcast [Vm,21] frame state): same bits again
    from gaussianmesh_amd.deform import deform_shade_packed, pack_mesh_state
    state = np.concatenate([V1.astype(np.float32), Rv.reshape(-1, 9), Sv.reshape(-1, 9)], axis=1).astype(np.float32)
    packed = pack_mesh_state(T(state), T(verts.astype(np.float32)))
    pk = packed.cpu().numpy()
    assert np.array_equal(pk[:, 0:3], (state[:, 0:3] - verts.astype(np.float32))) and np.array_equal(pk[:, 4:13], state[:, 3:12])
    assert np.array_equal(pk[:, 13:22], state[:, 12:21]) and (pk[:, [3, 22, 23]] == 0).all()
    pargs = (T(cl["tri"], dtype=torch.int32), T(cl["weights"]), packed, T(cov), T(cl["means"]), T(cl["shs"]), T(campos))
    qp, qc6, qrgb, qcov, qrot = (x.cpu().numpy() for x in deform_shade_packed(*pargs, deg=3, want_cov_rot=True))
    dV32 = T(pk[:, 0:3].copy())
    rp, rc6, rrgb = (x.cpu().numpy() for x in deform_shade(args[0], args[1], dV32, *args[3:], deg=3))
    assert np.array_equal(qp, rp) and np.array_equal(qc6, rc6) and np.array_equal(qrgb, rrgb) and np.array_equal(qrot, r)


def test_cov_to_scale_rot(oracle):
    """a22: eigh + quaternion route.  Parity on the reconstructed covariance and on the eigenvalues (numpy eigh)."""
    from gpu_utils import T, forward_state
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.deform import cov_to_scale_rot
    rng = np.random.default_rng(3)
    cl = scenes.make_cloud(4000, seed=8, scale_lo=0.01, scale_hi=0.5)
    cov = scenes.cov3d_from_scale_rot(cl["scales"], cl["rots"])
    cov[:10] = np.eye(3) * 0.04                                  # degenerate (all eigenvalues equal)
    sc_rep = cl["scales"][10:20].copy(); sc_rep[:, 2] = sc_rep[:, 1]                      # repeated eigenvalue
    cov[10:20] = scenes.cov3d_from_scale_rot(sc_rep, cl["rots"][10:20])
    cov = cov.astype(np.float32).astype(np.float64)                 # what the kernel actually sees
    s, q = (x.cpu().numpy().astype(np.float64) for x in cov_to_scale_rot(T(cov.astype(np.float32))))
    w = np.linalg.eigvalsh(cov)
    assert np.abs(s ** 2 - w).max() <= 1e-5 * np.abs(w).max()
    assert (np.diff(s, axis=1) >= -1e-7).all()                   # ascending, like eigh
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-6)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                  2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    rec = R @ (s[:, :, None] ** 2 * R.transpose(0, 2, 1))
    assert np.abs(rec - cov).max() <= 2e-6 * np.abs(cov).max() + 1e-9
    assert np.allclose(np.linalg.det(R), 1.0, atol=1e-6)
    # rendering with (scales, rotations) from the kernel == rendering with the covariance itself
    cam = scenes.orbit_camera(1, 6, 200, 120, radius=7.0)
    bg = np.zeros(3, np.float32)
    sc = dict(cl); sc["cov3D_precomp"] = scenes.strip_symmetric(cov); sc["colors_precomp"] = rng.uniform(0, 1, (4000, 3)).astype(np.float32)
    a = forward_state(sc, cam, bg, D=3, use_precomp_cov=True, use_precomp_color=True)
    sc2 = dict(sc); sc2["scales"] = s.astype(np.float32); sc2["rots"] = q.astype(np.float32)
    b = forward_state(sc2, cam, bg, D=3, use_precomp_cov=False, use_precomp_color=True)
    assert (a["radii"] != b["radii"]).mean() <= 2e-3            # ceil(3 sigma) may flip by one on a few Gaussians
    assert np.abs(a["color"] - b["color"]).max() <= 2e-3


def test_render_glue_contract(oracle):
    """8f-1: render() dict contract, mesh-bound get_xyz, gradients reaching the barycentric parameters."""
    from types import SimpleNamespace
    from gpu_utils import T
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.renderer import Camera, MeshBoundGaussians, render, render_deformed, strip_symmetric
    from gaussianmesh_amd.deform import SingleObjectDeform
    verts, faces = scenes.torus_mesh(24, 16)
    N = 3000
    rng = np.random.default_rng(0)
    cl = scenes.bind_cloud_to_mesh(N, verts, faces, seed=2)
    tri = faces[cl["fid"]]
    v1, v2, v3 = (verts[tri[:, k]].astype(np.float32) for k in range(3))
    n = np.cross(v2 - v1, v3 - v1); n /= np.linalg.norm(n, axis=1, keepdims=True)
    r = ((np.linalg.norm(v2 - v1, axis=1) + np.linalg.norm(v3 - v2, axis=1) + np.linalg.norm(v1 - v3, axis=1)) / 3)[:, None]
    pc = MeshBoundGaussians(T(rng.normal(size=(N, 3))), T(rng.normal(0, 0.3, size=(N, 1))), T(cl["shs"][:, :1]), T(cl["shs"][:, 1:]),
                            T(np.log(cl["scales"] * 8)), T(cl["rots"]), T(rng.normal(size=(N, 1))), T(v1), T(v2), T(v3), T(n), T(r)).cuda()
    cam_d = scenes.orbit_camera(1, 6, 160, 96, radius=7.0)
    cam = Camera(cam_d, "cuda")
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    bg = torch.zeros(3, device="cuda")
    out = render(cam, pc, pipe, bg)
    assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii", "vertex1", "vertex2", "vertex3", "scale"}
    assert out["render"].shape == (3, 96, 160) and out["radii"].shape == (N,)
    assert torch.equal(out["visibility_filter"], out["radii"] > 0)
    # same image as the oracle on the model's activated parameters
    sc = dict(means=pc.get_xyz.detach().cpu().numpy(), opac=pc.get_opacity.detach().cpu().numpy(), shs=pc.get_features.detach().cpu().numpy(),
              scales=pc.get_scaling.detach().cpu().numpy(), rots=pc.get_rotation.detach().cpu().numpy())
    fw = oracle.forward_full(sc, cam_d, np.zeros(3, np.float32), D=3)
    assert np.array_equal(out["radii"].cpu().numpy(), fw["geo"]["radii"])
    assert np.abs(out["render"].detach().cpu().numpy() - fw["color"]).max() <= FWD_TOL
    out["render"].sum().backward()
    for p in (pc._bc, pc._distance, pc._scaling, pc._rotation, pc._opacity, pc._features, pc.screenspace_points):
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().max() > 0
    # python-side covariance / colour switches give the same picture
    pipe2 = SimpleNamespace(convert_SHs_python=True, compute_cov3D_python=True, debug=False)
    with torch.no_grad():
        out2 = render(cam, pc, pipe2, bg)
    assert (out2["render"] - out["render"]).abs().max() <= 2e-4
    # edit-tool route: identity deformation of an object == rendering its rest state with colours from SH
    cov = scenes.cov3d_from_scale_rot(sc["scales"], sc["rots"]).astype(np.float32)
    Vm = verts.shape[0]
    I = np.tile(np.eye(3, dtype=np.float32), (Vm, 1, 1))
    obj = SingleObjectDeform(T(sc["means"]), T(cov), T(sc["opac"]), T(sc["shs"]), T(tri, dtype=torch.int32), T(cl["weights"]), T(verts))
    obj.deform(T(verts), T(I), T(I))
    img = render_deformed(cam, [obj], bg_color=bg)
    assert (img - out["render"].detach()).abs().max() <= 2e-4
    # ... and the fused frame (deformation + colour + preprocess in one kernel) gives the same picture as the two-step route
    img2 = obj.deform_and_render(T(verts), T(I), T(I), cam, bg_color=bg)
    assert (img2 - img).abs().max() <= 1e-6


def test_async_begin_finish_pipeline_matches_sync(oracle):
    """gm_forward_0_async + deferred gm_forward_1 on alternating streams == the synchronous operator, bit for bit."""
    from gpu_utils import T
    from gaussianmesh_amd import rasterizer as Rz, scenes
    sc = scenes.make_cloud(20000, seed=4, scale_lo=0.01, scale_hi=0.15)
    S = scenes.cov3d_from_scale_rot(sc["scales"], sc["rots"])
    cov6 = T(scenes.strip_symmetric(S)); rgb = T(np.random.default_rng(0).uniform(0, 1, (20000, 3)))
    means, opac, bg = T(sc["means"]), T(sc["opac"]), T(np.ones(3))
    cams = [scenes.orbit_camera(k, 8, 320, 200, radius=7.5) for k in range(6)]
    ct = [{k: T(c[k]) for k in ("view", "proj", "campos")} for c in cams]
    call = lambda k, **kw: (bg, means, rgb, opac, None, None, 1.0, cov6, ct[k]["view"], ct[k]["proj"], cams[k]["tanx"], cams[k]["tany"],
                            200, 320, None, 3, ct[k]["campos"])
    ref = []
    for k in range(6):
        nr, color, radii, *_ = Rz.rasterize_forward(*call(k), False, False)
        ref.append((nr, color.clone(), radii.clone()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    ws = [Rz.RasterWorkspace() for _ in range(3)]
    out, pend = {}, {}
    for k in range(6):
        with torch.cuda.stream(streams[k % 3]):
            pend[k] = Rz.rasterize_forward_begin(*call(k), workspace=ws[k % 3])
        if k - 1 in pend:
            out[k - 1] = pend.pop(k - 1).finish()
    out[5] = pend.pop(5).finish()
    torch.cuda.synchronize()
    for k in range(6):
        assert out[k][0] == ref[k][0]
        assert torch.equal(out[k][1], ref[k][1]) and torch.equal(out[k][2], ref[k][2])


@pytest.mark.parametrize("N", [5000, 64 * 40])
def test_fused_deform_forward_equals_unfused_chain(oracle, N):
    """gm_forward_0_deformed_async + gm_forward_1_geom (edit-loop fast path) == gm_deform_shade_packed followed by the
    ordinary forward on its outputs, bit for bit (image, radii, instance count, deformed cloud), and the image matches
    the oracle's deform -> colour -> forward chain within the forward tolerance."""
    from gpu_utils import T
    from gaussianmesh_amd import rasterizer as Rz, scenes
    from gaussianmesh_amd.deform import deform_shade_packed, pack_mesh_state
    verts, faces = scenes.torus_mesh(40, 30)
    cl = scenes.bind_cloud_to_mesh(N, verts, faces, seed=11)
    V1, Rv, Sv = scenes.twist_bend_frame(verts, t=5)
    cov = scenes.cov3d_from_scale_rot(cl["scales"], cl["rots"]).astype(np.float32)
    state = np.concatenate([V1.astype(np.float32), Rv.reshape(-1, 9), Sv.reshape(-1, 9)], axis=1).astype(np.float32)
    W, H = 300, 170
    cam = scenes.orbit_camera(1, 7, W, H, radius=6.0)
    ct = {k: T(cam[k]) for k in ("view", "proj", "campos")}
    bg = T(np.array([0.1, 0.5, 0.9], np.float32))
    packed = pack_mesh_state(T(state), T(verts.astype(np.float32)))
    g = dict(tri=T(cl["tri"], dtype=torch.int32), w=T(cl["weights"]), cov=T(cov), pos=T(cl["means"]), shs=T(cl["shs"]), opac=T(cl["opac"]))
    pos, cov6, rgb = deform_shade_packed(g["tri"], g["w"], packed, g["cov"], g["pos"], g["shs"], ct["campos"], deg=3)
    nr0, color0, radii0, *_ = Rz.rasterize_forward(bg, pos, rgb, g["opac"], None, None, 1.0, cov6, ct["view"], ct["proj"], cam["tanx"],
                                                   cam["tany"], H, W, None, 3, ct["campos"], False, False)
    for want in (False, True):
        h = Rz.forward_deformed_begin(bg, g["tri"], g["w"], packed, g["cov"], g["pos"], g["shs"], g["opac"], ct["view"], ct["proj"],
                                      cam["tanx"], cam["tany"], H, W, 3, ct["campos"], want_deformed=want)
        nr1, color1, radii1, *_ = h.finish()
        torch.cuda.synchronize()
        assert nr1 == nr0 and torch.equal(radii1, radii0) and torch.equal(color1, color0)
        if want:
            assert torch.equal(h.deformed[0], pos) and torch.equal(h.deformed[1], cov6) and torch.equal(h.deformed[2], rgb)
    # the rest covariances handed over as [N,6] (GM_STREAM_COV6: these matrices are M M^T products, symmetric bit for bit): the same frame,
    # deformed cloud included; a matrix that is not bit-symmetric is not packed
    from gaussianmesh_amd.deform import pack_cov6
    c6 = pack_cov6(g["cov"])
    assert c6 is not None and c6.shape == (N, 6)
    h = Rz.forward_deformed_begin(bg, g["tri"], g["w"], packed, c6, g["pos"], g["shs"], g["opac"], ct["view"], ct["proj"],
                                  cam["tanx"], cam["tany"], H, W, 3, ct["campos"], want_deformed=True)
    nr1, color1, radii1, *_ = h.finish()
    torch.cuda.synchronize()
    assert nr1 == nr0 and torch.equal(radii1, radii0) and torch.equal(color1, color0)
    assert torch.equal(h.deformed[0], pos) and torch.equal(h.deformed[1], cov6) and torch.equal(h.deformed[2], rgb)
    skew = g["cov"].clone().reshape(N, 3, 3)
    skew[7, 0, 1] = torch.nextafter(skew[7, 0, 1], skew[7, 0, 1] + 1)
    assert pack_cov6(skew) is None
    # a frame begun without the count copy (what a sync-free loop does): completed exactly all the same (the count is fetched
    # on demand), and after an overflow the status words supply it
    fb = lambda **kw: Rz.forward_deformed_begin(bg, g["tri"], g["w"], packed, g["cov"], g["pos"], g["shs"], g["opac"], ct["view"], ct["proj"],
                                                cam["tanx"], cam["tany"], H, W, 3, ct["campos"], want_count=False, **kw)
    nr2, color2, *_ = fb().finish()
    assert nr2 == nr0 and torch.equal(color2, color0)
    ws = Rz.RasterWorkspace()
    ws.capacity = 64                                    # far too small
    h = fb(workspace=ws)
    nr3, color3, *_ = h.finish(sync_free=True)
    ok, count = h.check()                               # status words written by the blend kernel itself
    assert nr3 == -1 and not ok and count == nr0 and torch.equal(color3, bg.view(3, 1, 1).expand(3, H, W))
    nr4, color4, *_ = h.finish()
    assert nr4 == nr0 and torch.equal(color4, color0)
    h = fb(workspace=ws)                                # the workspace has grown: the same frame now fits
    _, color5, *_ = h.finish(sync_free=True)
    ok, count = h.check()
    assert ok and count == nr0 and torch.equal(color5, color0)
    # image-only frame (GM_FWD_IMAGE_ONLY): same image, and the backward state of the image buffer (final transmittance at the
    # head of the buffer, contributor counts behind it) is left as it was
    h = fb(workspace=ws)
    img = h.img
    img[:2 * 4 * H * W].fill_(0xA5)
    snapshot = img[:2 * 4 * H * W].clone()
    for sync_free in (True, False):
        h = fb(workspace=ws) if h.result is not None else h
        _, color6, *rest = h.finish(sync_free=sync_free, image_only=True)
        assert h.check()[0]
        torch.cuda.synchronize()
        assert h.img.data_ptr() == img.data_ptr() and torch.equal(color6, color0) and torch.equal(img[:2 * 4 * H * W], snapshot)
    _, color7, *rest = fb(workspace=ws).finish()
    torch.cuda.synchronize()
    assert torch.equal(color7, color0) and not torch.equal(img[:4 * H * W], snapshot[:4 * H * W])
    # work hint: the blend's dispatch order follows what tiles cost in earlier frames; images never change
    hint = Rz.new_work_hint(W, H, color0.device)
    for k in range(4):
        h = fb(workspace=ws)
        _, color8, *rest = h.finish(sync_free=bool(k & 1), image_only=bool(k & 2), work_hint=hint)
        assert h.check()[0]
        torch.cuda.synchronize()
        assert torch.equal(color8, color0)
        hv = hint.cpu().numpy().view(np.uint32)
        assert int(hv[0]) == k + 1                                         # frames counted
        entries = hv[1:][hv[1:] != 0]
        assert entries.size > 0 and set((entries >> 20).tolist()) <= set(range(1, k + 2)) and (entries & 0xFFFFF).max() <= nr0
    with pytest.raises(Exception):
        fb(workspace=Rz.RasterWorkspace()).finish(work_hint=torch.zeros(4, dtype=torch.int32, device=color0.device))   # too small
    # against the oracle chain
    dV = (V1 - verts).astype(np.float32)
    p_ref, c_ref, r_ref = oracle.deform(cl["tri"], cl["weights"], dV, Rv, Sv, cov, cl["means"])
    rgb_ref = oracle.sh_colors_rotated(p_ref, cam["campos"], r_ref, cl["shs"], deg=3)
    sc = dict(means=p_ref.astype(np.float32), opac=cl["opac"], colors_precomp=rgb_ref.astype(np.float32),
              cov3D_precomp=scenes.strip_symmetric(c_ref).astype(np.float32))
    fw = oracle.forward_full(sc, cam, bg.cpu().numpy(), D=3, use_precomp_cov=True, use_precomp_color=True)
    assert_forward_gate(fw, color0.cpu().numpy(), W, H, FWD_TOL, "fused frame")


def test_policy_change_between_forward_halves_is_refused_safely():
    """gm_forward_1_geom under another emission policy than its gm_forward_0_async: the counted size belongs to the old
    policy, so emission is refused on the device, every list stays empty and the image is the background (no overrun, no
    hang); the status words report the refusal."""
    from gpu_utils import T
    from gaussianmesh_amd import rasterizer as Rz, scenes, _lib
    sc = scenes.make_cloud(5000, seed=2, scale_lo=0.01, scale_hi=0.2)
    cam = scenes.orbit_camera(0, 4, 200, 120, radius=7.0)
    ct = {k: T(cam[k]) for k in ("view", "proj", "campos")}
    bg = T(np.array([0.25, 0.5, 0.75], np.float32))
    args = (bg, T(sc["means"]), None, T(sc["opac"]), T(sc["scales"]), T(sc["rots"]), 1.0, None, ct["view"], ct["proj"], cam["tanx"],
            cam["tany"], 120, 200, T(sc["shs"]), 3, ct["campos"])
    h = Rz.rasterize_forward_begin(*args, emission_policy=2)
    h.policy = 0                                    # the second half is issued under the reference policy
    nr, color, radii, geom, *_ = h.finish()
    st = torch.zeros(4, dtype=torch.int32).pin_memory()
    _lib.check(_lib.lib().gm_forward_status_async(geom.data_ptr(), 5000, st.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert nr > 0 and torch.equal(color, bg.view(3, 1, 1).expand(3, 120, 200))
    assert int(st[0]) == nr and int(st[2]) == 2 and int(st[3]) == 1
    nr2, color2, *_ = Rz.rasterize_forward(*args, False, False, emission_policy=2)      # and the library is usable again afterwards
    assert nr2 == nr and not torch.equal(color2, color)


def test_sync_free_forward_matches_exact_and_reports_overflow():
    """finish(sync_free=True): the instance count never reaches the host; the image equals the exact path's bit for bit
    when the workspace's binning capacity suffices, and an overflow renders the background, is reported by check() and is
    repaired by finish()."""
    from gpu_utils import T
    from gaussianmesh_amd import rasterizer as Rz, scenes
    sc = scenes.make_cloud(20000, seed=6, scale_lo=0.01, scale_hi=0.15)
    cams = [scenes.orbit_camera(k, 8, 320, 200, radius=7.5) for k in range(3)]
    bg = T(np.array([0.1, 0.2, 0.3], np.float32))
    def args(k):
        ct = {n: T(cams[k][n]) for n in ("view", "proj", "campos")}
        return (bg, T(sc["means"]), None, T(sc["opac"]), T(sc["scales"]), T(sc["rots"]), 1.0, None, ct["view"], ct["proj"], cams[k]["tanx"],
                cams[k]["tany"], 200, 320, T(sc["shs"]), 3, ct["campos"])
    ref = [Rz.rasterize_forward(*args(k), False, False) for k in range(3)]
    ws = Rz.RasterWorkspace()
    h = Rz.rasterize_forward_begin(*args(0), workspace=ws)
    out = h.finish(sync_free=True)                 # no capacity known yet: takes the exact path and sizes the buffer
    assert out[0] == ref[0][0] and torch.equal(out[1], ref[0][1])
    for k in (1, 2):
        h = Rz.rasterize_forward_begin(*args(k), workspace=ws)
        nr, color, radii, *_ = h.finish(sync_free=True)
        assert nr == -1
        ok, count = h.check()
        assert ok and count == ref[k][0] and torch.equal(color, ref[k][1]) and torch.equal(radii, ref[k][2])
    ws.capacity = 1000                             # far too small for the next frame
    h = Rz.rasterize_forward_begin(*args(0), workspace=ws)
    nr, color, *_ = h.finish(sync_free=True)
    ok, count = h.check()
    assert not ok and count == ref[0][0] and torch.equal(color, bg.view(3, 1, 1).expand(3, 200, 320))
    nr, color, *_ = h.finish()                     # exact path on the same frame
    assert nr == ref[0][0] and torch.equal(color, ref[0][1])
    with pytest.raises(Exception):                 # a workspace serves one frame at a time
        h1 = Rz.rasterize_forward_begin(*args(1), workspace=ws)
        Rz.rasterize_forward_begin(*args(2), workspace=ws)
    h1.finish()


def test_fused_frame_edge_cases():
    """Edit-loop fast path on an empty cloud, a one-Gaussian cloud and a cloud entirely behind the camera."""
    from gpu_utils import T
    from gaussianmesh_amd import rasterizer as Rz, scenes
    from gaussianmesh_amd.deform import pack_mesh_state
    verts, faces = scenes.torus_mesh(12, 8)
    Vm = verts.shape[0]
    state = np.concatenate([verts.astype(np.float32), np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (Vm, 2))], axis=1).astype(np.float32)
    packed = pack_mesh_state(T(state), T(verts.astype(np.float32)))
    cam = scenes.orbit_camera(0, 4, 96, 64, radius=6.0)
    ct = {k: T(cam[k]) for k in ("view", "proj", "campos")}
    bg = T(np.array([0.2, 0.4, 0.6], np.float32))
    for N in (0, 1, 70):
        cl = scenes.bind_cloud_to_mesh(max(N, 1), verts, faces, seed=1)
        sl = lambda a: a[:N]
        cov = scenes.cov3d_from_scale_rot(cl["scales"], cl["rots"]).astype(np.float32)
        pos = cl["means"].copy()
        if N == 70:
            pos += 100.0 * (cam["campos"] - 0) / np.linalg.norm(cam["campos"])       # far behind the camera
        h = Rz.forward_deformed_begin(bg, T(sl(cl["tri"]), dtype=torch.int32).reshape(-1, 3), T(sl(cl["weights"])).reshape(-1, 3), packed,
                                      T(sl(cov)).reshape(-1, 3, 3), T(sl(pos)).reshape(-1, 3), T(sl(cl["shs"])).reshape(-1, 16, 3),
                                      T(sl(cl["opac"])).reshape(-1, 1), ct["view"], ct["proj"], cam["tanx"], cam["tany"], 64, 96, 3, ct["campos"])
        nr, color, radii, *_ = h.finish()
        torch.cuda.synchronize()
        assert radii.shape == (N,) and color.shape == (3, 64, 96) and torch.isfinite(color).all()
        if N in (0, 70):
            assert nr == 0 and torch.equal(color, bg.view(3, 1, 1).expand(3, 64, 96))
            assert N == 0 or int((radii > 0).sum()) == 0


@pytest.mark.parametrize("seed", [0, 1, 4, 9, 18, 21, 28, 34])
def test_emission_policies_agree_on_random_scenes(seed):
    """Randomised scenes (needle splats, opacities around 1/255, cameras inside the cloud, odd image sizes, huge splats):
    every culling policy renders bit-identical images to the reference emission (tools/fuzz_policies.py runs 40 seeds)."""
    from gpu_utils import forward_state
    from gaussianmesh_amd import scenes
    rng = np.random.default_rng(seed)
    P = int(rng.integers(200, 30000))
    lo = float(10 ** rng.uniform(-3, -1)); hi = lo * float(10 ** rng.uniform(0.3, 2.2))
    sc = scenes.make_cloud(P, seed=seed, scale_lo=lo, scale_hi=hi)
    if seed % 3 == 0:
        sc["scales"][:, 0] *= 20.0
    if seed % 4 == 1:
        sc["opac"][:] = rng.uniform(0.003, 0.02, size=sc["opac"].shape).astype(np.float32)
    W = int(rng.integers(17, 700)); H = int(rng.integers(17, 500))
    cam = scenes.orbit_camera(int(rng.integers(0, 16)), 16, W, H, radius=float(rng.uniform(0.5, 9.0)))
    bg = rng.random(3).astype(np.float32)
    ref = forward_state(sc, cam, bg, D=3, tile_cull=0)
    for mode in (1, 2, 3):
        cu = forward_state(sc, cam, bg, D=3, tile_cull=mode)
        assert np.array_equal(cu["radii"], ref["radii"]) and np.array_equal(cu["final_T"], ref["final_T"]), mode
        assert np.array_equal(cu["color"], ref["color"]), mode


@pytest.mark.parametrize("case", ["equal_depths", "depth_pileup", "many_tiles", "tiny", "twenty_octaves", "object_and_background"])
def test_ordering_paths(oracle, case):
    """gm_bucket.hip's special paths against the oracle's (tile, depth, id) stable sort, lists bit-exact:
    equal_depths: every Gaussian at the same view depth (one bucket, no low bits: order by id, bucket larger than the LDS);
    depth_pileup: 12k Gaussians inside a few ulps of one depth plus a sparse spread (one overfull bucket WITH low bits: the
                  global-memory slow path of bucket_sort_kernel, several passes);
    many_tiles:   more than 2048 list tiles (two 8-bit tile passes + tile_ranges_kernel), reference and default policy;
    tiny:         3 Gaussians;
    twenty_octaves: depths from 0.3 to 3e5 around the optical axis - 160 sparsely filled coarse depth bins;
    object_and_background: a dense cluster in a tenth of a depth unit in front of a sparse background and behind a few
                  floaters (the proportional bucket table: most buckets go to the cluster)."""
    from gpu_utils import forward_state
    from gaussianmesh_amd import scenes
    rng = np.random.default_rng(5)
    W, H = 320, 200
    if case in ("equal_depths", "depth_pileup"):
        P = 14000
        sc = scenes.make_cloud(P, seed=3, scale_lo=0.004, scale_hi=0.03)
        # camera on the -z axis looking along +z: view z = world z + 6 exactly
        cam = scenes.look_at_camera((0.0, 0.0, -6.0), (0.0, 0.0, 0.0), W, H)
        assert cam["view"].reshape(-1)[2] == 0 and cam["view"].reshape(-1)[6] == 0
        sc["means"][:, 2] = 0.5
        if case == "depth_pileup":
            sc["means"][:12000, 2] = (0.5 + rng.integers(0, 24, 12000) * 4.76837158203125e-07).astype(np.float32)
            sc["means"][12000:, 2] = rng.uniform(-4.0, 60.0, P - 12000).astype(np.float32)
        modes = (0, 2)
    elif case == "many_tiles":
        W, H = 1600, 1000                                       # 100 x 63 = 6300 16-px tiles (policy 0 / 1), 1600 parents (policy 2)
        sc = scenes.make_cloud(3000, seed=8, scale_lo=0.01, scale_hi=0.3)
        cam = scenes.orbit_camera(2, 9, W, H, radius=7.0)
        modes = (0, 1, 2)
    elif case in ("twenty_octaves", "object_and_background"):
        P = 6000 if case == "twenty_octaves" else 20000
        sc = scenes.make_cloud(P, seed=4, scale_lo=0.004, scale_hi=0.03)
        cam = scenes.look_at_camera((0.0, 0.0, -6.0), (0.0, 0.0, 0.0), W, H)
        if case == "twenty_octaves":
            z = (0.3 * 2.0 ** (20.0 * np.arange(P) / P)).astype(np.float32)
            sc["means"] = (rng.uniform(-0.05, 0.05, (P, 3)) * z[:, None]).astype(np.float32)     # inside the frustum at every depth
            sc["means"][:, 2] = rng.permutation(z) - 6.0
            sc["scales"] = (sc["scales"] * z[:, None] / 6.0).astype(np.float32)
        else:
            sc["means"][:17000, 2] = rng.uniform(0.0, 0.1, 17000).astype(np.float32)              # the object: view depth 6.0 .. 6.1
            sc["means"][17000:19900, 2] = rng.uniform(20.0, 900.0, 2900).astype(np.float32)       # background
            sc["means"][19900:, 2] = rng.uniform(-5.6, -5.0, 100).astype(np.float32)              # floaters right in front of the camera
            sc["means"][17000:, :2] *= 0.2
        modes = (0, 2)
    else:
        sc = scenes.make_cloud(3, seed=1, scale_lo=0.05, scale_hi=0.3)
        cam = scenes.orbit_camera(0, 4, W, H, radius=5.0)
        modes = (0, 2)
    bg = np.array([0.2, 0.3, 0.4], np.float32)
    fw = oracle.forward_full(sc, cam, bg, D=3)
    if case in ("twenty_octaves", "object_and_background"):
        assert (fw["geo"]["radii"] > 0).sum() > 0.5 * sc["means"].shape[0]
    ex = None
    for mode in modes:
        st = forward_state(sc, cam, bg, D=3, tile_cull=mode)
        assert np.array_equal(st["radii"], fw["geo"]["radii"])
        vis = np.nonzero(fw["geo"]["radii"] > 0)[0]
        dbits = fw["geo"]["depths"].view(np.uint32)[vis].astype(np.int64)
        want = vis[np.lexsort((vis, dbits))]                   # (depth bits, id) order of the visible Gaussians
        assert np.array_equal(st["order"], want.astype(np.uint32)), case
        if mode == 0:
            assert st["R"] == fw["bins"]["R"] and np.array_equal(st["point_list"], fw["bins"]["point_list"])
            assert np.array_equal(st["ranges"], fw["bins"]["ranges"])
            ex = st
        else:
            assert np.array_equal(st["color"], ex["color"]) and np.array_equal(st["final_T"], ex["final_T"])
        assert_forward_gate(fw, st["color"], W, H, FWD_TOL, "%s policy %d" % (case, mode))


def test_python_sh_route_is_differentiable_and_debug_snapshots(oracle, tmp_path, monkeypatch):
    """pipe.convert_SHs_python (gaussian_renderer/__init__.py:84-92): same image and the same gradients - to the features and,
    through the view direction, to the positions - as the in-op SH path; debug=True keeps the inputs of a failing call."""
    from types import SimpleNamespace
    from gpu_utils import T
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.renderer import Camera, render
    N = 3000
    cl = scenes.make_cloud(N, seed=6, scale_lo=0.03, scale_hi=0.3)
    cam = Camera(scenes.orbit_camera(1, 6, 160, 96, radius=7.0), "cuda")
    wgt = torch.randn((3, 96, 160), device="cuda")

    def run(py):
        pc = SimpleNamespace(get_xyz=T(cl["means"], True), get_opacity=T(cl["opac"]).reshape(-1, 1), get_scaling=T(cl["scales"]),
                             get_rotation=torch.nn.functional.normalize(T(cl["rots"])), get_features=T(cl["shs"], True), active_sh_degree=3,
                             max_sh_degree=3, screenspace_points=torch.zeros((N, 3), device="cuda", requires_grad=True))
        pipe = SimpleNamespace(convert_SHs_python=py, compute_cov3D_python=False, debug=False)
        out = render(cam, pc, pipe, torch.zeros(3, device="cuda"))
        (out["render"] * wgt).sum().backward()
        return out["render"].detach(), pc.get_xyz.grad, pc.get_features.grad
    img_a, gx_a, gf_a = run(False)
    img_b, gx_b, gf_b = run(True)
    assert (img_a - img_b).abs().max() <= 2e-5
    assert gf_b.abs().max() > 0 and (gf_a - gf_b).abs().max() <= 1e-3 * gf_a.abs().max()
    assert (gx_a - gx_b).abs().max() <= 1e-3 * gx_a.abs().max()
    # debug snapshot: an SH degree the library rejects, in debug mode -> snapshot_fw.dump next to the process
    from gaussianmesh_amd import GaussianRasterizationSettings, GaussianRasterizer
    monkeypatch.chdir(tmp_path)
    c = scenes.orbit_camera(1, 6, 64, 48, radius=7.0)
    rs = GaussianRasterizationSettings(48, 64, c["tanx"], c["tany"], torch.zeros(3, device="cuda"), 1.0, T(c["view"]), T(c["proj"]), 7,
                                       T(c["campos"]), False, True)
    with pytest.raises(Exception):
        GaussianRasterizer(rs)(T(cl["means"]), torch.zeros((N, 3), device="cuda"), T(cl["opac"]).reshape(-1, 1), shs=T(cl["shs"]),
                               scales=T(cl["scales"]), rotations=T(cl["rots"]))
    snap = torch.load(str(tmp_path / "snapshot_fw.dump"))
    assert snap["sh_degree"] == 7 and snap["means3D"].shape == (N, 3)


@pytest.mark.parametrize("W,H,policy", [(1600, 1000, 0), (1600, 1000, 2), (333, 211, 3)])
def test_work_hint_never_changes_an_image(W, H, policy):
    """gm_forward_1_geom's work_hint only reorders the blend's workgroups: images, final transmittance and contributor counts
    are those of a frame without it, on the one-pass tile sort (order from the scatter launch) and on the two-pass one
    (6300 list tiles: tile_order_kernel), frame after frame, with the camera moving under the same buffer."""
    from gpu_utils import T
    from gaussianmesh_amd import rasterizer as Rz, scenes
    sc = scenes.make_cloud(4000, seed=8, scale_lo=0.01, scale_hi=0.3)
    bg = T(np.array([0.2, 0.3, 0.4], np.float32))
    hint = Rz.new_work_hint(W, H, bg.device)
    lib = Rz._lib.lib()
    for k in range(5):
        cam = scenes.orbit_camera(k, 9, W, H, radius=7.0)
        args = (bg, T(sc["means"]), None, T(sc["opac"]), T(sc["scales"]), T(sc["rots"]), 1.0, None, T(cam["view"]), T(cam["proj"]), cam["tanx"],
                cam["tany"], H, W, T(sc["shs"]), 3, T(cam["campos"]), False, False)
        nr0, c0, r0, _, _, img0 = Rz.rasterize_forward_begin(*args, emission_policy=policy).finish()
        nr1, c1, r1, _, _, img1 = Rz.rasterize_forward_begin(*args, emission_policy=policy).finish(work_hint=hint)
        torch.cuda.synchronize()
        assert nr0 == nr1 and torch.equal(c0, c1) and torch.equal(r0, r1)
        for plane in (b"final_T", b"n_contrib"):
            o0 = lib.gm_image_field(img0.data_ptr(), W, H, plane) - img0.data_ptr()
            o1 = lib.gm_image_field(img1.data_ptr(), W, H, plane) - img1.data_ptr()
            assert torch.equal(img0[o0:o0 + 4 * W * H], img1[o1:o1 + 4 * W * H]), plane
        hv = hint.cpu().numpy().view(np.uint32)
        assert int(hv[0]) == k + 1 and (hv[1:] != 0).sum() > 0
        assert hv.size * 4 == lib.gm_work_hint_bytes(W, H)


def test_dispatch_order_is_a_permutation_while_the_work_hint_changes_under_it():
    """The work hint is shared by the frames in flight of a view stream: other frames' blend kernels update it while a frame's
    dispatch order is sorted from it.  Here a second stream rewrites the hint's entries with random costs for as long as the
    frames run; every frame's dispatch order must be a permutation of the list tiles and its image that of a frame without hint.
    (The sort used to read a tile's key in both of its passes; when the two reads disagreed, tiles were dispatched twice and
    others never - found by the two-rank bench test, a frame in ~10.)"""
    from gpu_utils import T
    from gaussianmesh_amd import rasterizer as Rz, scenes
    lib = Rz._lib.lib()
    for (W, H, policy) in ((640, 400, 2), (1600, 1000, 0)):         # one-pass tile sort (order from the scatter launch) / tile_order_kernel
        sc = scenes.make_cloud(6000, seed=8, scale_lo=0.01, scale_hi=0.2)
        bg = T(np.array([0.2, 0.3, 0.4], np.float32))
        cam = scenes.orbit_camera(1, 9, W, H, radius=7.0)
        args = (bg, T(sc["means"]), None, T(sc["opac"]), T(sc["scales"]), T(sc["rots"]), 1.0, None, T(cam["view"]), T(cam["proj"]), cam["tanx"],
                cam["tany"], H, W, T(sc["shs"]), 3, T(cam["campos"]), False, False)
        _, ref, *_ = Rz.rasterize_forward_begin(*args, emission_policy=policy).finish()
        hint = Rz.new_work_hint(W, H, bg.device)
        tiles = hint.numel() - 1 if policy == 0 else ((W + 31) // 32) * ((H + 31) // 32)
        side = torch.cuda.Stream()
        noise = [torch.randint(0, 1 << 20, (hint.numel() - 1,), dtype=torch.int32, device=bg.device) for _ in range(8)]
        bad_perm = bad_img = 0
        for k in range(8):
            noise[k] |= 1 << 20                                     # (a plausible frame tag, so that no entry is discarded as stale)
        for it in range(60):
            nr, color, _, _, _, img = Rz.rasterize_forward_begin(*args, emission_policy=policy).finish(work_hint=hint)
            with torch.cuda.stream(side):                           # rewrites of the costs racing with the frame's second half, which the
                for k in range(24):                                 # host has just enqueued (emission, tile pass + dispatch order, blend)
                    hint[1:].copy_(noise[(it + k) % 8])
            torch.cuda.synchronize()
            off = lib.gm_image_field(img.data_ptr(), W, H, b"tile_order") - img.data_ptr()
            order = img[off:off + 4 * tiles].view(torch.int32).cpu().numpy()
            bad_perm += int(not np.array_equal(np.sort(order), np.arange(tiles)))
            bad_img += int(not torch.equal(color, ref))
        assert bad_perm == 0 and bad_img == 0, (W, H, policy, bad_perm, bad_img)


def test_pipelined_deformed_loop_renders_every_frame_exactly():
    """bench.py's frame loop in small: mesh-driven frames pipelined over four streams, sync-free, image-only, one work hint shared by
    all of them, the first halves of two frames issued ahead of the oldest frame's completion - and EVERY frame compared, bit for
    bit, with the image the synchronous operator renders for its (mesh frame, camera)."""
    pipelined_deformed_loop(20000, 320, 200, 8, 640)


def pipelined_deformed_loop(P, W, H, F, frames, nstreams=4, ahead=2, lag=3, plan=False, cam_stride=3):
    """(also run at the bench's own size by tools/verify_loop_images.py)
    plan: the loop's frames share a DepthPlan (direct depth placement); a frame it refuses is rendered again by finish(), as a
    caller would, and compared like every other one.  Returns the plan (its .refused says how many)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from gpu_utils import T
    from gaussianmesh_amd import rasterizer as Rz, scenes
    from gaussianmesh_amd.deform import mesh_rs_packed, vertex_face_adjacency
    host = bench.build_scene(P, W, H, F)
    g = {k: T(host[k]) for k in ("weights", "pos", "cov", "opac", "shs", "verts")}
    g["tri"] = T(host["tri"], dtype=torch.int32)
    faces = T(host["faces"], dtype=torch.int32)
    off, adj = vertex_face_adjacency(host["faces"], host["verts"].shape[0])
    adjacency = (torch.tensor(off, device="cuda"), torch.tensor(adj, device="cuda"))
    v1 = [T(np.ascontiguousarray(host["mesh"][t][:, 0:3])) for t in range(F)]
    bg = torch.zeros(3, device="cuda")
    cams = []
    for k in range(F):
        cam = scenes.orbit_camera(k, F, W, H)
        cams.append((T(cam["view"]), T(cam["proj"]), cam["tanx"], cam["tany"], T(cam["campos"])))

    depth_plan = Rz.new_depth_plan(bg.device) if plan else None

    def begin(i, ws=None, dplan=None):
        t, c = i % F, cams[(cam_stride * i) % F]           # mesh frame and camera move at different rates: 8 x 8 combinations
        packed = mesh_rs_packed(g["verts"], v1[t], faces, adjacency)
        return Rz.forward_deformed_begin(bg, g["tri"], g["weights"], packed, g["cov"], g["pos"], g["shs"], g["opac"], c[0], c[1], c[2], c[3],
                                         H, W, 3, c[4], False, workspace=ws, want_count=ws is None, depth_plan=dplan)
    ref = {}
    for i in range(min(frames, F * F)):                     # the (mesh frame, camera) pairs the loop visits
        key = (i % F, (cam_stride * i) % F)
        if key not in ref:
            ref[key] = begin(i).finish()[1].clone()
    hint = Rz.new_work_hint(W, H, bg.device)
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    nws = nstreams + ahead + lag
    ws = [Rz.RasterWorkspace() for _ in range(nws)]
    for k, w in enumerate(ws):                              # size the binning buffers: every visited pair once through the exact path
        for i in range(min(frames, F * F, 4 * F)):
            h = begin(i, w)
            h.finish(work_hint=hint)
        w.capacity = int(w.capacity * 1.2)
    torch.cuda.synchronize()
    pending, done, bad = {}, [], []

    def complete(j):
        h = pending.pop(j)
        done.append((j, h, h.finish(sync_free=True, image_only=True, work_hint=hint)[1]))

    def verify():
        j, h, img = done.pop(0)
        ok, _ = h.check()
        if not ok and plan and h.refusal == 2:              # the direct placement refused the frame: again, on the partition path
            img = h.finish(image_only=True, work_hint=hint)[1]
            torch.cuda.synchronize()
        else:
            assert ok, "frame %d outgrew a buffer sized for the heaviest frame" % j
        if not torch.equal(img, ref[(j % F, (cam_stride * j) % F)]):
            bad.append(j)
    for i in range(frames):
        with torch.cuda.stream(streams[i % nstreams]):
            pending[i] = begin(i, ws[i % nws], depth_plan)
        if i - ahead in pending:
            complete(i - ahead)
        while len(done) > lag:
            verify()
    for j in sorted(pending):
        complete(j)
    while done:
        verify()
    assert not bad, "%d of %d pipelined frames differ from the synchronous render: %s" % (len(bad), frames, bad[:8])
    return depth_plan


def test_splat_centred_on_a_pixel_is_not_dropped(oracle):
    """Round 3's forward evaluates the exponent as a polynomial on the matrix core (|error| ~1e-5) and therefore clamps it at 0
    where the reference skips `power > 0` (RAST/forward.cu:338-339: a guard against its own rounding at power = -0 +- 1e-7).
    Splats whose centres fall EXACTLY on pixel centres - power == 0 there, the polynomial lands on either side of it - must still
    be rendered at their brightest pixel: image and gradients against the oracle, which keeps power == 0."""
    from gpu_utils import forward_state
    from gaussianmesh_amd import scenes
    W, H, D = 96, 64, 3
    cam = scenes.orbit_camera(0, 8, W, H, radius=6.0)
    sc = scenes.make_cloud(400, seed=21, scale_lo=0.02, scale_hi=0.15, D=D)
    # move every Gaussian onto the ray through a pixel centre: project, round to the pixel grid, unproject at the same depth
    view = cam["view"].astype(np.float64); proj = cam["proj"].astype(np.float64)
    m = sc["means"].astype(np.float64)
    hom = np.concatenate([m, np.ones((len(m), 1))], 1) @ proj
    ndc = hom[:, :2] / hom[:, 3:4]
    pix = ((ndc + 1.0) * np.array([W, H]) - 1.0) * 0.5
    tgt = np.clip(np.round(pix), 2, [W - 3, H - 3])
    ndc_t = (2.0 * tgt + 1.0) / np.array([W, H]) - 1.0
    vz = (np.concatenate([m, np.ones((len(m), 1))], 1) @ view)[:, 2]
    vx, vy = ndc_t[:, 0] * cam["tanx"] * vz, ndc_t[:, 1] * cam["tany"] * vz
    back = np.concatenate([np.stack([vx, vy, vz], 1), np.ones((len(m), 1))], 1) @ np.linalg.inv(view)
    sc["means"] = back[:, :3].astype(np.float32)
    sc["opac"][:] = np.linspace(0.3, 0.95, len(m), dtype=np.float32).reshape(sc["opac"].shape)
    bg = np.zeros(3, np.float32)
    fw = oracle.forward_full(sc, cam, bg, D=D)
    on_centre = np.abs(fw["geo"]["xy"] - np.round(fw["geo"]["xy"])).max(axis=1) < 2e-4
    assert on_centre.sum() > 300                                          # (float32 projection: most land within 2e-4 px of a centre)
    st = forward_state(sc, cam, bg, D=D)
    assert np.array_equal(st["radii"], fw["geo"]["radii"])
    assert_forward_gate(fw, st["color"], W, H, FWD_TOL, "centred splats")
    dpix = np.random.default_rng(5).normal(size=(3, H, W)).astype(np.float32)
    bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=D)
    _, _, g = _grads_gpu(sc, cam, bg, dpix, D, False, False)
    _grad_gate(g["opac"].reshape(-1), bw["dopacity"], "opacity")
    _grad_gate(g["means"], bw["dmean3D"], "means")


@pytest.mark.parametrize("mode", ["sh_scale_rot", "precomp"])
def test_hip_path_vs_independent_dense_autograd(mode):
    """The HIP operator straight against oracle/torch_dense.py - the float64, textbook-form restatement whose gradients come from
    AUTOGRAD, not from a hand-derived chain rule - with no C oracle in between (the C oracle and the HIP backward restate the same
    hand derivation of RAST/backward.cu; this is the check that does not share it).  2400 Gaussians on 48 tiles: lists of several
    hundred entries, saturating pixels, both input modes."""
    from oracle import torch_dense as td
    from gpu_utils import T, settings
    from gaussianmesh_amd import GaussianRasterizer
    D = 3
    sc, cam = small_scene(P=2400, W=128, H=96, seed=9, D=D, scale_lo=0.05, scale_hi=0.35)
    bg = np.array([0.2, 0.5, 0.9], np.float32)
    pre = mode == "precomp"
    t64 = lambda a, rg=False: torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=rg)
    means, opac = t64(sc["means"], True), t64(sc["opac"], True)
    m2d = torch.zeros(means.shape[0], 3, dtype=torch.float64, requires_grad=True)
    if pre:
        ref_in = dict(colors_precomp=t64(sc["colors_precomp"], True), cov3D_precomp=t64(sc["cov3D_precomp"], True))
    else:
        ref_in = dict(shs=t64(sc["shs"], True), scales=t64(sc["scales"], True), rots=t64(sc["rots"], True))
    out, aux = td.render(means, opac, t64(cam["view"]), t64(cam["proj"]), t64(cam["campos"]), cam["W"], cam["H"], cam["tanx"], cam["tany"],
                         t64(bg), D=D, means2D=m2d, **ref_in)
    dpix = np.random.default_rng(10).normal(size=tuple(out.shape)).astype(np.float32)
    (out * t64(dpix)).sum().backward()
    # HIP
    g_means, g_opac = T(sc["means"], True), T(sc["opac"], True)
    g_m2d = torch.zeros_like(g_means, requires_grad=True)
    if pre:
        g_in = dict(colors_precomp=T(sc["colors_precomp"], True), cov3D_precomp=T(sc["cov3D_precomp"], True))
    else:
        g_in = dict(shs=T(sc["shs"], True), scales=T(sc["scales"], True), rotations=T(sc["rots"], True))
    color, radii = GaussianRasterizer(settings(cam, bg, D))(g_means, g_m2d, g_opac, **g_in)
    (color * T(dpix)).sum().backward()
    torch.cuda.synchronize()
    assert np.array_equal(radii.cpu().numpy(), aux["radii"].numpy())
    err = np.abs(color.detach().cpu().numpy() - out.detach().numpy())
    assert (err > 1e-4).sum() <= 2 and err.max() <= 2.0 / 255.0 + 1e-3, (int((err > 1e-4).sum()), float(err.max()))     # (a flipped threshold or two at most)
    pairs = [(g_means.grad, means.grad), (g_opac.grad.reshape(-1), opac.grad.reshape(-1)), (g_m2d.grad[:, :2], m2d.grad[:, :2])]
    if pre:
        pairs += [(g_in["colors_precomp"].grad, ref_in["colors_precomp"].grad), (g_in["cov3D_precomp"].grad, ref_in["cov3D_precomp"].grad)]
    else:
        pairs += [(g_in["shs"].grad, ref_in["shs"].grad), (g_in["scales"].grad, ref_in["scales"].grad), (g_in["rotations"].grad, ref_in["rots"].grad)]
    for k, (got, ref) in enumerate(pairs):
        _grad_gate(got.cpu().numpy(), ref.numpy(), "tensor %d" % k)


@pytest.mark.parametrize("case", ["small", "100k"])
def test_forward_exact_exponent_build_differs_only_on_accounted_pixels(oracle, case):
    """render_fwd_kernel<.., EXACT>: the same kernel with the exponents of a group from the pixel-relative form (the backward's, round
    2's: |e - e_exact| ~ 5e-7) instead of the matrix core's polynomial (~1e-5).  What the polynomial costs in accuracy, shown: (i) the
    two builds agree to 3e-5 on every pixel except pixels with an entry AT a decision threshold (the same accounting as the forward
    gate, run on the difference of the two images); (ii) both pass the gate against the oracle, and the error of the pixels without
    a flip is printed for both - the matrix-core build must stay within half the 1e-4 budget."""
    import ctypes as C
    from gpu_utils import forward_state
    from helpers import account_outlier_pixels
    from gaussianmesh_amd import _lib, scenes
    if case == "small":
        sc, cam = small_scene(P=3000, W=160, H=96, seed=13, D=3, scale_lo=0.02, scale_hi=0.3)
    else:
        sc = scenes.make_cloud(100_000, seed=3, scale_lo=0.005, scale_hi=0.06)
        cam = scenes.orbit_camera(5, 16, 640, 360)
    W, H = cam["W"], cam["H"]
    bg = np.array([0.1, 0.4, 0.8], np.float32)
    fw = oracle.forward_full(sc, cam, bg, D=3)
    lib = C.CDLL(_lib.lib()._name)                                        # the verification switch is not part of the public header
    poly = forward_state(sc, cam, bg, D=3, tile_cull=2)
    lib.gm_debug_forward_exact_exponent(1)
    try:
        exact = forward_state(sc, cam, bg, D=3, tile_cull=2)
    finally:
        lib.gm_debug_forward_exact_exponent(0)
    again = forward_state(sc, cam, bg, D=3, tile_cull=2)
    assert np.array_equal(again["color"], poly["color"]) and not np.array_equal(exact["color"], poly["color"])     # the switch switches, and back
    n_out, n_bad, worst = account_outlier_pixels(dict(fw, color=exact["color"].astype(np.float64)), poly["color"], W, H, tol=3e-5)
    d = np.abs(poly["color"].astype(np.float64) - exact["color"]).max(axis=0)
    print("exact vs matrix-core exponent (%s): %d pixel(s) differ by more than 3e-5 (%d unexplained, worst %.3g); the others by at most %.3g"
          % (case, n_out, n_bad, worst, d[d <= 3e-5].max()))
    assert n_bad == 0 and n_out <= max(4, 2e-4 * W * H)
    assert_forward_gate(fw, exact["color"], W, H, FWD_TOL, "exact exponent " + case, plain_tol=2.5e-5)
    assert_forward_gate(fw, poly["color"], W, H, FWD_TOL, "matrix-core exponent " + case, plain_tol=5e-5)
    # the backward state agrees wherever no flip happened
    assert (exact["n_contrib"] != poly["n_contrib"]).sum() <= max(4, 2e-4 * W * H)
    if case == "small":
        # GM_FWD_EXACT_EXPONENT (what the operator sets for a forward a backward follows) selects that build for ONE frame; it is refused
        # together with GM_FWD_IMAGE_ONLY
        from gpu_utils import T
        from gaussianmesh_amd import rasterizer as Rz
        a = (T(bg), T(sc["means"]), None, T(sc["opac"]), T(sc["scales"]), T(sc["rots"]), 1.0, None, T(cam["view"]), T(cam["proj"]), cam["tanx"],
             cam["tany"], H, W, T(sc["shs"]), 3, T(cam["campos"]), False, False)
        flagged = Rz.rasterize_forward_begin(*a, emission_policy=2).finish(exact_exponent=True)[1].cpu().numpy()
        plain = Rz.rasterize_forward_begin(*a, emission_policy=2).finish()[1].cpu().numpy()
        assert np.array_equal(flagged, exact["color"]) and np.array_equal(plain, poly["color"])
        with pytest.raises(_lib.GmeshError):
            Rz.rasterize_forward_begin(*a, emission_policy=2).finish(exact_exponent=True, image_only=True)
