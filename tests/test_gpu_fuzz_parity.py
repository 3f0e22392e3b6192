"""Random-scene parity INSIDE the suite (round 5; until round 4 this lived in tools/fuzz_oracle_parity.py and its failures in a log).

Ordinary scenes (40 seeds): the north-star bars as they stand - radii equal, the strict forward gate, every gradient tensor within
1e-3 of the tensor's size of the C oracle's.

Needle scenes (every third seed stretches each splat's first axis ten-fold: 100:1 splats): the float32 formula of
RAST/backward.cu:196-215 (conic -> cov2D, three quadratic forms whose terms cancel by det / (a c)) is ITSELF 3e-3 ... 1.2e-2 away
from float64 autograd on dL/dmeans, dL/dscales, dL/drots of such scenes (the C oracle restates it in float32, tools/needle_stages.py
shows the loss is in that block and not in the blend).  The HIP preprocess backward evaluates that block in binary64, so on these
scenes it is the ORACLE that is off.  What is asserted for every tensor of a needle scene: within 1e-3 of the oracle, OR - with
oracle/torch_dense.py (float64, autograd) as the truth - the HIP path within 1e-3 of the truth and no further from it than twice the
oracle's own distance (the review's criterion).  The table is printed either way.
"""
import numpy as np
import pytest
import torch

from helpers import fuzz_scene, assert_forward_gate
from test_gpu_parity import _grads_gpu, _rel

pytestmark = pytest.mark.gpu

ORDINARY = [s for s in range(60) if s % 3]                     # 40 scenes
RECORDED = [120, 126, 138, 162, 168, 216]                      # above 1e-3 vs the C oracle in round 4's 240-scene runs (profiles/r04_fuzz_oracle_parity_240*.txt)
NEEDLES = RECORDED + [0, 21, 33, 45, 57, 90]                   # + six more needle scenes (seed 21: the 177-px needle of tools/diag_fuzz_seed.py)


def _tensors(bw, pre_cov, pre_col):
    pairs = [("means", bw["dmean3D"]), ("opac", bw["dopacity"])]
    pairs += [("colors", bw["dcolor"])] if pre_col else [("shs", bw["dsh"])]
    pairs += [("cov", bw["dcov3D"])] if pre_cov else [("scales", bw["dscale"]), ("rots", bw["drot"])]
    return pairs


def _float64_truth(sc, cam, bg, D, pre_cov, pre_col, dpix):
    from oracle import torch_dense as td
    t64 = lambda a, rg=False: torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=rg)
    means, opac = t64(sc["means"], True), t64(sc["opac"], True)
    kw, leaves = {}, {"means": means, "opac": opac}
    if pre_col:
        kw["colors_precomp"] = leaves["colors"] = t64(sc["colors_precomp"], True)
    else:
        kw["shs"] = leaves["shs"] = t64(sc["shs"], True)
    if pre_cov:
        kw["cov3D_precomp"] = leaves["cov"] = t64(sc["cov3D_precomp"], True)
    else:
        kw["scales"] = leaves["scales"] = t64(sc["scales"], True); kw["rots"] = leaves["rots"] = t64(sc["rots"], True)
    out, _ = td.render(means, opac, t64(cam["view"]), t64(cam["proj"]), t64(cam["campos"]), cam["W"], cam["H"], cam["tanx"], cam["tany"],
                       t64(bg), D=D, **kw)
    (out * t64(dpix)).sum().backward()
    return out.detach().numpy(), {k: v.grad.numpy() for k, v in leaves.items()}


def test_ordinary_random_scenes_meet_the_bars(oracle):
    problems, worst_f, worst_g = [], 0.0, 0.0
    for seed in ORDINARY:
        sc, cam, bg, D, pre_cov, pre_col, dpix = fuzz_scene(seed)
        fw = oracle.forward_full(sc, cam, bg, D=D, use_precomp_cov=pre_cov, use_precomp_color=pre_col)
        bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=D, use_precomp_cov=pre_cov, use_precomp_color=pre_col)
        color, radii, g = _grads_gpu(sc, cam, bg, dpix, D, pre_cov, pre_col)
        if not np.array_equal(radii, fw["geo"]["radii"]):
            problems.append((seed, "radii"))
        try:
            assert_forward_gate(fw, color, cam["W"], cam["H"], 1e-4, "fuzz seed %d" % seed, plain_tol=5e-5, bg=bg)
        except AssertionError as e:
            problems.append((seed, str(e)))
        rels = {k: _rel(np.asarray(g[k]).reshape(np.asarray(r).shape), r) for k, r in _tensors(bw, pre_cov, pre_col)}
        worst_f = max(worst_f, float(np.abs(color - fw["color"]).max())); worst_g = max(worst_g, max(rels.values()))
        if max(rels.values()) > 1e-3:
            problems.append((seed, rels))
    print("ordinary fuzz scenes: %d, worst image error %.2e (flips included), worst gradient error %.2e of the tensor's size" % (len(ORDINARY), worst_f, worst_g))
    assert not problems, problems


def test_needle_scenes_against_the_oracle_and_float64_truth(oracle):
    problems, rows = [], []
    for seed in NEEDLES:
        sc, cam, bg, D, pre_cov, pre_col, dpix = fuzz_scene(seed)
        fw = oracle.forward_full(sc, cam, bg, D=D, use_precomp_cov=pre_cov, use_precomp_color=pre_col)
        bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=D, use_precomp_cov=pre_cov, use_precomp_color=pre_col)
        color, radii, g = _grads_gpu(sc, cam, bg, dpix, D, pre_cov, pre_col)
        if not np.array_equal(radii, fw["geo"]["radii"]):
            problems.append((seed, "radii"))
        try:
            assert_forward_gate(fw, color, cam["W"], cam["H"], 1e-4, "needle seed %d" % seed, plain_tol=5e-5, bg=bg)
        except AssertionError as e:
            problems.append((seed, str(e)))
        truth = None
        for k, ref in _tensors(bw, pre_cov, pre_col):
            hip = np.asarray(g[k]).reshape(np.asarray(ref).shape)
            vs_oracle = _rel(hip, ref)
            if vs_oracle <= 1e-3 and seed not in RECORDED:
                rows.append((seed, k, vs_oracle, None, None))
                continue
            if truth is None:
                truth = _float64_truth(sc, cam, bg, D, pre_cov, pre_col, dpix)[1]
            t = truth[k].reshape(hip.shape)
            hip_t, orc_t = _rel(hip, t), _rel(ref, t)
            rows.append((seed, k, vs_oracle, hip_t, orc_t))
            if vs_oracle > 1e-3 and not (hip_t <= 1e-3 and hip_t <= 2.0 * orc_t):
                problems.append((seed, k, "HIP vs oracle %.2e, HIP vs float64 %.2e, oracle vs float64 %.2e" % (vs_oracle, hip_t, orc_t)))
    print("needle scenes (error of a gradient tensor as a fraction of its largest entry):")
    print("  seed  tensor   |HIP - oracle|  |HIP - float64|  |oracle - float64|")
    for seed, k, a, b, c in rows:
        print("  %4d  %-7s  %.2e       %s        %s%s" % (seed, k, a, "%.2e" % b if b is not None else "   -    ", "%.2e" % c if c is not None else "   -    ",
                                                         "   <-- oracle (float32 formula) off by more than 1e-3" if c is not None and c > 1e-3 else ""))
    n_over = sum(1 for r in rows if r[2] > 1e-3)
    print("tensors more than 1e-3 from the C oracle: %d of %d; every one of them within 1e-3 of float64 autograd: %s" % (
        n_over, len(rows), all(r[3] <= 1e-3 for r in rows if r[2] > 1e-3)))
    assert not problems, problems


def _stage_grads_gpu(sc, cam, bg, dpix, D, pre_cov, pre_col):
    """Forward + backward through the low-level pair, keeping the BLEND stage's outputs (dL/dmean2D, dL/dconic, dL/dcolour, dL/dopacity)
    next to the final gradients: what the preprocess backward consumed and what it made of it."""
    from gpu_utils import T
    from gaussianmesh_amd import rasterizer as Rz
    t = lambda k: T(sc[k])
    means, opac = t("means"), t("opac")
    shs = None if pre_col else t("shs"); colors = t("colors_precomp") if pre_col else None
    scales = None if pre_cov else t("scales"); rots = None if pre_cov else t("rots"); cov = t("cov3D_precomp") if pre_cov else None
    H, W = cam["H"], cam["W"]
    a = (T(bg), means, colors, opac, scales, rots, 1.0, cov, T(cam["view"]), T(cam["proj"]), cam["tanx"], cam["tany"])
    nr, color, radii, geom, binning, img = Rz.rasterize_forward_begin(*a, H, W, shs, D, T(cam["campos"])).finish(exact_exponent=True)
    out = Rz.rasterize_backward(a[0], means, radii, colors, scales, rots, 1.0, cov, a[8], a[9], cam["tanx"], cam["tany"], T(dpix), shs, D,
                                T(cam["campos"]), geom, nr, binning, img, False, want_conic=True)
    torch.cuda.synchronize()
    names = ("dmean2D", "dcolor", "dopacity", "dmean3D", "dcov3D", "dsh", "dscale", "drot", "dconic")
    return {k: (None if v is None else v.cpu().numpy()) for k, v in zip(names, out)}


def test_needle_scenes_reference_float32_block_is_the_only_divergence(oracle):
    """The HIP preprocess backward deviates from RAST/backward.cu in ONE block by design: conic -> cov2D (backward.cu:196-215) is evaluated in
    binary64 (gm_preprocess.hip).  gm_debug_backward_cov2d_float32 (a test switch outside the public header) restores the reference's
    float32 statement.  Proven here ON THE GPU PATH for all twelve needle scenes, stage by stage (an end-to-end comparison cannot show it:
    that block amplifies the ~1e-4 float-atomic noise of its inputs by a c / det ~ 500, in the HIP path and in the oracle alike):
      1. blend stage: HIP's dL/dmean2D, dL/dconic, dL/dcolour, dL/dopacity within 1e-3 of the C oracle's (orc_render_bwd);
      2. preprocess stage with the switch ON: the C oracle's preprocess backward (orc_preprocess_bwd, the float32 formula) applied to the
         SAME blend-stage gradients the HIP kernel consumed reproduces HIP's dL/dmeans3D, dL/dcov3D | dL/dscales, dL/drots, dL/dSH to 1e-3
         (measured: ~1e-6 - same inputs, same expressions, contraction off on both sides);
      3. with the switch OFF (the product) the same comparison differs by more than 1e-3 on the recorded seeds - the binary64 block, and
         nothing else, is what separates the product from the reference formula."""
    import ctypes as C
    from gaussianmesh_amd import _lib
    lib = C.CDLL(_lib.lib()._name)
    rows, problems, n_off_differs = [], [], 0
    for seed in NEEDLES:
        sc, cam, bg, D, pre_cov, pre_col, dpix = fuzz_scene(seed)
        fw = oracle.forward_full(sc, cam, bg, D=D, use_precomp_cov=pre_cov, use_precomp_color=pre_col)
        bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=D, use_precomp_cov=pre_cov, use_precomp_color=pre_col)
        res = {}
        for on in (1, 0):
            lib.gm_debug_backward_cov2d_float32(on)
            try:
                res[on] = _stage_grads_gpu(sc, cam, bg, dpix, D, pre_cov, pre_col)
            finally:
                lib.gm_debug_backward_cov2d_float32(0)
        g = res[1]
        # 1. the blend stage against the oracle's
        stage = {"dmean2D": _rel(g["dmean2D"][:, :2], bw["dmean2D"][:, :2]), "dconic": _rel(g["dconic"].reshape(-1, 4)[:, [0, 1, 3]], bw["dconic"][:, [0, 1, 3]]),
                 "dcolor": _rel(g["dcolor"], bw["dcolor"]), "dopacity": _rel(g["dopacity"].reshape(-1), bw["dopacity"].reshape(-1))}
        for k, v in stage.items():
            if v > 1e-3:
                problems.append((seed, "blend stage", k, v))
        # 2. the oracle's preprocess backward on HIP's own blend-stage gradients
        def through_oracle(gg):
            return oracle.preprocess_bwd(sc["means"], fw["geo"], cam["view"], cam["proj"], cam["campos"], cam["W"], cam["H"], cam["tanx"], cam["tany"],
                                         gg["dmean2D"], gg["dconic"].reshape(-1, 4), gg["dcolor"], D=D, shs=None if pre_col else sc["shs"],
                                         scales=None if pre_cov else sc["scales"], rots=None if pre_cov else sc["rots"])
        def compare(gg):
            dmean3D, dcov, dsh, dscale, drot = through_oracle(gg)
            out = {"means": _rel(gg["dmean3D"], dmean3D)}
            if pre_cov:
                out["cov"] = _rel(gg["dcov3D"], dcov)
            else:
                out["scales"] = _rel(gg["dscale"], dscale); out["rots"] = _rel(gg["drot"], drot)
            if not pre_col:
                out["shs"] = _rel(gg["dsh"].reshape(dsh.shape), dsh)
            return out
        on_, off_ = compare(res[1]), compare(res[0])
        for k, v in on_.items():
            if v > 1e-3:
                problems.append((seed, "preprocess stage, float32 block", k, v))
        n_off_differs += 1 if max(off_.values()) > 1e-3 else 0
        rows.append((seed, max(stage.values()), max(on_.values()), max(off_.values())))
    print("needle scenes, stage by stage (largest error over the stage's tensors, as a fraction of the tensor's largest entry):")
    print("  seed  blend stage vs oracle   preprocess stage vs oracle formula: float32 block | binary64 block (the product)")
    for seed, a, b, c in rows:
        print("  %4d  %.2e                %.2e | %.2e%s" % (seed, a, b, c, "   <-- the product deviates from the float32 formula here" if c > 1e-3 else ""))
    assert not problems, problems
    assert n_off_differs >= 3, "the binary64 block should separate the product from the float32 formula on the recorded needle seeds (%d)" % n_off_differs
