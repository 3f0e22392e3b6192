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
            assert_forward_gate(fw, color, cam["W"], cam["H"], 1e-4, "fuzz seed %d" % seed)
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
            assert_forward_gate(fw, color, cam["W"], cam["H"], 1e-4, "needle seed %d" % seed)
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
