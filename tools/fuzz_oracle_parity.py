"""Fuzz (run on a GPU box): forward image and all gradients of random small scenes against the CPU oracle - and, where a gradient tensor is
more than 1e-3 (of its largest entry) away from the oracle, against float64 autograd (oracle/torch_dense.py), as
tests/test_gpu_fuzz_parity.py does for its fixed seeds.  Every third seed is a scene of 100:1 needle splats, on which the reference's float32
conic -> cov2D backward step (and with it the C oracle) is off by up to 1e-2.
    python tools/fuzz_oracle_parity.py [NSEEDS]"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from oracle import oracle
from test_gpu_parity import _grads_gpu, _rel
from test_gpu_fuzz_parity import _float64_truth, _tensors
from helpers import fuzz_scene
worst_f, worst_g, worst_t, bad, over = 0.0, 0.0, 0.0, 0, 0
NSEEDS = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for seed in range(NSEEDS):
    sc, cam, bg, D, pre_cov, pre_col, dpix = fuzz_scene(seed); P, W, H = sc["means"].shape[0], cam["W"], cam["H"]
    fw = oracle.forward_full(sc, cam, bg, D=D, use_precomp_cov=pre_cov, use_precomp_color=pre_col)
    bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=D, use_precomp_cov=pre_cov, use_precomp_color=pre_col)
    color, radii, g = _grads_gpu(sc, cam, bg, dpix, D, pre_cov, pre_col)
    err = np.abs(color - fw["color"])
    f_out = (err > 1e-4).mean()
    rels = {k: _rel(np.asarray(g[k]).reshape(np.asarray(r).shape), r) for k, r in _tensors(bw, pre_cov, pre_col)}
    note = ""
    ok = np.array_equal(radii, fw["geo"]["radii"]) and f_out <= 1e-4 and err.max() <= 5e-3
    big = {k: v for k, v in rels.items() if v > 1e-3}
    if big:                                          # the oracle's float32 formula or this implementation?  ask float64 autograd
        over += 1
        truth = _float64_truth(sc, cam, bg, D, pre_cov, pre_col, dpix)[1]
        refs = dict(_tensors(bw, pre_cov, pre_col))
        for k in big:
            t = truth[k].reshape(np.asarray(refs[k]).shape)
            hip_t, orc_t = _rel(np.asarray(g[k]).reshape(t.shape), t), _rel(refs[k], t)
            worst_t = max(worst_t, hip_t)
            note += "  %s: vs oracle %.2e, vs float64 %.2e (oracle vs float64 %.2e)" % (k, big[k], hip_t, orc_t)
            ok = ok and hip_t <= 1e-3 and hip_t <= 2.0 * orc_t
    worst_f = max(worst_f, err.max()); worst_g = max(worst_g, max(rels.values()))
    bad += (not ok)
    print("seed", seed, "needles" if seed % 3 == 0 else "       ", "P", P, "%dx%d" % (W, H), "D", D, "precomp", pre_cov, pre_col, "fwd max %.2e (>1e-4: %.1e)" % (err.max(), f_out),
          "grad rel max %.2e" % max(rels.values()), note, "" if ok else "  <-- FAIL", flush=True)
print("failures: %d of %d scenes; worst fwd %.2e; worst gradient vs the C oracle %.2e; scenes with a tensor more than 1e-3 from the oracle: %d, "
      "worst of those tensors vs float64 autograd %.2e" % (bad, NSEEDS, worst_f, worst_g, over, worst_t))
