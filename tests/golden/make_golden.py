"""Generates tests/golden/*.npz by EXECUTING reference python functions in the dev container.

Runs only where /root/reference exists (never on the GPU box; the fixtures are committed).
The reference modules `import jittor` at module top although the functions used here are pure
python/numpy arithmetic (eval_sh: "Works with torch/np/jnp").  jittor is not installed in this
image, so an INERT placeholder module is registered for the import statement only: any attribute
access on it raises, which proves no jittor functionality is substituted - every number in the
fixtures is computed by the reference's own function bodies on numpy arrays.

Functions executed (reference file:line):
  utils/sh_utils.py:57-112       eval_sh           (train-time python SH path)
  edittool/sh_utils.py:33-88     eval_sh           (edit-tool python SH path)
  edittool/general_utils.py:73-88 get_barycentric_coordinate
  utils/graphics_utils.py:38-50  getWorld2View2
  utils/graphics_utils.py:73-77  fov2focal / focal2fov
  utils/general_utils.py:28-62   get_expon_lr_func (position learning-rate schedule of the training loop)
  arguments/__init__.py:70-93    OptimizationParams defaults (learning rates, lambda_dssim, alpha_mrloss, densify schedule)
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


class _Inert(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        raise RuntimeError("inert jittor placeholder touched: %s" % name)


def _load(relpath, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    assert os.path.isdir(REF), "reference tree not present; fixtures can only be regenerated in the dev container"
    sys.modules.setdefault("jittor", _Inert("jittor"))
    sh_train = _load("utils/sh_utils.py", "ref_utils_sh_utils")
    sh_edit = _load("edittool/sh_utils.py", "ref_edittool_sh_utils")
    gen = _load("edittool/general_utils.py", "ref_edittool_general_utils")
    gfx = _load("utils/graphics_utils.py", "ref_utils_graphics_utils")

    rng = np.random.default_rng(20240928)
    # --- SH polynomial, degrees 0..3, float32 inputs as in the rasterizer -----------------
    N = 257
    shs = rng.normal(0, 0.4, size=(N, 16, 3)).astype(np.float32)          # rasterizer layout [N,M,3]
    dirs = rng.normal(size=(N, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dirs = dirs.astype(np.float32)
    sh_view = np.ascontiguousarray(shs.transpose(0, 2, 1))                # [N,3,M] as the python callers build it
    fix = dict(shs=shs, dirs=dirs)
    for deg in range(4):
        a = sh_train.eval_sh(deg, sh_view, dirs)
        b = sh_edit.eval_sh(deg, sh_view, dirs)
        assert np.array_equal(a, b)
        fix["rgb_deg%d" % deg] = np.asarray(a, np.float32)
    np.savez_compressed(os.path.join(OUT, "sh_eval.npz"), **fix)

    # --- barycentric weights -------------------------------------------------------------
    Nb = 200
    p1, p2, p3 = (rng.normal(size=(Nb, 3)) for _ in range(3))
    w = rng.dirichlet([1, 1, 1], size=Nb)
    g = w[:, :1] * p1 + w[:, 1:2] * p2 + w[:, 2:3] * p3                   # points inside the faces
    g[:20] += 0.05 * rng.normal(size=(20, 3))                             # some off-plane points
    coord = gen.get_barycentric_coordinate(g, p1, p2, p3)
    np.savez_compressed(os.path.join(OUT, "barycentric.npz"), g=g, p1=p1, p2=p2, p3=p3, coord=coord)

    # --- camera extrinsics -----------------------------------------------------------------
    Rs, Ts, TRs, SCs, W2V = [], [], [], [], []
    for k in range(6):
        A = rng.normal(size=(3, 3)); Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        t = rng.normal(size=3) * 3
        tr = rng.normal(size=3) * (k % 2)
        sc = 1.0 + 0.5 * (k % 3)
        Rs.append(Q); Ts.append(t); TRs.append(tr); SCs.append(sc)
        W2V.append(gfx.getWorld2View2(Q, t, tr, sc))
    foc = np.array([[gfx.fov2focal(f, p), gfx.focal2fov(gfx.fov2focal(f, p), p)] for f, p in
                    [(0.5, 640), (1.0471975511965976, 1920), (1.2, 1080)]])
    np.savez_compressed(os.path.join(OUT, "camera.npz"), R=np.array(Rs), T=np.array(Ts), translate=np.array(TRs),
                        scale=np.array(SCs), W2V=np.array(W2V), foc=foc)
    # --- training schedule: learning-rate function and the optimisation defaults -------------------
    gu = _load("utils/general_utils.py", "ref_utils_general_utils")
    import argparse
    args_mod = _load("arguments/__init__.py", "ref_arguments")
    op = args_mod.OptimizationParams(argparse.ArgumentParser())
    defaults = {k: (float(v) if not isinstance(v, bool) else bool(v)) for k, v in vars(op).items() if not k.startswith("_")}
    steps = np.array([-1, 0, 1, 10, 100, 1000, 7000, 15000, 29999, 30000, 40000], np.int64)
    cases = [(op.position_lr_init, op.position_lr_final, 0, op.position_lr_delay_mult, op.position_lr_max_steps),
             (1e-2, 1e-4, 100, 0.1, 1000), (0.0, 0.0, 0, 1.0, 10), (3e-3, 3e-3, 50, 0.5, 500)]
    lr = np.array([[gu.get_expon_lr_func(a, b, lr_delay_steps=d, lr_delay_mult=m, max_steps=n)(int(st)) for st in steps]
                   for a, b, d, m, n in cases], np.float64)
    np.savez_compressed(os.path.join(OUT, "schedule.npz"), steps=steps, cases=np.array(cases, np.float64), lr=lr,
                        opt_names=np.array(sorted(defaults)), opt_values=np.array([float(defaults[k]) for k in sorted(defaults)]))
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
