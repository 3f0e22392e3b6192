import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from gaussianmesh_amd import scenes, rasterizer
from oracle import oracle
from test_gpu_parity import _grads_gpu, _rel
seed = 21
rng = np.random.default_rng(100 + seed)
P = int(rng.integers(50, 3000))
lo = float(10 ** rng.uniform(-2.3, -1)); hi = lo * float(10 ** rng.uniform(0.3, 1.6))
sc = scenes.make_cloud(P, seed=seed, scale_lo=lo, scale_hi=hi)
if seed % 3 == 0:
    sc["scales"][:, 0] *= 10.0
W = int(rng.integers(17, 160)); H = int(rng.integers(17, 120))
cam = scenes.orbit_camera(int(rng.integers(0, 16)), 16, W, H, radius=float(rng.uniform(2.0, 9.0)))
bg = rng.random(3).astype(np.float32)
D = int(rng.integers(0, 4))
pre_cov, pre_col = bool(seed % 2), bool((seed // 2) % 2)
if pre_cov:
    sc["cov3D_precomp"] = scenes.strip_symmetric(scenes.cov3d_from_scale_rot(sc["scales"], sc["rots"])).astype(np.float32)
if pre_col:
    sc["colors_precomp"] = rng.random((P, 3)).astype(np.float32)
dpix = rng.normal(size=(3, H, W)).astype(np.float32)
fw = oracle.forward_full(sc, cam, bg, D=D, use_precomp_cov=pre_cov, use_precomp_color=pre_col)
bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=D, use_precomp_cov=pre_cov, use_precomp_color=pre_col)
ref = bw["dmean3D"]
print("P", P, W, H, "D", D, "scales", lo, hi, "max |dmean|", np.abs(ref).max())
import hashlib
for rep in range(3):
    for mode in (0, 2):
        rasterizer.set_default_emission_policy(mode)
        color, radii, g = _grads_gpu(sc, cam, bg, dpix, D, pre_cov, pre_col)
        gm = np.asarray(g["means"]).reshape(ref.shape)
        print("rep", rep, "mode", mode, "dmean[254] %r" % gm[254].tolist(), "img md5", hashlib.md5(np.ascontiguousarray(color).tobytes()).hexdigest()[:8],
              "grad md5", hashlib.md5(np.ascontiguousarray(gm).tobytes()).hexdigest()[:8], "cov[254]", np.asarray(g["cov"]).reshape(-1, 6)[254, :2].tolist())
