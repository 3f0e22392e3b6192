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
res = {}
for mode in (0, 1, 2, 3):
    rasterizer.set_default_emission_policy(mode)
    color, radii, g = _grads_gpu(sc, cam, bg, dpix, D, pre_cov, pre_col)
    d = np.asarray(g["means"]).reshape(ref.shape) - ref
    i = np.unravel_index(np.abs(d).argmax(), d.shape)
    res[mode] = np.asarray(g["means"]).reshape(ref.shape)
    print("mode", mode, "img err max %.3e" % np.abs(color - fw["color"]).max(), "dmeans rel %.3e" % _rel(res[mode], ref), "worst at", i, "gpu", res[mode][i], "ref", ref[i],
          "radius", fw["geo"]["radii"][i[0]])
for mode in (1, 2, 3):
    print("mode", mode, "vs mode 0: rel %.3e" % _rel(res[mode], res[0]))
