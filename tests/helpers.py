"""Shared scene builders for the tests (small enough for the CPU oracle to finish in seconds)."""
import numpy as np

from gaussianmesh_amd import scenes


def small_scene(P=300, W=48, H=40, seed=0, D=3, scale_lo=0.05, scale_hi=0.6, cam_k=1, cam_K=7, radius=6.0,
                behind=True):
    sc = scenes.make_cloud(P, seed=seed, scale_lo=scale_lo, scale_hi=scale_hi, D=D)
    if behind:  # put a few Gaussians behind / very near the camera to exercise the cull branch
        cam0 = scenes.orbit_camera(cam_k, cam_K, W, H, radius=radius)
        n = max(2, P // 50)
        sc["means"][:n] = cam0["campos"][None, :] * np.linspace(0.97, 1.3, n)[:, None].astype(np.float32)
    cam = scenes.orbit_camera(cam_k, cam_K, W, H, radius=radius)
    S = scenes.cov3d_from_scale_rot(sc["scales"], sc["rots"])
    sc["cov3D_precomp"] = scenes.strip_symmetric(S)
    rng = np.random.default_rng(seed + 100)
    sc["colors_precomp"] = rng.uniform(0, 1, size=(P, 3)).astype(np.float32)
    return sc, cam
