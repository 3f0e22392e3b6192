"""Pins the oracle (and the host-side helpers) against fixtures produced by EXECUTING reference python code
(tests/golden/make_golden.py): SH polynomial, barycentric weights, camera extrinsics."""
import os

import numpy as np

G = os.path.join(os.path.dirname(__file__), "golden")


def test_sh_polynomial_matches_reference_eval_sh(oracle):
    f = np.load(os.path.join(G, "sh_eval.npz"))
    for deg in range(4):
        got = oracle.sh_to_rgb(deg, f["shs"], f["dirs"])
        ref = f["rgb_deg%d" % deg]
        # same polynomial, same constants; the reference python groups a few products differently -> <= 2 ulp
        assert np.abs(got - ref).max() <= 2e-6, deg


def test_barycentric_matches_reference(oracle):
    f = np.load(os.path.join(G, "barycentric.npz"))
    w = oracle.bary_weights(f["g"], f["p1"], f["p2"], f["p3"])
    assert np.abs(w - f["coord"]).max() <= 1e-12
    assert np.allclose(w.sum(1), 1.0)


def test_bind_weights_helper_matches_reference():
    from gaussianmesh_amd.deform import barycentric_weights
    f = np.load(os.path.join(G, "barycentric.npz"))
    w = barycentric_weights(f["g"], f["p1"], f["p2"], f["p3"])
    assert np.abs(w - f["coord"]).max() <= 1e-12


def test_world2view_matches_reference():
    from gaussianmesh_amd import scenes
    f = np.load(os.path.join(G, "camera.npz"))
    for R, T, tr, sc, W2V in zip(f["R"], f["T"], f["translate"], f["scale"], f["W2V"]):
        assert np.array_equal(scenes.world2view2(R, T, tr, float(sc)), W2V)


def test_camera_conventions():
    """scene/cameras.py:47-50: view stored transposed, full_proj = view @ proj, camera centre = inv(view)[3,:3]."""
    from gaussianmesh_amd import scenes
    cam = scenes.look_at_camera((3.0, 1.0, -2.0), (0.2, 0.1, 0.3), 640, 360)
    assert np.allclose(cam["campos"], (3.0, 1.0, -2.0), atol=1e-5)
    p = np.array([0.2, 0.1, 0.3, 1.0], np.float32)           # the look-at target projects to the image centre
    pv = p @ cam["view"]
    assert pv[2] > 0 and abs(pv[0]) < 1e-5 and abs(pv[1]) < 1e-5
    ph = p @ cam["proj"]
    assert abs(ph[0] / ph[3]) < 1e-5 and abs(ph[1] / ph[3]) < 1e-5
    assert np.isclose(cam["tanx"] / cam["tany"], 640 / 360)


def test_torch_sh_route_matches_reference_eval_sh():
    """renderer.eval_sh_torch (the differentiable pipe.convert_SHs_python route) on the reference-executed fixture."""
    import torch
    from gaussianmesh_amd.renderer import eval_sh_torch
    f = np.load(os.path.join(G, "sh_eval.npz"))
    for deg in range(4):
        got = eval_sh_torch(deg, torch.tensor(f["shs"]), torch.tensor(f["dirs"])).numpy()
        assert np.abs(got - f["rgb_deg%d" % deg]).max() <= 1e-6, deg
    sh = torch.tensor(f["shs"][:5], dtype=torch.float64, requires_grad=True)
    d = torch.tensor(f["dirs"][:5], dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda a, b: eval_sh_torch(3, a, b), (sh, d), eps=1e-6, atol=1e-6)
