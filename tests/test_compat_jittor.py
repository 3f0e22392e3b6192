"""CPU: the torch-backed `jittor` subset (gaussianmesh_amd/compat).

Part 1 checks the bridged semantics on their own.  Part 2 runs only where the reference tree is mounted
(/root/reference, this container): it executes the reference's own model / utility python on top of the subset to show
the API surface is sufficient -- an API-coverage check, not an oracle pin (the subset is not Jittor)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

import gaussianmesh_amd.compat as compat

REF = "/root/reference"


@pytest.fixture(scope="module")
def jt():
    return compat.install(operators=True)


def test_creation_dtypes_and_reductions(jt):
    assert jt.array([1.0, 2.0]).dtype == torch.float32 and jt.array(np.zeros(3)).dtype == torch.float32
    assert jt.array([1, 2]).dtype == torch.int32 and jt.array(np.arange(3), dtype=jt.float).dtype == torch.float32
    assert jt.zeros((2, 3)).shape == (2, 3) and jt.zeros(2, 3).shape == (2, 3) and jt.empty(0).shape == (0,)
    assert jt.ones((4, 1), dtype=jt.float32).sum() == 4 and jt.rand((3)).shape == (3,) and jt.rand((2, 3)).shape == (2, 3)
    x = jt.array([[1.0, 5.0], [3.0, 2.0]])
    assert torch.equal(jt.max(x, dim=1), torch.tensor([5.0, 3.0]))          # values only
    assert float(jt.max(x)) == 5.0 and float(jt.sum(x)) == 11.0
    assert torch.equal(jt.clamp(x, min_v=2.0, max_v=4.0), torch.tensor([[2.0, 4.0], [3.0, 2.0]]))
    assert torch.allclose(jt.norm(x, dim=1), torch.linalg.norm(x, dim=1))
    assert torch.allclose(jt.normalize(x), torch.nn.functional.normalize(x, dim=1))
    assert torch.allclose(jt.nn.softmax(x, dim=1).sum(1), torch.ones(2))
    assert torch.equal(jt.concat([x, x], dim=0), torch.cat([x, x])) and jt.init.eye(3).shape == (3, 3)
    assert torch.allclose(jt.linalg.inv(x) @ x, torch.eye(2), atol=1e-6)
    with pytest.raises(AttributeError):
        jt.this_symbol_is_not_in_the_subset


def test_var_methods(jt):
    x = jt.array([1.0, 2.0, 3.0]).requires_grad_(True)
    y = (x * 2)
    assert not y.stop_grad().requires_grad and y.sync() is y
    assert isinstance(y.numpy(), np.ndarray) and y.numpy().tolist() == [2.0, 4.0, 6.0]     # works on a graph node
    m = jt.zeros(3)
    m.update(jt.array([1.0, 2.0]))                      # storage rebind with a new shape (densifier on Adam state)
    assert m.shape == (2,) and m.tolist() == [1.0, 2.0]
    assert torch.equal(x.copy(), x) and x.copy() is not x


def test_function_bridge_and_module(jt):
    class Square(jt.Function):
        def execute(self, x, k):
            self.x = x
            return x * x * k

        def grad(self, g):
            return 2 * self.x * g * 3.0, None

    x = jt.array([1.0, -2.0]).requires_grad_(True)
    y = Square()(x, 3.0)
    y.sum().backward()
    assert torch.equal(x.grad, torch.tensor([6.0, -12.0]))

    class Net(jt.nn.Module):
        def execute(self, a):
            return a + 1
    assert float(Net()(jt.array([1.0]))) == 2.0


def test_adam_matches_update_rule_and_torch(jt):
    from jittor import nn
    w0 = np.random.default_rng(0).standard_normal((5, 3)).astype(np.float32)
    p = jt.array(w0)
    opt = nn.Adam([{"params": [p], "lr": 0.01, "name": "w"}], lr=0.0, eps=1e-15)
    assert opt.param_groups[0]["params"][0] is p and p.requires_grad         # the caller's tensor became the parameter
    q = torch.tensor(w0, requires_grad=True)
    ref = torch.optim.Adam([q], lr=0.01, eps=1e-15)
    for it in range(5):
        loss = ((p - 1.0) ** 2).sum() + p[:, 0].sum()
        opt.backward(loss)
        assert torch.allclose(opt.param_groups[0]["grads"][0], 2 * (p.detach() - 1.0) + torch.tensor([1.0, 0, 0]))
        opt.step(); opt.zero_grad()
        ref.zero_grad(); (((q - 1.0) ** 2).sum() + q[:, 0].sum()).backward(); ref.step()
    assert torch.allclose(p.detach(), q.detach(), atol=1e-6)
    sd = opt.state_dict()
    opt2 = nn.Adam([{"params": [jt.zeros((5, 3))], "lr": 0.5, "name": "w"}], lr=0.0)
    opt2.load_state_dict(sd)
    assert torch.equal(opt2.param_groups[0]["params"][0].detach(), p.detach()) and opt2.param_groups[0]["lr"] == 0.01


def test_densifier_style_edits_of_param_groups(jt):
    """The operations scene/mesh_based_gaussian_model.py:411-480 performs on the optimizer."""
    from jittor import nn
    p = jt.array(np.arange(12, dtype=np.float32).reshape(4, 3))
    opt = nn.Adam([{"params": [p], "lr": 0.1, "name": "bc"}], lr=0.0, eps=1e-15)
    opt.add_param_group({"params": [jt.zeros((4, 3)) + 0], "lr": 0.0, "name": "screenspace_points"})
    opt.backward((opt.param_groups[0]["params"][0] ** 2).sum() + (opt.param_groups[1]["params"][0] * 3).sum())
    assert torch.equal(opt.param_groups[-1]["grads"][0], torch.full((4, 3), 3.0))
    opt.step(); opt.zero_grad()
    g = opt.param_groups[0]
    mask = torch.tensor([True, False, True, True])
    with jt.no_grad():                                                   # prune
        g["m"][0].update(g["m"][0][mask]); g["values"][0].update(g["values"][0][mask])
        with jt.enable_grad():
            old = g["params"].pop(); g["params"].append(old[mask]); del old
    assert g["params"][0].shape == (3, 3) and g["params"][0].is_leaf and g["params"][0].requires_grad
    ext = jt.ones((2, 3))
    with jt.no_grad():                                                   # cat
        g["m"][0] = jt.concat((g["m"][0], jt.zeros_like(ext)), dim=0)
        g["values"][0] = jt.concat((g["values"][0], jt.zeros_like(ext)), dim=0)
        old = g["params"].pop()
        with jt.enable_grad():
            g["params"].append(jt.concat((old, ext), dim=0))
    assert g["params"][0].shape == (5, 3) and g["params"][0].is_leaf and g["params"][0].requires_grad
    with jt.no_grad():                                                   # replace (reset_opacity)
        with jt.enable_grad():
            g["params"][0] = (g["params"][0] * 0.5).copy()
        g["m"][0] = jt.zeros_like(g["params"][0]); g["values"][0] = jt.zeros_like(g["params"][0])
    opt.param_groups.pop()
    opt.backward((g["params"][0] ** 2).sum()); opt.step(); opt.zero_grad()
    assert g["params"][0].shape == (5, 3) and g["m"][0].shape == (5, 3)


# ---------------------------------------------------------------------------------------------------------------
def _load_ref(name, relpath):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def ref_env(jt):
    if not os.path.isdir(REF):
        pytest.skip("reference tree not mounted")
    added = []
    for name in ("plyfile", "igl"):                                  # absent third-party imports of the model file
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.PlyData = m.PlyElement = None
            sys.modules[name] = m; added.append(name)
    if "scene" not in sys.modules:                                   # keep scene/__init__.py (dataset readers, PIL...) out
        pkg = types.ModuleType("scene"); pkg.__path__ = []
        sys.modules["scene"] = pkg; added.append("scene")
    sys.path.insert(0, REF)
    yield
    sys.path.remove(REF)
    for name in added + [k for k in list(sys.modules) if k == "utils" or k.startswith("utils.")]:
        sys.modules.pop(name, None)


def test_reference_utils_run_on_the_subset(jt, ref_env):
    from gaussianmesh_amd import scenes
    gu = _load_ref("ref_general_utils", "utils/general_utils.py")
    rng = np.random.default_rng(0)
    q = rng.standard_normal((7, 4)).astype(np.float32); s = np.exp(rng.standard_normal((7, 3))).astype(np.float32)
    L = gu.build_scaling_rotation(jt.array(s), jt.array(q))
    cov = gu.strip_symmetric(L @ L.transpose(1, 2)).numpy()
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    assert np.allclose(cov, scenes.strip_symmetric(scenes.cov3d_from_scale_rot(s, qn)), atol=1e-5)
    x = jt.array([0.1, 0.5, 0.9])
    assert torch.allclose(torch.sigmoid(gu.inverse_sigmoid(x)), x, atol=1e-6)
    f = gu.get_expon_lr_func(1e-2, 1e-4, max_steps=100)
    assert abs(f(0) - 1e-2) < 1e-12 and abs(f(100) - 1e-4) < 1e-12


def test_reference_ssim_on_the_subset_agrees_with_loss_oracle(jt, ref_env):
    from oracle import loss_oracle as lo
    lu = _load_ref("ref_loss_utils", "utils/loss_utils.py")
    rng = np.random.default_rng(1)
    a = rng.random((3, 40, 36)).astype(np.float32); b = np.clip(a + 0.1 * rng.standard_normal(a.shape), 0, 1).astype(np.float32)
    assert abs(float(lu.ssim(jt.array(a), jt.array(b))) - lo.ssim(a, b)) < 2e-6
    assert abs(float(lu.l1_loss(jt.array(a), jt.array(b))) - lo.l1(a, b)) < 1e-7
    assert np.allclose(lu.gaussian(11, 1.5).numpy(), lo.window_1d(), rtol=3e-7, atol=0)     # float32 sum order may differ by an ulp


def test_mesh_restrict_loss_matches_the_reference_function(jt, ref_env):
    from gaussianmesh_amd import loss
    lu = _load_ref("ref_loss_utils2", "utils/loss_utils.py")
    rng = np.random.default_rng(5)
    sc = np.exp(rng.standard_normal((50, 3))).astype(np.float32) * 0.3
    p1, p2, p3 = (rng.standard_normal((50, 3)).astype(np.float32) for _ in range(3))
    want = float(lu.mesh_restrict_loss(jt.array(sc), jt.array(p1), jt.array(p2), jt.array(p3), weight=0.4))
    t = lambda a: torch.tensor(a, requires_grad=True)
    ts = t(sc)
    got = loss.mesh_restrict_loss(ts, t(p1), t(p2), t(p3), weight=0.4)
    assert want > 0 and abs(float(got) - want) <= 1e-5 * want
    got.backward()
    assert ts.grad is not None and float(ts.grad.abs().sum()) > 0
    assert torch.allclose(loss.circumradius(t(p1), t(p2), t(p3)), lu.circumradius(jt.array(p1), jt.array(p2), jt.array(p3)), rtol=1e-6)


def test_reference_mesh_model_trains_and_densifies_on_the_subset(jt, ref_env):
    from gaussianmesh_amd import scenes
    mm = _load_ref("ref_mesh_model", "scene/mesh_based_gaussian_model.py")
    verts, faces = scenes.torus_mesh(6, 4)
    n = faces.shape[0]
    g = mm.MeshBasedGaussianModel(3)
    v = verts.astype(np.float32)
    g.vertex1, g.vertex2, g.vertex3 = (jt.array(v[faces[:, k]]).stop_grad() for k in range(3))
    e = lambda a, b: jt.unsqueeze(jt.norm(a - b, dim=1), 1)
    g.r = (e(g.vertex1, g.vertex2) + e(g.vertex2, g.vertex3) + e(g.vertex3, g.vertex1)) / 3
    nrm = np.cross(v[faces[:, 1]] - v[faces[:, 0]], v[faces[:, 2]] - v[faces[:, 0]])
    g.normal = jt.array(nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).stop_grad()
    g.fid = jt.unsqueeze(jt.arange(n), 1).stop_grad()
    g.vertex_index = jt.array(faces).stop_grad()
    g.v = jt.array(v).stop_grad()
    g._bc = jt.ones((n, 3)) / 3
    g._distance = jt.zeros((n, 1))
    g._features_dc = jt.zeros((n, 1, 3)); g._features_rest = jt.zeros((n, 15, 3))
    g._scaling = jt.log(jt.ones((n, 3)) * 0.05); g._rotation = jt.zeros((n, 4)); g._rotation[:, 0] = 1
    g._opacity = mm.inverse_sigmoid(0.1 * jt.ones((n, 1)))
    g.max_radii2D = jt.zeros((n,))
    args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                                 position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001)
    g.spatial_lr_scale = 1.0
    g.training_setup(args)
    g.reset_viewspace_point()
    # the model's own position formula against this package's renderer glue (scene/mesh_based_gaussian_model.py:138-152)
    xyz = g.get_xyz
    centroid = (v[faces[:, 0]] + v[faces[:, 1]] + v[faces[:, 2]]) / 3
    assert np.allclose(xyz.numpy(), centroid, atol=1e-6)
    target = jt.array(centroid + 0.01)
    loss = ((g.get_xyz - target) ** 2).sum() + (g.get_opacity ** 2).sum() + (g.screenspace_points * 2.0).sum()
    g.optimizer.backward(loss)
    assert torch.equal(g.get_viewspace_point_grad(), torch.full((n, 3), 2.0))
    assert float(g.optimizer.param_groups[1]["grads"][0].abs().sum()) > 0          # distance gets gradient through the offset
    g.update_learning_rate(1)
    g.optimizer.step(); g.optimizer.zero_grad()
    # prune + reset opacity through the reference's own optimizer surgery
    mask = jt.zeros((n,), dtype=jt.bool); mask[:5] = True
    with jt.no_grad():
        g.prune_points(mask)
        g.reset_opacity()
    assert g.get_number == n - 5 and g._bc.requires_grad and g._bc.is_leaf and g.get_xyz.shape == (n - 5, 3)
    g.reset_viewspace_point()
    g.optimizer.backward((g.get_xyz ** 2).sum() + (g.get_opacity).sum())
    g.optimizer.step(); g.optimizer.zero_grad()
    st = g.capture()
    assert st[1].shape == (n - 5, 3)
    # face-splitting densification (scene/mesh_based_gaussian_model.py:504-569): 3 selected faces -> 4 children each
    m = g.get_number
    with jt.no_grad():
        g.add_densification_stats(jt.ones((m, 3)), jt.ones((m,), dtype=jt.bool))
        g.bc_gradient_accum[3:] = 0.0
        g.densify_and_prune(0.5, 0.005, 1.0, None, 4)
    assert g.get_number == m - 3 + 12 and g.vertex1.shape == (m + 9, 3) and g._bc.is_leaf and g._bc.requires_grad
    assert g.optimizer.param_groups[0]["m"][0].shape == (m + 9, 3)
    g.reset_viewspace_point()
    g.optimizer.backward((g.get_xyz ** 2).sum()); g.optimizer.step(); g.optimizer.zero_grad()


def test_uninstall_restores_torch_tensor():
    import torch
    import gaussianmesh_amd.compat as compat
    compat.install(force=True)
    t = torch.ones(2, requires_grad=True)
    assert t.numpy().tolist() == [1.0, 1.0] and hasattr(torch.Tensor, "stop_grad")          # Jittor semantics while installed
    compat.uninstall()
    assert not hasattr(torch.Tensor, "stop_grad")
    with pytest.raises(RuntimeError):
        t.numpy()                                                                          # torch's own behaviour is back
    compat.install(force=True)                                                             # (other tests expect the shim)


def test_igl_subset_and_edit_py_import_block(tmp_path):
    """compat.install(edit_tool=True): `import igl` (edit.py:7; libigl is not installed in this image) resolves to the subset with
    the four calls the in-scope reference code makes, next to `from edittool import ...` and `from render_origin import ...`; where the
    reference tree is mounted, every import statement at the top of its edit.py is executed as it stands."""
    import ast
    import importlib.util
    placeholder = sys.modules.get("igl")                         # (ref_env above parks an empty module under this name)
    if placeholder is not None and getattr(placeholder, "__file__", None) is None:
        del sys.modules["igl"]
    real_igl = "igl" not in sys.modules and importlib.util.find_spec("igl") is not None
    compat.install(edit_tool=True)
    try:
        _check_igl_and_imports(tmp_path, real_igl)
    finally:
        if placeholder is not None and getattr(placeholder, "__file__", None) is None:
            sys.modules["igl"] = placeholder


def _check_igl_and_imports(tmp_path, real_igl):
    import ast
    ns = {}
    exec("import igl\nfrom edittool import ObjectVisualTool, SceneVisualTool\nfrom render_origin import save_image", ns)
    igl = ns["igl"]
    assert callable(ns["save_image"]) and ns["ObjectVisualTool"].__module__.startswith("gaussianmesh_amd")
    if os.path.isfile(os.path.join(REF, "edit.py")):
        tree = ast.parse(open(os.path.join(REF, "edit.py")).read())
        imports = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
        assert any(isinstance(n, ast.Import) and n.names[0].name == "igl" for n in imports)
        exec(compile(ast.Module(body=imports, type_ignores=[]), "edit.py imports", "exec"), {})
    if real_igl:
        return                                                   # a real libigl answers; nothing of ours to check
    assert getattr(igl, "__gaussianmesh_compat__", False)
    # unit square of two triangles in z = 0, one degenerate face
    V = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float64)
    F = np.array([[0, 1, 2], [0, 2, 3], [1, 1, 2]], np.int32)
    n = igl.per_face_normals(V, F, np.array([1.0, 0.0, 0.0]))
    assert np.allclose(n, [[0, 0, 1], [0, 0, 1], [1, 0, 0]])      # the fallback row for the degenerate face (mesh_based_gaussian_model.py:193)
    P = np.array([[0.75, 0.25, 2.0], [0.25, 0.75, -1.0], [2.0, 0.5, 0.0], [-1.0, -1.0, 0.0], [0.5, 0.5, 0.5]])
    sqr, idx, close = igl.point_mesh_squared_distance(P, V, F[:2])
    assert np.allclose(sqr, [4.0, 1.0, 1.0, 2.0, 0.25]) and idx.tolist()[:2] == [0, 1]
    assert np.allclose(close, [[0.75, 0.25, 0], [0.25, 0.75, 0], [1, 0.5, 0], [0, 0, 0], [0.5, 0.5, 0]])
    # brute force over a dense sampling of a random mesh
    rng = np.random.default_rng(0)
    Vr = rng.normal(size=(12, 3)); Fr = rng.integers(0, 12, size=(20, 3)).astype(np.int32)
    Fr = Fr[(Fr[:, 0] != Fr[:, 1]) & (Fr[:, 1] != Fr[:, 2]) & (Fr[:, 0] != Fr[:, 2])]
    Pr = rng.normal(size=(40, 3))
    sqr, idx, close = igl.point_mesh_squared_distance(Pr, Vr, Fr)
    u = np.linspace(0, 1, 60); uu, vv = np.meshgrid(u, u); keep = uu + vv <= 1
    bary = np.stack([1 - uu[keep] - vv[keep], uu[keep], vv[keep]], 1)                      # [S,3]
    samples = np.einsum("sk,fkc->fsc", bary, Vr[Fr])                                      # [F,S,3]
    d2 = ((Pr[:, None, None, :] - samples[None]) ** 2).sum(-1).min(axis=2)                 # [P,F]
    assert (sqr <= d2.min(axis=1) + 1e-12).all() and np.allclose(sqr, d2.min(axis=1), atol=5e-2)
    assert np.allclose(((Pr - close) ** 2).sum(1), sqr)
    # files: OBJ and OFF round trips, polygon fan triangulation
    for ext in (".obj", ".off"):
        path = str(tmp_path / ("m" + ext))
        assert igl.write_triangle_mesh(path, Vr, Fr)
        v2, f2 = igl.read_triangle_mesh(path)
        assert v2.dtype == np.float64 and np.array_equal(v2, Vr) and np.array_equal(f2, Fr)
    with pytest.raises(AttributeError):
        igl.cotmatrix
    with pytest.raises(ValueError):
        igl.read_triangle_mesh(str(tmp_path / "m.stl"))
