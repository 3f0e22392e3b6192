"""CPU: topology changes of the training loop (8f-3) against fixtures produced by EXECUTING the reference's own model code
(scene/mesh_based_gaussian_model.py:334-339, 411-563, 596-647 through tests/golden/make_golden_model.py): after
Trainer.densify_and_prune / prune_points / reset_opacity / densify_and_split_for_init every parameter row, BOTH Adam moments,
every per-face buffer and the densification statistics equal the reference's, bit for bit (the edits only move and copy rows;
the one formula, log(exp(s) / 3.2), is the same two float32 operations)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "densify.npz")
GROUPS = ("bc", "distance", "opacity", "scaling", "rotation")


def _trainer(fx, tag):
    from gaussianmesh_amd.renderer import MeshBoundGaussians
    from gaussianmesh_amd.train import Trainer
    t = lambda k: torch.tensor(fx["%s_%s" % (tag, k)])
    g = MeshBoundGaussians(t("p_bc"), t("p_distance"), t("p_f_dc"), t("p_f_rest"), t("p_scaling"), t("p_rotation"), t("p_opacity"),
                           t("b_vertex1"), t("b_vertex2"), t("b_vertex3"), t("b_normal"), t("b_r"), fid=t("b_fid"),
                           vertex_index=t("b_vertex_index"), v=t("b_v"))
    tr = Trainer(g, densify_stats=True)
    for grp in tr.optimizer.param_groups:
        n = grp["name"]
        if n == "f_dc+f_rest":
            grp["m"][0] = torch.cat((t("m_f_dc"), t("m_f_rest")), dim=1)
            grp["values"][0] = torch.cat((t("v_f_dc"), t("v_f_rest")), dim=1)
        else:
            grp["m"][0], grp["values"][0] = t("m_" + n), t("v_" + n)
    tr.max_radii2D, tr.bc_gradient_accum, tr.denom = t("b_max_radii2D"), t("b_bc_gradient_accum"), t("b_denom")
    return tr


def _assert_state(tr, fx, tag):
    e = lambda k: fx["%s_%s" % (tag, k)]
    eq = lambda a, b, what: (a.shape == b.shape and np.array_equal(a, b)) or pytest.fail("%s %s differs (shapes %s %s)" % (tag, what, a.shape, b.shape))
    n = lambda x: x.detach().numpy()
    for grp in tr.optimizer.param_groups:
        name = grp["name"]
        if name == "f_dc+f_rest":
            for key, src in (("p", grp["params"][0]), ("m", grp["m"][0]), ("v", grp["values"][0])):
                eq(n(src[:, :1]), e(key + "_f_dc"), key + "_f_dc")
                eq(n(src[:, 1:]), e(key + "_f_rest"), key + "_f_rest")
        else:
            eq(n(grp["params"][0]), e("p_" + name), "p_" + name)
            eq(n(grp["m"][0]), e("m_" + name), "m_" + name)
            eq(n(grp["values"][0]), e("v_" + name), "v_" + name)
        assert grp["params"][0].is_leaf and grp["params"][0].requires_grad
    g = tr.g
    # the model attributes ARE the optimizer's tensors
    for name, attr in (("bc", "_bc"), ("distance", "_distance"), ("f_dc+f_rest", "_features"), ("opacity", "_opacity"), ("scaling", "_scaling"),
                       ("rotation", "_rotation")):
        assert getattr(g, attr) is [q for q in tr.optimizer.param_groups if q["name"] == name][0]["params"][0], name
    for b in ("vertex1", "vertex2", "vertex3", "normal", "r", "fid", "vertex_index", "v"):
        eq(n(getattr(g, b)), e("b_" + b).astype(n(getattr(g, b)).dtype), b)
    eq(n(tr.max_radii2D), e("b_max_radii2D"), "max_radii2D")
    eq(n(tr.bc_gradient_accum), e("b_bc_gradient_accum"), "bc_gradient_accum")
    eq(n(tr.denom), e("b_denom"), "denom")
    assert g.screenspace_points.shape == (g.get_number, 3) and g.screenspace_points.requires_grad and g.screenspace_points.is_leaf


@pytest.fixture(scope="module")
def fx():
    return np.load(GOLD)


def test_fixture_initial_state_round_trips(fx):
    _assert_state(_trainer(fx, "A0"), fx, "A0")


def test_densify_and_prune_split_into_four(fx):
    tr = _trainer(fx, "A0")
    n0 = tr.g.get_number
    n1 = tr.densify_and_prune(float(fx["A_threshold"]), 0.005, 1.0, None, 4)
    assert n1 == fx["A1_p_bc"].shape[0] and n1 > n0 and tr.resizes == 1
    _assert_state(tr, fx, "A1")


def test_densify_and_prune_split_into_five(fx):
    tr = _trainer(fx, "B0")
    tr.densify_and_prune(float(fx["B_threshold"]), 0.005, 1.0, None, 5)
    _assert_state(tr, fx, "B1")


def test_prune_points_then_reset_opacity(fx):
    tr = _trainer(fx, "C0")
    tr.prune_points(torch.tensor(fx["C_mask"]))
    _assert_state(tr, fx, "C1")
    tr.reset_opacity()
    _assert_state(tr, fx, "C2")


def test_densify_and_split_for_init(fx):
    tr = _trainer(fx, "D0")
    n0 = tr.g.get_number
    assert tr.densify_and_split_for_init() == 4 * n0
    _assert_state(tr, fx, "D1")


def test_resize_argument_checks_and_noop_selection(fx):
    tr = _trainer(fx, "A0")
    n0 = tr.g.get_number
    assert tr.densify_and_split(torch.zeros(n0, dtype=torch.bool)) == n0 and tr.resizes == 0          # nothing selected: nothing happens (:520-521)
    with pytest.raises(ValueError):
        tr.resize(keep_mask=torch.ones(n0 + 1, dtype=torch.bool))
    with pytest.raises(ValueError):
        tr.resize(new_rows={"bc": torch.zeros(2, 3)})                                                   # appended rows need their face buffers
    with pytest.raises(ValueError):
        tr.densify_and_split(torch.ones(n0, dtype=torch.bool), N=3)
