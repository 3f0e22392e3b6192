"""CPU: host-side pieces of the training harness."""
import numpy as np


def test_lr_schedule_matches_the_reference_formula():
    from gaussianmesh_amd.train import get_expon_lr_func
    f = get_expon_lr_func(1.6e-4, 1.6e-6, lr_delay_mult=0.01, max_steps=30000)
    assert abs(f(0) - 1.6e-4) < 1e-12 and abs(f(30000) - 1.6e-6) < 1e-15
    assert abs(f(15000) - np.sqrt(1.6e-4 * 1.6e-6)) < 1e-12            # log-linear midpoint
    g = get_expon_lr_func(1e-2, 1e-4, lr_delay_steps=100, lr_delay_mult=0.1, max_steps=1000)
    assert abs(g(0) - 1e-3) < 1e-12 and g(-1) == 0.0


def test_schedule_and_defaults_match_values_computed_by_the_reference():
    """tests/golden/schedule.npz: utils/general_utils.py get_expon_lr_func and arguments/__init__.py OptimizationParams
    executed in the dev container (tests/golden/make_golden.py)."""
    import os
    from gaussianmesh_amd.train import DEFAULT_OPT, get_expon_lr_func
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "schedule.npz"))
    for (a, b, d, m, n), ref in zip(f["cases"], f["lr"]):
        fn = get_expon_lr_func(a, b, lr_delay_steps=int(d), lr_delay_mult=m, max_steps=int(n))
        got = np.array([fn(int(st)) for st in f["steps"]])
        assert np.abs(got - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-30)
    ref_opt = dict(zip([str(k) for k in f["opt_names"]], f["opt_values"]))
    for k, v in DEFAULT_OPT.items():
        assert float(v) == ref_opt[k], (k, v, ref_opt[k])


def test_iteration_schedule_is_the_reference_loops():
    """Trainer.schedule: which iterations of train_mesh_gaussian.py:66-148 raise the SH degree (:70-71), keep the densification
    statistics (:119-124), run densify_and_prune (:126-128), skip the optimizer step (:137-139 update_flag), reset the opacity
    (:129-130) - with the defaults of arguments/__init__.py:70-93 (pinned by tests/golden/schedule.npz above)."""
    from types import SimpleNamespace
    from gaussianmesh_amd.train import DEFAULT_OPT, Trainer
    tr = Trainer.__new__(Trainer)                                 # schedule() reads self.opt only
    tr.opt = SimpleNamespace(**DEFAULT_OPT)
    plans = {it: tr.schedule(it) for it in range(1, 30001)}
    assert [it for it, p in plans.items() if p["oneup"]] == list(range(1000, 30001, 1000))
    assert [it for it, p in plans.items() if p["densify"]] == list(range(600, 15000, 200))        # > 500, every 200, < 15000
    assert all(p["stats"] == (it < 15000) for it, p in plans.items())
    assert [it for it, p in plans.items() if p["reset_opacity"]] == [3000, 6000, 9000, 12000]
    assert [it for it, p in plans.items() if not p["optimizer_step"]] == list(range(600, 15000, 200)) + [30000]
    assert plans[600]["size_threshold"] is None and plans[3000]["size_threshold"] is None and plans[3200]["size_threshold"] == 20
    white = [it for it in range(1, 3001) if tr.schedule(it, white_background=True)["reset_opacity"]]
    assert white == [500, 3000]
    # the first 1000 iterations (BASELINE config C5): densify at 600, 800 - and 1000, where the degree goes from 0 to 1 first
    assert [it for it in range(1, 1001) if plans[it]["densify"]] == [600, 800, 1000] and plans[1000]["oneup"]


def test_two_trainers_do_not_share_sync_free_state():
    """rasterizer.SyncFreeState is per owner and current per thread: a trainer's capacity guess and unverified forwards are
    invisible to another trainer, to code outside step(), and to another thread."""
    import threading
    from gaussianmesh_amd import rasterizer as Rz
    assert Rz.current_sync_free() is None
    a, b = Rz.SyncFreeState(), Rz.SyncFreeState()
    seen = []
    with a:
        assert Rz.current_sync_free() is a
        t = threading.Thread(target=lambda: seen.append(Rz.current_sync_free()))
        t.start(); t.join()
        with b:
            assert Rz.current_sync_free() is b
        assert Rz.current_sync_free() is a
    assert Rz.current_sync_free() is None and seen == [None]
    a.note_count("dev", 1000)
    assert b.capacity == {} and a.capacity["dev"] == int(1000 * a.growth) + 4096
    assert not hasattr(Rz, "_sync_free")


def _tiny_model(N=12, seed=0):
    import torch
    from gaussianmesh_amd.renderer import MeshBoundGaussians
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return MeshBoundGaussians(r(N, 3), r(N, 1), r(N, 1, 3), r(N, 15, 3), r(N, 3), r(N, 4), r(N, 1), r(N, 3), r(N, 3), r(N, 3), r(N, 3),
                              torch.rand(N, 1, generator=g), sh_degree=3)


def test_dense_coefficient_zero_leaf_is_not_a_registered_parameter():
    """The dense [N,1,3] leaf of begin_dense_dc() lives beside the model's parameters, not among them: parameters() and state_dict() keep
    the keys of a fresh model, the serialised rows carry the CURRENT coefficient 0 (the live rows are stale until the fold), and a strict
    load into a fresh model succeeds."""
    import torch
    m, fresh = _tiny_model(), _tiny_model(seed=1)
    m.active_sh_degree = 0
    keys = set(m.state_dict().keys())
    names = {n for n, _ in m.named_parameters()}
    leaf = m.begin_dense_dc()
    assert set(m.state_dict().keys()) == keys == set(fresh.state_dict().keys())
    assert {n for n, _ in m.named_parameters()} == names and all(p is not leaf for p in m.parameters())
    with torch.no_grad():
        leaf.add_(1.0)                                            # "training" the leaf: the rows' coefficient 0 is now stale
    sd = m.state_dict()
    assert torch.equal(sd["_features"][:, :1], leaf.detach()) and not torch.equal(m._features.detach()[:, :1], leaf.detach())
    assert torch.equal(sd["_features"][:, 1:], m._features.detach()[:, 1:])
    fresh.load_state_dict(sd, strict=True)
    assert torch.equal(fresh.get_features.detach(), m.get_features.detach())
    m.end_dense_dc()
    assert m._features_dc0 is None and torch.equal(m._features.detach()[:, :1], leaf.detach())


def test_a_rows_trainer_folds_a_model_left_in_dense_mode():
    """Trainer(dense_dc=False) on a model another trainer left in dense mode folds the leaf first and steps the ROWS (its optimizer
    would otherwise hold a tensor the model drops at the next oneupSHdegree()); train_iteration only asks for densification statistics
    when the schedule reaches a densify iteration."""
    import pytest
    import torch
    from gaussianmesh_amd.train import Trainer
    m = _tiny_model()
    m.active_sh_degree = 0
    leaf = m.begin_dense_dc()
    with torch.no_grad():
        leaf.mul_(2.0)
    tr = Trainer(m, dense_dc=False)
    assert m._features_dc0 is None and torch.equal(m._features.detach()[:, :1], leaf.detach())
    grp = next(g for g in tr.optimizer.param_groups if g["name"] == "f_dc+f_rest")
    assert grp["params"][0] is m._features and grp["period"] == 48
    tr.iteration = 599                                            # iteration 600 densifies: refused without statistics, BEFORE anything runs
    with pytest.raises(ValueError, match="densify_stats=True"):
        tr.train_iteration(None, None, None)
    assert tr.iteration == 599
