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
