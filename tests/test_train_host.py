"""CPU: host-side pieces of the training harness."""
import numpy as np


def test_lr_schedule_matches_the_reference_formula():
    from gaussianmesh_amd.train import get_expon_lr_func
    f = get_expon_lr_func(1.6e-4, 1.6e-6, lr_delay_mult=0.01, max_steps=30000)
    assert abs(f(0) - 1.6e-4) < 1e-12 and abs(f(30000) - 1.6e-6) < 1e-15
    assert abs(f(15000) - np.sqrt(1.6e-4 * 1.6e-6)) < 1e-12            # log-linear midpoint
    g = get_expon_lr_func(1e-2, 1e-4, lr_delay_steps=100, lr_delay_mult=0.1, max_steps=1000)
    assert abs(g(0) - 1e-3) < 1e-12 and g(-1) == 0.0
