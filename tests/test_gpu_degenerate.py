"""Degenerate inputs against the oracle (round 5): the inputs the reference's kernels meet at the start of a training run and in
edited scenes - zero scales (the projected covariance is the 0.3-px low-pass filter alone), opacities of exactly 0 and 1, splats
far larger than the image (rectangle = every tile), a 1x1 image, an image of exactly one tile, one pixel row - forward through the
strict gate, radii bit-identical, gradients through the 1e-3 gate and finite."""
import numpy as np
import pytest
import torch

from helpers import small_scene, assert_forward_gate
from test_gpu_parity import _grads_gpu, _rel

pytestmark = pytest.mark.gpu


def _run(oracle, sc, cam, D=3, what=""):
    bg = np.array([0.3, 0.1, 0.8], np.float32)
    W, H = cam["W"], cam["H"]
    dpix = np.random.default_rng(3).normal(size=(3, H, W)).astype(np.float32)
    fw = oracle.forward_full(sc, cam, bg, D=D)
    bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=D)
    color, radii, g = _grads_gpu(sc, cam, bg, dpix, D, False, False)
    assert np.array_equal(radii, fw["geo"]["radii"]), what
    assert np.isfinite(color).all(), what
    assert_forward_gate(fw, color, W, H, 1e-4, what)
    for k, ref in (("means", bw["dmean3D"]), ("opac", bw["dopacity"]), ("shs", bw["dsh"]), ("scales", bw["dscale"]), ("rots", bw["drot"])):
        got = np.asarray(g[k]).reshape(np.asarray(ref).shape)
        assert np.isfinite(got).all(), (what, k)
        if np.abs(ref).max() > 0:
            assert _rel(got, ref) <= 1e-3, (what, k, _rel(got, ref))
        else:
            assert np.abs(got).max() == 0, (what, k)
    return fw, color


def test_zero_scales_and_extreme_opacities(oracle):
    sc, cam = small_scene(P=400, W=64, H=48, seed=2, behind=False)
    sc["scales"][:100] = 0.0                        # points: cov2D = diag(0.3, 0.3)
    sc["scales"][100:150, 1:] = 0.0                 # needles of zero width
    sc["opac"][150:200] = 0.0                       # never accepted (alpha < 1/255 everywhere)
    sc["opac"][200:250] = 1.0                       # alpha clamps at 0.99
    fw, _ = _run(oracle, sc, cam, what="zero scales / opacity 0 and 1")
    assert (fw["geo"]["radii"][:100] > 0).any()


def test_splats_larger_than_the_image(oracle):
    sc, cam = small_scene(P=60, W=150, H=90, seed=4, behind=False, scale_lo=3.0, scale_hi=40.0)
    fw, _ = _run(oracle, sc, cam, what="huge splats")
    assert fw["geo"]["radii"].max() > 2000           # rectangles clipped to the whole 10 x 6 tile grid


@pytest.mark.parametrize("W,H", [(1, 1), (16, 16), (40, 1), (1, 33), (17, 15)])
def test_tiny_and_ragged_images(oracle, W, H):
    sc, cam = small_scene(P=120, W=W, H=H, seed=6, behind=False, scale_lo=0.1, scale_hi=0.8)
    _run(oracle, sc, cam, D=1, what="%dx%d" % (W, H))
