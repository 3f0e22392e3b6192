"""CPU tests of tests/helpers.account_outlier_pixels (the strict forward gate's flip accounting)."""
import numpy as np

from helpers import account_outlier_pixels, assert_forward_gate


def _blend(xy, co, rgb, order, W, H, drop=None):
    """float64 blend of one 16x16 tile's list with the reference's three decisions; `drop` = (pixel, entry) to skip."""
    img = np.zeros((3, H, W))
    for y in range(H):
        for x in range(W):
            T, C = 1.0, np.zeros(3)
            for e, g in enumerate(order):
                dx, dy = xy[g, 0] - x, xy[g, 1] - y
                power = -0.5 * (co[g, 0] * dx * dx + co[g, 2] * dy * dy) - co[g, 1] * dx * dy
                if power > 0:
                    continue
                alpha = min(0.99, co[g, 3] * np.exp(power))
                if alpha < 1.0 / 255.0 or (drop is not None and drop == ((y, x), e)):
                    continue
                if T * (1 - alpha) < 1e-4:
                    break
                C += rgb[g] * alpha * T
                T *= 1 - alpha
            img[:, y, x] = C
    return img


def _tile_scene():
    rng = np.random.default_rng(0)
    n = 12
    xy = rng.uniform(0, 16, (n, 2)).astype(np.float32)
    co = np.stack([rng.uniform(0.02, 0.2, n), rng.uniform(-0.01, 0.01, n), rng.uniform(0.02, 0.2, n), rng.uniform(0.2, 0.9, n)], 1).astype(np.float32)
    rgb = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    # entry 5 sits exactly on pixel (7, 9) with opacity exactly 1/255: alpha == threshold there
    xy[5] = (9.0, 7.0)
    co[5, 3] = np.float32(1.0 / 255.0)
    order = np.arange(n, dtype=np.uint32)
    fw = dict(geo=dict(xy=xy, conic_op=co, rgb=rgb), bins=dict(point_list=order, ranges=np.array([[0, n]], np.uint32)))
    return fw, xy.astype(np.float64), co.astype(np.float64), rgb.astype(np.float64), order


def test_identical_image_has_no_outliers():
    fw, xy, co, rgb, order = _tile_scene()
    fw["color"] = _blend(xy, co, rgb, order, 16, 16)
    assert account_outlier_pixels(fw, fw["color"], 16, 16) == (0, 0, 0.0)
    assert assert_forward_gate(fw, fw["color"], 16, 16) == 0


def test_threshold_flip_is_explained_and_a_plain_error_is_not():
    fw, xy, co, rgb, order = _tile_scene()
    fw["color"] = _blend(xy, co, rgb, order, 16, 16)
    # the other implementation rejects entry 5 at pixel (7, 9), where its alpha equals 1/255 to the last bit
    flipped = _blend(xy, co, rgb, order, 16, 16, drop=((7, 9), 5))
    d = np.abs(flipped - fw["color"]).max()
    assert 1e-4 < d < 2.0 / 255.0
    n_out, n_bad, worst = account_outlier_pixels(fw, flipped, 16, 16)
    assert (n_out, n_bad) == (1, 0) and abs(worst - d) < 1e-12
    # the same size of error on a pixel with no entry near a threshold is NOT excused
    wrong = fw["color"].copy()
    wrong[1, 2, 3] += d
    n_out, n_bad, _ = account_outlier_pixels(fw, wrong, 16, 16)
    assert (n_out, n_bad) == (1, 1)
    try:
        assert_forward_gate(fw, wrong, 16, 16)
    except AssertionError:
        pass
    else:
        raise AssertionError("gate accepted an unexplained outlier")
