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


def account_outlier_pixels(fw, color, W, H, tol=1e-4):
    """Flip accounting for the forward gate (north_star: <= 1e-4 max-abs per pixel).

    `fw` is the oracle's forward (oracle.forward_full), `color` the image under test.  Two correct float evaluations of
    the blend can only differ by more than rounding on a pixel where a DISCRETE decision of RAST/forward.cu:336-352
    (`power > 0`, `alpha < 1/255`, `T (1 - alpha) < 1e-4`) sits within rounding distance of its threshold for one of the
    entries the pixel visits.  For every pixel whose colour differs from the oracle's by more than `tol`, the pixel's
    list is re-walked in float64 and such an entry must exist; the rounding distance of an entry is derived from its own
    cancellation (the quadratic form's terms), not from a blanket tolerance.  Returns
    (n_outliers, n_unexplained, worst_error_among_explained)."""
    geo, bins = fw["geo"], fw["bins"]
    err = np.abs(np.asarray(color, np.float64) - fw["color"]).max(axis=0)           # [H, W]
    ys, xs = np.nonzero(err > tol)
    gx = (W + 15) // 16
    eps = 2.0 ** -23
    xy = geo["xy"].astype(np.float64); co = geo["conic_op"].astype(np.float64)
    unexplained, worst = 0, 0.0
    for y, x in zip(ys, xs):
        t = (y // 16) * gx + (x // 16)
        r0, r1 = bins["ranges"][t]
        g = bins["point_list"][r0:r1]
        dx = xy[g, 0] - x; dy = xy[g, 1] - y
        t1 = 0.5 * co[g, 0] * dx * dx; t2 = 0.5 * co[g, 2] * dy * dy; t3 = co[g, 1] * dx * dy
        power = -(t1 + t2) - t3
        mag = np.abs(t1) + np.abs(t2) + np.abs(t3)
        d_pow = 8 * eps * (mag + 1.0) + 2e-6           # rounding distance of `power` (and of log alpha) for this entry
        op = co[g, 3]
        with np.errstate(over="ignore"):
            alpha = np.minimum(0.99, op * np.exp(np.minimum(power, 50.0)))
        acc = (power <= 0) & (alpha >= 1.0 / 255.0)
        keep = np.where(acc, 1.0 - alpha, 1.0)
        T_before = np.concatenate([[1.0], np.cumprod(keep)[:-1]])
        test_T = T_before * (1.0 - alpha)
        definite_stop = acc & (test_T < 1e-4 * (1 - 1e-3))
        n_vis = int(np.argmax(definite_stop)) + 1 if definite_stop.any() else len(g)
        v = slice(0, n_vis)
        amb_alpha = np.abs(np.log(np.maximum(alpha[v], 1e-300) * 255.0)) <= d_pow[v]
        amb_power = (np.abs(power[v]) <= d_pow[v]) & (op[v] >= 1.0 / 255.0 * 0.99)
        amb_stop = acc[v] & (np.abs(np.log(np.maximum(test_T[v], 1e-300) / 1e-4)) <= 1e-3)
        if (amb_alpha | amb_power | amb_stop).any():
            worst = max(worst, float(err[y, x]))
        else:
            unexplained += 1
    return len(ys), unexplained, worst


def assert_forward_gate(fw, color, W, H, tol=1e-4, what=""):
    """The strict forward gate: every pixel within `tol` of the oracle, except pixels with a provable threshold flip
    (account_outlier_pixels), which are bounded by one flipped entry's weight: alpha * T * |colour| <= 2 / 255."""
    n_out, n_bad, worst = account_outlier_pixels(fw, color, W, H, tol)
    assert n_bad == 0, "%s: %d of %d outlier pixels (> %g) have no entry at a decision threshold" % (what, n_bad, n_out, tol)
    assert n_out <= max(2, 1e-4 * W * H), "%s: %d outlier pixels" % (what, n_out)
    assert worst <= 2.0 / 255.0 + 1e-3, "%s: explained outlier of %g" % (what, worst)
    return n_out
