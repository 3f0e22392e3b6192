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


def fuzz_scene(seed):
    """The random scene of tools/fuzz_oracle_parity.py / tests/test_gpu_fuzz_parity.py for a seed: 50-3000 Gaussians, 17-160 x 17-120 pixels,
    random orbit camera / background / SH degree / input mode; EVERY THIRD SEED stretches the first axis of every splat ten-fold
    (100:1 needles: the ill-conditioned regime of DESIGN.md section 2).  Returns (scene, cam, bg, D, pre_cov, pre_col, dL_dpix)."""
    rng = np.random.default_rng(100 + seed)
    P = int(rng.integers(50, 3000))
    lo = float(10 ** rng.uniform(-2.3, -1)); hi = lo * float(10 ** rng.uniform(0.3, 1.6))
    sc = scenes.make_cloud(P, seed=seed, scale_lo=lo, scale_hi=hi)
    if seed % 3 == 0:
        sc["scales"][:, 0] *= 10.0
    W = int(rng.integers(17, 160)); H = int(rng.integers(17, 120))
    cam = scenes.orbit_camera(int(rng.integers(0, 16)), 16, W, H, radius=float(rng.uniform(2.0, 9.0)))
    bg = rng.random(3).astype(np.float32)
    D = int(rng.integers(0, 4))
    pre_cov, pre_col = bool(seed % 2), bool((seed // 2) % 2)
    if pre_cov:
        sc["cov3D_precomp"] = scenes.strip_symmetric(scenes.cov3d_from_scale_rot(sc["scales"], sc["rots"])).astype(np.float32)
    if pre_col:
        sc["colors_precomp"] = rng.random((P, 3)).astype(np.float32)
    dpix = rng.normal(size=(3, H, W)).astype(np.float32)
    return sc, cam, bg, D, pre_cov, pre_col, dpix


def account_outlier_pixels(fw, color, W, H, tol=1e-4, mask=None, bg=None, detail=None):
    """Flip accounting for the forward gate (north_star: <= 1e-4 max-abs per pixel).

    `fw` is the oracle's forward (oracle.forward_full), `color` the image under test.  Two correct float evaluations of
    the blend can only differ by more than rounding on a pixel where a DISCRETE decision of RAST/forward.cu:336-352
    (`power > 0`, `alpha < 1/255`, `T (1 - alpha) < 1e-4`) sits within rounding distance of its threshold for one of the
    entries the pixel visits.  For every pixel whose colour differs from the oracle's by more than `tol`, the pixel's
    list is re-walked in float64 and such an entry must exist; the rounding distance of an entry is derived from its own
    cancellation (the quadratic form's terms), not from a blanket tolerance.  Returns
    (n_outliers, n_unexplained, worst_error_among_explained).
    mask ([H, W] bool, optional): examine THESE pixels instead of the ones whose colour is off by more than tol - e.g. the pixels whose
    contributor count differs from the oracle's: a different last contributor is a flipped decision too.
    bg + detail (a list): for every examined pixel WITHOUT such an entry the float64 walk also yields (round 6)
      * the pixel's colour in exact arithmetic on the float32 inputs (no decision is ambiguous there, so float64 takes the decisions
        every float32 evaluation takes), and
      * a first-order ROUNDING BOUND of the blend at that pixel as a function of the conics it visits: an entry's exponent is a sum of
        cancelling terms of magnitude mag = |con.x| dx^2 / 2 + |con.z| dy^2 / 2 + |con.y dx dy|; a float32 evaluation of it is off by at
        most d = 8 ulp (mag + 1) (the oracle's own statement, forward.cu:330-336) resp. 8 ulp (mag' + 1) for the matrix-core polynomial
        of gm_render.hip, whose coefficients are formed about a point up to 3.5 pixels away (mag' = mag at |dx| + 3.5, |dy| + 3.5), and
        the image moves by  sum_i alpha_i (d_i + d'_i) |T_i c_i - S_i / (1 - alpha_i)|  (S_i: everything blended behind entry i incl.
        the background - the derivative the backward pass uses, backward.cu:470-500); entries clamped at 0.99 do not move.
      detail gets (y, x, error vs the oracle, error of `color` vs float64, error of the oracle vs float64, bound) per such pixel."""
    geo, bins = fw["geo"], fw["bins"]
    err = np.abs(np.asarray(color, np.float64) - fw["color"]).max(axis=0)           # [H, W]
    ys, xs = np.nonzero(err > tol if mask is None else np.asarray(mask, bool).reshape(H, W))
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
        # rounding distance of `power` (and of log alpha) for this entry: eight float32 roundings of its cancelling terms
        # (pre-multiplied conic, two differences, three products, two sums) plus eight for exp and the opacity product -
        # no flat allowance on top
        d_pow = 8 * eps * (mag + 1.0)
        op = co[g, 3]
        with np.errstate(over="ignore"):
            alpha = np.minimum(0.99, op * np.exp(np.minimum(power, 50.0)))
        acc = (power <= 0) & (alpha >= 1.0 / 255.0)
        keep = np.where(acc, 1.0 - alpha, 1.0)
        T_before = np.concatenate([[1.0], np.cumprod(keep)[:-1]])
        test_T = T_before * (1.0 - alpha)
        # rounding distance of log T at each entry: every accepted factor (1 - alpha) carries alpha / (1 - alpha) times the
        # entry's own log-alpha distance plus one rounding of the running product - accumulated over the entries visited,
        # instead of a blanket 1e-3 band around the 1e-4 threshold
        d_T = np.cumsum(np.where(acc, alpha / (1.0 - alpha) * d_pow + eps, 0.0)) + eps
        definite_stop = acc & (test_T < 1e-4 * np.exp(-d_T))
        n_vis = int(np.argmax(definite_stop)) + 1 if definite_stop.any() else len(g)
        v = slice(0, n_vis)
        amb_alpha = np.abs(np.log(np.maximum(alpha[v], 1e-300) * 255.0)) <= d_pow[v]
        amb_power = (np.abs(power[v]) <= d_pow[v]) & (op[v] >= 1.0 / 255.0 * 0.99)
        amb_stop = acc[v] & (np.abs(np.log(np.maximum(test_T[v], 1e-300) / 1e-4)) <= d_T[v])
        if (amb_alpha | amb_power | amb_stop).any():
            worst = max(worst, float(err[y, x]))
        else:
            unexplained += 1
            if detail is not None and bg is not None:
                stop_at = n_vis - 1 if definite_stop.any() else len(g)            # entries [0, stop_at) are applied (the stopping one is not)
                a_ = np.where(acc[:stop_at], alpha[:stop_at], 0.0)
                Tb = T_before[:stop_at]
                w = a_ * Tb
                rgb = geo["rgb"].astype(np.float64)[g[:stop_at]]                    # [n, 3]
                T_fin = float(Tb[-1] * (1.0 - a_[-1])) if stop_at else 1.0
                bg64 = np.asarray(bg, np.float64).reshape(3)
                c64 = (w[:, None] * rgb).sum(axis=0) + T_fin * bg64
                # S_i: colour blended behind entry i (suffix sums) + the background
                contrib = w[:, None] * rgb
                behind = np.concatenate([np.cumsum(contrib[::-1], axis=0)[::-1][1:], np.zeros((1, 3))], axis=0) + T_fin * bg64 if stop_at else np.zeros((0, 3))
                U = np.abs(dx[:stop_at]) + 3.5; V = np.abs(dy[:stop_at]) + 3.5
                cg = co[g[:stop_at]]
                mag_hc = 0.5 * np.abs(cg[:, 0]) * U * U + 0.5 * np.abs(cg[:, 2]) * V * V + np.abs(cg[:, 1]) * U * V
                d_all = d_pow[:stop_at] + 8 * eps * (mag_hc + 1.0)
                moves = (a_ < 0.99)[:, None] * np.abs(Tb[:, None] * rgb - behind / np.maximum(1.0 - a_, 1e-2)[:, None])
                bound = float((a_[:, None] * d_all[:, None] * moves).sum(axis=0).max()) if stop_at else 0.0
                got = np.asarray(color, np.float64)[:, y, x]
                detail.append((int(y), int(x), float(err[y, x]), float(np.abs(got - c64).max()),
                               float(np.abs(fw["color"][:, y, x].astype(np.float64) - c64).max()), bound))
    return len(ys), unexplained, worst


GATE_LOG = []          # (what, W, H, outliers, unexplained, worst) of every gate evaluated in this process


def assert_grads_elementwise(got, ref, what="", floor=1e-3, rtol=1e-2, stragglers=1e-4, straggler_rtol=1e-1):
    """Element-wise companion of the max-norm gradient gate (which a Gaussian with a small gradient can pass while being
    entirely wrong): EVERY entry whose reference magnitude is at least `floor` x the tensor's maximum must agree to `rtol`
    RELATIVE TO ITSELF.  A gradient entry is a float32 sum of cancelling terms (in the oracle as well), so its absolute error is
    that of the large entries: with a max-norm error of 2e-5 an entry at the floor is 2 % off in either implementation.  At
    most `stragglers` of the checked entries (never more than a handful) may therefore exceed `rtol`, and none may exceed
    `straggler_rtol` - an entry that is simply wrong (relative error ~1) fails.  Returns (entries checked, stragglers, worst)."""
    got = np.asarray(got, np.float64).reshape(-1); ref = np.asarray(ref, np.float64).reshape(-1)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    big = np.abs(ref) >= floor * max(np.abs(ref).max(), 1e-300)
    rel = np.abs(got[big] - ref[big]) / np.abs(ref[big])
    n, bad, worst = int(big.sum()), int((rel > rtol).sum()), float(rel.max()) if big.any() else 0.0
    assert bad <= max(2, int(stragglers * n)), "%s: %d of %d entries above %g x max differ by more than %g relative (worst %.3g)" % (
        what, bad, n, floor, rtol, worst)
    assert worst <= straggler_rtol, "%s: an entry above %g x max is off by %.3g relative" % (what, floor, worst)
    return n, bad, worst


def assert_contributor_counts(fw, n_contrib, color, W, H, what=""):
    """n_contrib (forward.cu:370: list position of the last contributor; what the backward pass starts from) against the oracle's:
    equal, except on pixels where re-walking the oracle's list finds an entry at a decision threshold - every differing pixel is
    accounted for, none is waved through.  Returns the number of differing pixels."""
    diff = np.asarray(n_contrib).reshape(H, W) != np.asarray(fw["n_contrib"]).reshape(H, W)
    n, n_bad, _ = account_outlier_pixels(fw, color, W, H, mask=diff)
    print("contributor counts %-20s %dx%d: %d pixel(s) differ, %d unexplained" % (what, W, H, n, n_bad))
    assert n_bad == 0, "%s: %d of %d pixels with another last contributor have no entry at a decision threshold" % (what, n_bad, n)
    assert n <= max(2, 1e-3 * W * H), "%s: %d pixels with another last contributor" % (what, n)
    return n


def assert_forward_gate(fw, color, W, H, tol=1e-4, what="", plain_tol=None, bg=None):
    """The strict forward gate: every pixel within `tol` of the oracle, except pixels with a provable threshold flip
    (account_outlier_pixels), which are bounded by one flipped entry's weight: alpha * T * |colour| <= 2 / 255.
    How much of the 1e-4 budget plain rounding uses is printed every time: the largest error among the pixels below `tol`, and -
    with `plain_tol` (the full-size configurations pass 5e-5: half the budget) - asserted in the only form that separates rounding
    from flips: EVERY pixel that is off by more than plain_tol must have an entry at a decision threshold too (a flipped entry of
    small weight moves a pixel by less than 1e-4; rounding alone must stay below plain_tol).
    With `bg` (round 6: the random and needle scenes, where a flat half-budget does not hold against a float32 ORACLE - the reference's
    exponent is itself a cancelling sum, 100:1 splats seen from far off their centre lose four digits of it) a pixel between plain_tol
    and tol without a flip must be EXPLAINED BY ROUNDING: the image under test within plain_tol of the float64 value of the reference's
    formula there, or its distance from the oracle inside the first-order rounding bound of the conics the pixel visits
    (account_outlier_pixels); the table of such pixels is printed."""
    n_out, n_bad, worst = account_outlier_pixels(fw, color, W, H, tol)
    err = np.abs(np.asarray(color, np.float64) - fw["color"]).max(axis=0)
    below = float(err[err <= tol].max()) if (err <= tol).any() else 0.0
    msg = "forward gate %-22s %dx%d: %d pixel(s) above %g, %d unexplained, worst explained %.3g; largest error below the gate %.3g" % (
        what, W, H, n_out, tol, n_bad, worst, below)
    n_mid = n_mid_bad = None
    if plain_tol is not None and plain_tol < tol:
        detail = [] if bg is not None else None
        n_mid, n_mid_bad, _ = account_outlier_pixels(fw, color, W, H, plain_tol, bg=bg, detail=detail)
        if detail:
            bad = 0
            for (y, x, e_orc, e_64, o_64, bound) in detail:
                ok = e_64 <= plain_tol or e_orc <= bound
                bad += 0 if ok else 1
                print("   %s pixel (%d, %d): %.2e from the oracle, %.2e from float64 (the oracle: %.2e), rounding bound %.2e%s" % (
                    what, y, x, e_orc, e_64, o_64, bound, "" if ok else "   <-- UNEXPLAINED"))
            n_mid_bad = bad
        plain = float(err[err <= plain_tol].max()) if (err <= plain_tol).any() else 0.0
        msg += "; %d pixel(s) above %g, %d of them without a flip; largest error of all the others %.3g" % (n_mid, plain_tol, n_mid_bad, plain)
    GATE_LOG.append((what, W, H, n_out, n_bad, worst, below, n_mid, n_mid_bad))
    print(msg)
    assert n_bad == 0, "%s: %d of %d outlier pixels (> %g) have no entry at a decision threshold" % (what, n_bad, n_out, tol)
    assert n_out <= max(2, 1e-4 * W * H), "%s: %d outlier pixels" % (what, n_out)
    assert worst <= 2.0 / 255.0 + 1e-3, "%s: explained outlier of %g" % (what, worst)
    if n_mid is not None:
        assert n_mid_bad == 0, "%s: %d pixel(s) without a threshold flip are off by more than %g" % (what, n_mid_bad, plain_tol)
        assert n_mid <= max(4, 2e-4 * W * H), "%s: %d pixels above %g" % (what, n_mid, plain_tol)
    return n_out
