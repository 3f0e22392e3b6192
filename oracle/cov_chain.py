"""numpy restatement of the covariance part of the preprocess backward (RAST/backward.cu:144-341: computeCov2DCUDA backward up to
dL/dcov3D, computeCov3D backward to dL/dscale, dL/drot) with a selectable float type per block.  TEST INFRASTRUCTURE (tests/ and tools/
only).  In float32 throughout it reproduces oracle/gm_oracle.c orc_preprocess_bwd bit for bit on these outputs (same association order);
with dtB = float64 it is what gm_preprocess.hip does since round 5: covariance entries, determinant and dL/d(a, b, c) in binary64 from the
float32 T and cov3D (tests/test_needle_conditioning.py, tools/needle_stages.py)."""
import numpy as np


def chain(sc, cam, geo, dconic, dt2=np.float32, dt3=np.float32, mod=1.0, dtB=None, dtC=None):
    """dt2: float type of the computeCov2D block (T, J, ...), dtB: of its conic -> cov2D part (a, b, c, det, dL/da, dL/db, dL/dc; default dt2),
    dtC: of the dL/dcov3D assembly behind it (default dtB), dt3: of the computeCov3D backward.  dconic: [P, 4] as backward.cu:549-551 leaves it
    (x, y, -, w slots; y holds HALF the off-diagonal derivative).  Returns (dL/dcov3D [P, 6] float32, dL/dscale, dL/drot)."""
    W, H = cam["W"], cam["H"]
    f = lambda x: np.asarray(x, dt2)
    tanx, tany = f(cam["tanx"]), f(cam["tany"])
    fx, fy = f(W) / (f(2.0) * tanx), f(H) / (f(2.0) * tany)
    v = f(cam["view"]).reshape(-1)
    mean = f(sc["means"]); c3 = f(geo["cov3D"])
    t = np.stack([mean[:, 0] * v[0 + k] + mean[:, 1] * v[4 + k] + mean[:, 2] * v[8 + k] + v[12 + k] for k in range(3)], 1)
    limx, limy = f(1.3) * tanx, f(1.3) * tany
    txtz, tytz = t[:, 0] / t[:, 2], t[:, 1] / t[:, 2]
    tx = np.clip(txtz, -limx, limx) * t[:, 2]; ty = np.clip(tytz, -limy, limy) * t[:, 2]; tz = t[:, 2]
    j00, j02, j11, j12 = fx / tz, -(fx * tx) / (tz * tz), fy / tz, -(fy * ty) / (tz * tz)
    T0 = np.stack([v[4 * i] * j00 + v[4 * i + 2] * j02 for i in range(3)], 1)
    T1 = np.stack([v[4 * i + 1] * j11 + v[4 * i + 2] * j12 for i in range(3)], 1)
    if dtB is not None:
        T0, T1, c3 = T0.astype(dtB), T1.astype(dtB), c3.astype(dtB); f = lambda x: np.asarray(x, dtB)
    V = np.stack([c3[:, 0], c3[:, 1], c3[:, 2], c3[:, 1], c3[:, 3], c3[:, 4], c3[:, 2], c3[:, 4], c3[:, 5]], 1)
    A0 = np.stack([T0[:, 0] * V[:, k] + T0[:, 1] * V[:, 3 + k] + T0[:, 2] * V[:, 6 + k] for k in range(3)], 1)
    A1 = np.stack([T1[:, 0] * V[:, k] + T1[:, 1] * V[:, 3 + k] + T1[:, 2] * V[:, 6 + k] for k in range(3)], 1)
    a = (A0 * T0).sum(1) + f(0.3); b = (A1 * T0).sum(1); c = (A1 * T1).sum(1) + f(0.3)
    dcx, dcy, dcz = f(dconic[:, 0]), f(dconic[:, 1]), f(dconic[:, 3])
    denom = a * c - b * b
    d2 = f(1.0) / (denom * denom + f(1e-7))
    dL_da = d2 * (-c * c * dcx + 2 * b * c * dcy + (denom - a * c) * dcz)
    dL_dc = d2 * (-a * a * dcz + 2 * a * b * dcy + (denom - a * c) * dcx)
    dL_db = d2 * 2 * (b * c * dcx - (denom + 2 * b * b) * dcy + a * b * dcz)
    if dtC is not None:
        T0, T1, dL_da, dL_db, dL_dc = [u.astype(dtC) for u in (T0, T1, dL_da, dL_db, dL_dc)]
    dcov = np.stack([T0[:, 0] * T0[:, 0] * dL_da + T0[:, 0] * T1[:, 0] * dL_db + T1[:, 0] * T1[:, 0] * dL_dc,
                     2 * T0[:, 0] * T0[:, 1] * dL_da + (T0[:, 0] * T1[:, 1] + T0[:, 1] * T1[:, 0]) * dL_db + 2 * T1[:, 0] * T1[:, 1] * dL_dc,
                     2 * T0[:, 0] * T0[:, 2] * dL_da + (T0[:, 0] * T1[:, 2] + T0[:, 2] * T1[:, 0]) * dL_db + 2 * T1[:, 0] * T1[:, 2] * dL_dc,
                     T0[:, 1] * T0[:, 1] * dL_da + T0[:, 1] * T1[:, 1] * dL_db + T1[:, 1] * T1[:, 1] * dL_dc,
                     2 * T0[:, 2] * T0[:, 1] * dL_da + (T0[:, 1] * T1[:, 2] + T0[:, 2] * T1[:, 1]) * dL_db + 2 * T1[:, 1] * T1[:, 2] * dL_dc,
                     T0[:, 2] * T0[:, 2] * dL_da + T0[:, 2] * T1[:, 2] * dL_db + T1[:, 2] * T1[:, 2] * dL_dc], 1)
    vis = geo["radii"] > 0
    dcov = np.where(vis[:, None], dcov, 0).astype(np.float32)        # handed on as float32 (the C ABI's dL_dcov3D)
    g = lambda x: np.asarray(x, dt3)
    q = g(sc["rots"]); r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rg = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                   2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1)
    s = g(mod) * g(sc["scales"])
    Mc = np.stack([s[:, k] * Rg[:, 3 * cc + k] for cc in range(3) for k in range(3)], 1)
    d = g(dcov)
    dS = np.stack([d[:, 0], 0.5 * d[:, 1], 0.5 * d[:, 2], 0.5 * d[:, 1], d[:, 3], 0.5 * d[:, 4], 0.5 * d[:, 2], 0.5 * d[:, 4], d[:, 5]], 1)
    dM = np.stack([(2 * Mc[:, i]) * dS[:, 3 * j] + (2 * Mc[:, 3 + i]) * dS[:, 3 * j + 1] + (2 * Mc[:, 6 + i]) * dS[:, 3 * j + 2] for j in range(3) for i in range(3)], 1)
    Rt = np.stack([Rg[:, 3 * rr + cc] for cc in range(3) for rr in range(3)], 1)
    dMt = np.stack([dM[:, 3 * rr + cc] for cc in range(3) for rr in range(3)], 1)
    dsc = np.stack([(Rt[:, 3 * cc:3 * cc + 3] * dMt[:, 3 * cc:3 * cc + 3]).sum(1) for cc in range(3)], 1)
    dMt = dMt * np.repeat(s, 3, axis=1)
    DM = lambda cc, rr: dMt[:, 3 * cc + rr]
    dq = np.stack([2 * z * (DM(0, 1) - DM(1, 0)) + 2 * y * (DM(2, 0) - DM(0, 2)) + 2 * x * (DM(1, 2) - DM(2, 1)),
                   2 * y * (DM(1, 0) + DM(0, 1)) + 2 * z * (DM(2, 0) + DM(0, 2)) + 2 * r * (DM(1, 2) - DM(2, 1)) - 4 * x * (DM(2, 2) + DM(1, 1)),
                   2 * x * (DM(1, 0) + DM(0, 1)) + 2 * r * (DM(2, 0) - DM(0, 2)) + 2 * z * (DM(1, 2) + DM(2, 1)) - 4 * y * (DM(2, 2) + DM(0, 0)),
                   2 * r * (DM(0, 1) - DM(1, 0)) + 2 * x * (DM(2, 0) + DM(0, 2)) + 2 * y * (DM(1, 2) + DM(2, 1)) - 4 * z * (DM(1, 1) + DM(0, 0))], 1)
    return dcov, np.where(vis[:, None], dsc, 0), np.where(vis[:, None], dq, 0)
