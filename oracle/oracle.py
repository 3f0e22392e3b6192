"""ctypes/numpy front-end of the CPU oracle (oracle/libgm_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package gaussianmesh_amd never does.  See gm_oracle.c for the parity status
("parity unpinned" for the CUDA core) and the reference file:line each function restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "libgm_oracle.so")
    src = os.path.join(_HERE, "gm_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libgm_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_bin.restype = C.c_int64
        _LIB.orc_forward.restype = C.c_int64
        _LIB.orc_higher_msb.restype = C.c_uint32
        _LIB.orc_num_threads.restype = C.c_int
    return _LIB


def _f(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def num_threads():
    return lib().orc_num_threads()


def higher_msb(n):
    return int(lib().orc_higher_msb(C.c_uint32(n)))


def sh_to_rgb(deg, shs, dirs):
    shs = _f(shs); dirs = _f(dirs)
    N, M = shs.shape[0], shs.shape[1]
    out = np.zeros((N, 3), np.float32)
    lib().orc_sh_to_rgb(N, deg, M, _ptr(shs), _ptr(dirs), _ptr(out))
    return out


def preprocess(means, opac, view, proj, campos, W, H, tanx, tany, D=0, shs=None, colors_precomp=None,
               scales=None, rots=None, cov3D_precomp=None, mod=1.0):
    """Returns dict(radii, xy, depths, cov3D, rgb, conic_op, tiles, clamped)."""
    means = _f(means); P = means.shape[0]
    opac = _f(opac).reshape(-1)
    shs = _f(shs); colors_precomp = _f(colors_precomp); scales = _f(scales); rots = _f(rots)
    cov3D_precomp = _f(cov3D_precomp)
    view = _f(view).reshape(-1); proj = _f(proj).reshape(-1); campos = _f(campos).reshape(-1)
    M = shs.shape[1] if shs is not None else 0
    o = dict(radii=np.zeros(P, np.int32), xy=np.zeros((P, 2), np.float32), depths=np.zeros(P, np.float32),
             cov3D=np.zeros((P, 6), np.float32), rgb=np.zeros((P, 3), np.float32),
             conic_op=np.zeros((P, 4), np.float32), tiles=np.zeros(P, np.uint32),
             clamped=np.zeros((P, 3), np.uint8))
    lib().orc_preprocess(P, D, M, _ptr(means), _ptr(scales), C.c_float(mod), _ptr(rots), _ptr(opac), _ptr(shs),
                         _ptr(cov3D_precomp), _ptr(colors_precomp), _ptr(view), _ptr(proj), _ptr(campos),
                         W, H, C.c_float(tanx), C.c_float(tany), _ptr(o["radii"]), _ptr(o["xy"]),
                         _ptr(o["depths"]), _ptr(o["cov3D"]), _ptr(o["rgb"]), _ptr(o["conic_op"]),
                         _ptr(o["tiles"]), _ptr(o["clamped"]))
    if cov3D_precomp is not None:
        o["cov3D"] = cov3D_precomp.reshape(P, 6).copy()
    if colors_precomp is not None:
        o["rgb"] = colors_precomp.reshape(P, 3).copy()
    return o


def mark_visible(means, view, proj):
    means = _f(means); P = means.shape[0]
    out = np.zeros(P, np.uint8)
    lib().orc_mark_visible(P, _ptr(means), _ptr(_f(view).reshape(-1)), _ptr(_f(proj).reshape(-1)), _ptr(out))
    return out.astype(bool)


def bin_instances(geo, W, H):
    """Returns dict(R, keys, point_list, ranges[T,2])."""
    P = geo["radii"].shape[0]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    R = lib().orc_bin(P, _ptr(geo["xy"]), _ptr(geo["depths"]), _ptr(geo["radii"]), _ptr(geo["tiles"]),
                      W, H, None, None, None)
    keys = np.zeros(max(R, 1), np.uint64); vals = np.zeros(max(R, 1), np.uint32)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    lib().orc_bin(P, _ptr(geo["xy"]), _ptr(geo["depths"]), _ptr(geo["radii"]), _ptr(geo["tiles"]),
                  W, H, _ptr(keys), _ptr(vals), _ptr(ranges))
    return dict(R=int(R), keys=keys[:R], point_list=vals[:R], ranges=ranges)


def instance_needed(W, H, bins, geo):
    """uint8 [R]: 1 where some pixel of the instance's tile accepts the entry (checker for emission-time tile culling)."""
    R = bins["R"]
    out = np.zeros(max(R, 1), np.uint8)
    if R:
        tile_of = np.ascontiguousarray((bins["keys"] >> np.uint64(32)).astype(np.uint32))
        lib().orc_instance_needed(C.c_int64(R), W, H, _ptr(tile_of), _ptr(bins["point_list"]), _ptr(geo["xy"]),
                                  _ptr(geo["conic_op"]), _ptr(out))
    return out[:R]


def render_fwd(W, H, bins, geo, bg):
    out = np.zeros((3, H, W), np.float32); fT = np.zeros(H * W, np.float32); nc = np.zeros(H * W, np.uint32)
    pl = bins["point_list"] if bins["R"] > 0 else np.zeros(1, np.uint32)
    lib().orc_render_fwd(W, H, _ptr(bins["ranges"]), _ptr(pl), _ptr(geo["xy"]), _ptr(geo["rgb"]),
                         _ptr(geo["conic_op"]), _ptr(_f(bg).reshape(-1)), _ptr(out), _ptr(fT), _ptr(nc))
    return out, fT, nc


def render_bwd(W, H, bins, geo, bg, final_T, n_contrib, dL_dpix):
    P = geo["radii"].shape[0]
    dmean2D = np.zeros((P, 3), np.float32); dconic = np.zeros((P, 4), np.float32)
    dop = np.zeros(P, np.float32); dcol = np.zeros((P, 3), np.float32)
    pl = bins["point_list"] if bins["R"] > 0 else np.zeros(1, np.uint32)
    lib().orc_render_bwd(P, W, H, _ptr(bins["ranges"]), _ptr(pl), _ptr(_f(bg).reshape(-1)), _ptr(geo["xy"]),
                         _ptr(geo["conic_op"]), _ptr(geo["rgb"]), _ptr(final_T), _ptr(n_contrib),
                         _ptr(_f(dL_dpix)), _ptr(dmean2D), _ptr(dconic), _ptr(dop), _ptr(dcol))
    return dmean2D, dconic, dop, dcol


def preprocess_bwd(means, geo, view, proj, campos, W, H, tanx, tany, dmean2D, dconic, dcolor, D=0, shs=None,
                   scales=None, rots=None, mod=1.0):
    means = _f(means); P = means.shape[0]
    shs = _f(shs); scales = _f(scales); rots = _f(rots)
    M = shs.shape[1] if shs is not None else 0
    dmean3D = np.zeros((P, 3), np.float32); dcov = np.zeros((P, 6), np.float32)
    dsh = np.zeros((P, max(M, 1), 3), np.float32); dscale = np.zeros((P, 3), np.float32)
    drot = np.zeros((P, 4), np.float32)
    lib().orc_preprocess_bwd(P, D, M, _ptr(means), _ptr(geo["radii"]), _ptr(shs), _ptr(geo["clamped"]),
                             _ptr(scales), _ptr(rots), C.c_float(mod), _ptr(geo["cov3D"]),
                             _ptr(_f(view).reshape(-1)), _ptr(_f(proj).reshape(-1)), _ptr(_f(campos).reshape(-1)),
                             W, H, C.c_float(tanx), C.c_float(tany), _ptr(_f(dmean2D)), _ptr(_f(dconic)),
                             _ptr(dmean3D), _ptr(_f(dcolor)), _ptr(dcov), _ptr(dsh), _ptr(dscale), _ptr(drot))
    return dmean3D, dcov, dsh, dscale, drot


def forward_full(scene, cam, bg, D=None, use_precomp_cov=False, use_precomp_color=False, mod=1.0):
    """Composed forward through the stage functions; returns every intermediate (tests)."""
    D = scene.get("D", 3) if D is None else D
    kw = dict(D=D, mod=mod)
    if use_precomp_color:
        kw["colors_precomp"] = scene["colors_precomp"]
    else:
        kw["shs"] = scene["shs"]
    if use_precomp_cov:
        kw["cov3D_precomp"] = scene["cov3D_precomp"]
    else:
        kw["scales"] = scene["scales"]; kw["rots"] = scene["rots"]
    geo = preprocess(scene["means"], scene["opac"], cam["view"], cam["proj"], cam["campos"], cam["W"], cam["H"],
                     cam["tanx"], cam["tany"], **kw)
    bins = bin_instances(geo, cam["W"], cam["H"])
    color, fT, nc = render_fwd(cam["W"], cam["H"], bins, geo, bg)
    return dict(geo=geo, bins=bins, color=color, final_T=fT, n_contrib=nc)


def backward_full(scene, cam, bg, fwd, dL_dpix, D=None, use_precomp_cov=False, use_precomp_color=False, mod=1.0):
    D = scene.get("D", 3) if D is None else D
    geo, bins = fwd["geo"], fwd["bins"]
    dmean2D, dconic, dop, dcol = render_bwd(cam["W"], cam["H"], bins, geo, bg, fwd["final_T"], fwd["n_contrib"], dL_dpix)
    dmean3D, dcov, dsh, dscale, drot = preprocess_bwd(
        scene["means"], geo, cam["view"], cam["proj"], cam["campos"], cam["W"], cam["H"], cam["tanx"], cam["tany"],
        dmean2D, dconic, dcol, D=D, shs=None if use_precomp_color else scene["shs"],
        scales=None if use_precomp_cov else scene["scales"], rots=None if use_precomp_cov else scene["rots"], mod=mod)
    return dict(dmean2D=dmean2D, dconic=dconic, dopacity=dop, dcolor=dcol, dmean3D=dmean3D, dcov3D=dcov,
                dsh=dsh, dscale=dscale, drot=drot)


def forward_fast(scene, cam, bg, D=None, mod=1.0, use_precomp_cov=False, use_precomp_color=False):
    """OpenMP whole-forward (orc_forward): the CPU baseline of bench.py.  Returns (color, radii, R)."""
    D = scene.get("D", 3) if D is None else D
    means = _f(scene["means"]); P = means.shape[0]
    shs = None if use_precomp_color else _f(scene["shs"])
    colp = _f(scene["colors_precomp"]) if use_precomp_color else None
    scales = None if use_precomp_cov else _f(scene["scales"])
    rots = None if use_precomp_cov else _f(scene["rots"])
    covp = _f(scene["cov3D_precomp"]) if use_precomp_cov else None
    M = shs.shape[1] if shs is not None else 0
    W, H = cam["W"], cam["H"]
    out = np.zeros((3, H, W), np.float32); radii = np.zeros(P, np.int32)
    R = lib().orc_forward(P, D, M, _ptr(_f(bg).reshape(-1)), W, H, _ptr(means), _ptr(shs), _ptr(colp),
                          _ptr(_f(scene["opac"]).reshape(-1)), _ptr(scales), C.c_float(mod), _ptr(rots), _ptr(covp),
                          _ptr(_f(cam["view"]).reshape(-1)), _ptr(_f(cam["proj"]).reshape(-1)),
                          _ptr(_f(cam["campos"]).reshape(-1)), C.c_float(cam["tanx"]), C.c_float(cam["tany"]),
                          _ptr(out), _ptr(radii), None, None)
    return out, radii, int(R)


def knn_mean_dist2(pts):
    pts = _f(pts); P = pts.shape[0]
    out = np.zeros(P, np.float32)
    lib().orc_knn_mean_dist2(P, _ptr(pts), _ptr(out))
    return out


def bary_weights(g, p1, p2, p3):
    g, p1, p2, p3 = (np.ascontiguousarray(a, np.float64) for a in (g, p1, p2, p3))
    w = np.zeros((g.shape[0], 3), np.float64)
    lib().orc_bary_weights(g.shape[0], _ptr(g), _ptr(p1), _ptr(p2), _ptr(p3), _ptr(w))
    return w


def deform(tri, w, dV, Rv, Sv, cov, pos):
    tri = np.ascontiguousarray(tri, np.int32); N = tri.shape[0]
    w, dV, Rv, Sv, cov, pos = (_f(a) for a in (w, dV, Rv, Sv, cov, pos))
    pos_o = np.zeros((N, 3), np.float32); cov_o = np.zeros((N, 3, 3), np.float32); rot_o = np.zeros((N, 3, 3), np.float32)
    lib().orc_deform(N, _ptr(tri), _ptr(w), _ptr(dV), _ptr(Rv), _ptr(Sv), _ptr(cov), _ptr(pos),
                     _ptr(pos_o), _ptr(cov_o), _ptr(rot_o))
    return pos_o, cov_o, rot_o


def sh_colors_rotated(pos, campos, rot, shs, deg=3):
    pos, rot, shs = _f(pos), _f(rot), _f(shs)
    N, M = shs.shape[0], shs.shape[1]
    rgb = np.zeros((N, 3), np.float32)
    lib().orc_sh_colors_rotated(N, deg, M, _ptr(pos), _ptr(_f(campos).reshape(-1)), _ptr(rot), _ptr(shs), _ptr(rgb))
    return rgb
