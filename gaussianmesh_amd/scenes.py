"""Synthetic scenes, cameras and mesh deformations for tests and bench.py.

Everything here is numpy (host side, deterministic from a seed); the arrays are uploaded by the
caller.  Camera matrices follow the reference conventions exactly:
  * world_view_transform = getWorld2View2(R, T).T, projection = getProjectionMatrix(...).T,
    full_proj = view @ proj, camera_center = inv(view)[3, :3]      (scene/cameras.py:47-50)
  * getWorld2View2 / getProjectionMatrix                             (utils/graphics_utils.py:38-71)
so element [i] of the flattened float32 arrays is what the kernels index
(cuda_rasterizer/auxiliary.h:57-76).
"""
import math

import numpy as np


# ----------------------------------------------------------------------------------------------
# cameras
def world2view2(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """World-to-view matrix [A | b; 0 1] of a camera whose world-to-view rotation block is R^T and translation is t, after its
    centre has been moved by `translate` and scaled by `scale`.  Same result as the reference's getWorld2View2
    (utils/graphics_utils.py:38-50, which inverts the 4x4 twice); pinned against it by tests/golden/camera.npz.
    The rotation block is untouched by the recentring, so only the translation column is recomputed:
    centre c solves A c + t = 0, c' = (c + translate) * scale, b = -A c'."""
    A = np.asarray(R, np.float64).T
    centre = np.linalg.solve(A, -np.asarray(t, np.float64))
    centre = (centre + np.asarray(translate, np.float64)) * scale
    M = np.eye(4)
    M[:3, :3] = A
    M[:3, 3] = -A @ centre
    return M.astype(np.float32)


def projection_matrix(znear, zfar, fovX, fovY):
    """Perspective matrix of a symmetric frustum in the reference's convention (utils/graphics_utils.py:52-71
    getProjectionMatrix: z maps to [0, 1], w = +z, float32 storage).  For a symmetric frustum the off-centre terms
    (r+l)/(r-l), (t+b)/(t-b) vanish and 2n/(r-l) = 1/tan(fovX/2)."""
    P = np.zeros((4, 4), np.float32)
    half_w = math.tan(fovX / 2) * znear
    half_h = math.tan(fovY / 2) * znear
    P[0, 0] = znear / half_w
    P[1, 1] = znear / half_h
    depth = zfar - znear
    P[2, 2] = zfar / depth
    P[2, 3] = -(zfar * znear) / depth
    P[3, 2] = 1.0
    return P


def camera_from_RT(R, T, fovx, fovy, W, H, znear=0.01, zfar=100.0):
    """scene/cameras.py:47-50: R is camera-to-world rotation, T the world-to-camera translation."""
    view = world2view2(R, T).transpose().copy()                       # stored transposed
    proj = projection_matrix(znear, zfar, fovx, fovy).transpose().copy()
    full = (view @ proj).astype(np.float32)
    campos = np.linalg.inv(view)[3, :3].astype(np.float32)
    return dict(view=np.ascontiguousarray(view, np.float32), proj=np.ascontiguousarray(full, np.float32),
                campos=campos, W=int(W), H=int(H), tanx=math.tan(fovx * 0.5), tany=math.tan(fovy * 0.5),
                fovx=fovx, fovy=fovy)


def look_at_camera(eye, target, W, H, fovx_deg=60.0, up=(0.0, 1.0, 0.0)):
    eye = np.asarray(eye, np.float64); target = np.asarray(target, np.float64); up = np.asarray(up, np.float64)
    f = target - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, up); r /= np.linalg.norm(r)        # camera +x (right)
    d = np.cross(f, r)                                  # camera +y (down)
    R = np.stack([r, d, f], axis=1)                     # columns = camera axes in world (c2w)
    T = -R.T @ eye
    fovx = math.radians(fovx_deg)
    fovy = 2.0 * math.atan(math.tan(fovx / 2) * H / W)
    return camera_from_RT(R, T, fovx, fovy, W, H)


def orbit_camera(k, K, W, H, radius=8.0, height=1.5, fovx_deg=60.0):
    """Camera k of K on the benchmark circle."""
    a = 2.0 * math.pi * k / K
    return look_at_camera((radius * math.cos(a), height, radius * math.sin(a)), (0, 0, 0), W, H, fovx_deg)


# ----------------------------------------------------------------------------------------------
# clouds
def num_sh_coeffs(deg):
    return (deg + 1) ** 2


def _ball(rng, P):
    out = np.empty((0, 3))
    while out.shape[0] < P:
        c = rng.uniform(-1, 1, size=(int((P - out.shape[0]) * 2.2) + 16, 3))
        out = np.concatenate([out, c[(c * c).sum(1) <= 1.0]], 0)
    return out[:P]


def make_cloud(P, seed=0, M=16, extent=3.0, scale_lo=0.005, scale_hi=0.05, D=3):
    """Random Gaussian cloud.  Layouts as at the reference boundary:
    means [P,3], scales [P,3] (post-exp), rots [P,4] (r,x,y,z normalised), opac [P,1] (post-sigmoid),
    shs [P,M,3]."""
    rng = np.random.default_rng(seed)
    means = (_ball(rng, P) * extent).astype(np.float32)
    scales = np.exp(rng.uniform(math.log(scale_lo), math.log(scale_hi), size=(P, 3))).astype(np.float32)
    q = rng.normal(size=(P, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = rng.uniform(0.05, 0.95, size=(P, 1)).astype(np.float32)
    shs = np.empty((P, M, 3), np.float32)
    shs[:, 0] = rng.normal(0, 0.5, size=(P, 3))
    if M > 1:
        shs[:, 1:] = rng.normal(0, 0.1, size=(P, M - 1, 3))
    return dict(means=means, scales=scales, rots=q.astype(np.float32), opac=opac, shs=shs, D=D)


def cov3d_from_scale_rot(scales, rots, mod=1.0):
    """Host helper: Sigma = R diag(s)^2 R^T as the 6 upper-triangular floats
    (utils/general_utils.py:64-72 strip_symmetric layout: xx,xy,xz,yy,yz,zz)."""
    q = rots.astype(np.float64); q = q / np.linalg.norm(q, axis=1, keepdims=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                  2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                  2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    L = R * (mod * scales.astype(np.float64))[:, None, :]
    S = L @ L.transpose(0, 2, 1)
    return S


def strip_symmetric(S):
    return np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# proxy mesh + binding + deformation  (config C3)
def torus_mesh(nu=100, nv=75, R=2.0, r=0.7):
    """nu x nv quads -> 2*nu*nv triangles, nu*nv vertices (100x75 -> 15000 faces / 7500 verts)."""
    u = np.arange(nu) * (2 * math.pi / nu)
    v = np.arange(nv) * (2 * math.pi / nv)
    U, Vv = np.meshgrid(u, v, indexing="ij")
    X = (R + r * np.cos(Vv)) * np.cos(U)
    Y = r * np.sin(Vv)
    Z = (R + r * np.cos(Vv)) * np.sin(U)
    verts = np.stack([X, Y, Z], -1).reshape(-1, 3)
    idx = np.arange(nu * nv).reshape(nu, nv)
    a = idx
    b = np.roll(idx, -1, 0)
    c = np.roll(np.roll(idx, -1, 0), -1, 1)
    d = np.roll(idx, -1, 1)
    faces = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)], 0)
    return verts.astype(np.float64), faces.astype(np.int32)


def bind_cloud_to_mesh(P, verts, faces, seed=0, M=16, alpha_distance=4.0):
    """Gaussians bound to faces: x = softmax(bc).(v1,v2,v3) + 4 r (sigmoid(d)-0.5) n
    (scene/mesh_based_gaussian_model.py:138-152; r = mean edge length :208-215)."""
    rng = np.random.default_rng(seed)
    cloud = make_cloud(P, seed=seed + 1, M=M)
    F = faces.shape[0]
    fid = rng.integers(F, size=P).astype(np.int32)
    bc_raw = rng.normal(size=(P, 3))
    e = np.exp(bc_raw - bc_raw.max(1, keepdims=True)); bc = e / e.sum(1, keepdims=True)
    dist = rng.normal(0, 0.3, size=(P, 1))
    tri = faces[fid]
    v1, v2, v3 = verts[tri[:, 0]], verts[tri[:, 1]], verts[tri[:, 2]]
    n = np.cross(v2 - v1, v3 - v1); n /= np.linalg.norm(n, axis=1, keepdims=True)
    rr = (np.linalg.norm(v2 - v1, axis=1) + np.linalg.norm(v3 - v2, axis=1) + np.linalg.norm(v1 - v3, axis=1))[:, None] / 3
    proj = bc[:, :1] * v1 + bc[:, 1:2] * v2 + bc[:, 2:3] * v3
    xyz = proj + alpha_distance * rr * (1 / (1 + np.exp(-dist)) - 0.5) * n
    # scale the splats to the local face size so the bound cloud looks like a surface
    cloud["scales"] = (cloud["scales"] * (rr / 0.12)).astype(np.float32)
    cloud["means"] = xyz.astype(np.float32)
    cloud.update(fid=fid, tri=tri.astype(np.int32), weights=bc.astype(np.float32), proj_xyz=proj)
    return cloud


def twist_bend_frame(verts, t, period=64):
    """Analytic twist about +y: phi(p) = Rot_y(a*y) p, a = 0.5 sin(2 pi t / period).
    Returns deformed verts V1 and per-vertex (R, S) from the polar decomposition F = Q S of the analytic Jacobian
    F = Rot_y(theta) + a (Rot_y'(theta) p) e_y^T, with R = Q^T: the row-vector convention in which the reference's
    deform_gaussian takes pyACAP's GetRS output (it transposes the blended R and transforms covariances by R^T S,
    edittool/__init__.py:118-129, which is F C F^T).  gm_mesh_rs computes the same pair from the mesh itself."""
    a = 0.5 * math.sin(2 * math.pi * t / period)
    th = a * verts[:, 1]
    c, s = np.cos(th), np.sin(th)
    Rot = np.zeros((verts.shape[0], 3, 3))
    Rot[:, 0, 0] = c; Rot[:, 0, 2] = s; Rot[:, 1, 1] = 1; Rot[:, 2, 0] = -s; Rot[:, 2, 2] = c
    dRot = np.zeros_like(Rot)
    dRot[:, 0, 0] = -s; dRot[:, 0, 2] = c; dRot[:, 2, 0] = -c; dRot[:, 2, 2] = -s
    V1 = np.einsum("nij,nj->ni", Rot, verts)
    Fm = Rot.copy()
    Fm[:, :, 1] += a * np.einsum("nij,nj->ni", dRot, verts)
    U, sig, Vt = np.linalg.svd(Fm)
    Q = U @ Vt
    Sp = Vt.transpose(0, 2, 1) @ (sig[:, :, None] * Vt)
    return V1, np.ascontiguousarray(Q.transpose(0, 2, 1)), Sp
