"""File-based edit surface: the classes edit.py drives, with the reference's names and call signatures.

Mirrors edittool/__init__.py of the reference:
  SingleObjectDeform(fg_path, mesh_path, name)      :40-131   load_gaussian / load_mesh (both branches) / deform_gaussian(path)
  SceneVisualTool(bg_gaussian_path)                 :133-231  add_gaussian / deform_one_gaussian / render_gaussian (background
                                                              cloud + objects, eigh -> (scale, quaternion) route) / get_camera
  ObjectVisualTool()                                :378-475  the same without a background (colors_precomp + cov3D_precomp route)
Replaced dependencies: plyfile -> io.read_ply, igl.read_triangle_mesh -> io.read_obj, igl.point_mesh_squared_distance ->
closest_triangles() below, pyACAP.GetRS -> gm_mesh_rs (deform.mesh_rs), Jittor tensor algebra -> the HIP kernels behind
deform.SingleObjectDeform / renderer.  Host-side set-up (parsing, barycentric weights) is numpy, once per object; every
per-frame step runs on the GPU.
"""
import os

import numpy as np
import torch

from . import io as gio
from .deform import SingleObjectDeform as _TensorObject
from .deform import barycentric_weights, cov_to_scale_rot, mesh_rs, vertex_face_adjacency
from .rasterizer import GaussianRasterizationSettings, NewGaussianRasterizer
from .renderer import Camera, camera_work_hint, render_deformed


def _covariance(scaling_raw, rotation_raw):
    """build_covariance_from_scaling_rotation (edittool/mesh_based_gaussian.py:23-27, general_utils.py:39-71):
    L = R(q / |q|) diag(exp(s)), cov = L L^T, full [N,3,3]."""
    s = torch.exp(scaling_raw)
    q = torch.nn.functional.normalize(rotation_raw)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                     2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    L = R * s[:, None, :]
    return L @ L.transpose(1, 2)


def point_mesh_squared_distance(points, vertex, triangles, chunk=4096):
    """igl.point_mesh_squared_distance(P, V, F) -> (sqrD [n], I [n], C [n,3]) (edittool/__init__.py:80): squared distance to,
    index of, and closest point on the closest triangle, exact, brute force in chunks: closest point on each triangle by the
    region test of Ericson, "Real-Time Collision Detection" 5.1.5 (ties go to the lowest face index).  Host side (numpy, float64),
    used once per object and only when the Gaussian file carries no face ids."""
    P = np.asarray(points, np.float64).reshape(-1, 3); V = np.asarray(vertex, np.float64); F = np.asarray(triangles, np.int64)
    a, b, c = V[F[:, 0]][None], V[F[:, 1]][None], V[F[:, 2]][None]
    ab, ac = b - a, c - a
    idx = np.zeros(len(P), np.int64); sqr = np.zeros(len(P), np.float64); close = np.zeros((len(P), 3), np.float64)
    for s0 in range(0, len(P), chunk):
        p = P[s0:s0 + chunk][:, None, :]
        ap = p - a
        d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
        bp = p - b
        d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
        cp = p - c
        d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
        vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
        with np.errstate(divide="ignore", invalid="ignore"):
            denom = 1.0 / (va + vb + vc)
            v_in, w_in = vb * denom, vc * denom
            t_ab = d1 / (d1 - d3); t_ac = d2 / (d2 - d6); t_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        q = a + ab * v_in[..., None] + ac * w_in[..., None]                          # interior
        m = (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0); q = np.where(m[..., None], b + (c - b) * t_bc[..., None], q)
        m = (vb <= 0) & (d2 >= 0) & (d6 <= 0); q = np.where(m[..., None], a + ac * t_ac[..., None], q)
        m = (vc <= 0) & (d1 >= 0) & (d3 <= 0); q = np.where(m[..., None], a + ab * t_ab[..., None], q)
        m = (d6 >= 0) & (d5 <= d6); q = np.where(m[..., None], c, q)
        m = (d3 >= 0) & (d4 <= d3); q = np.where(m[..., None], b, q)
        m = (d1 <= 0) & (d2 <= 0); q = np.where(m[..., None], a, q)
        d2all = ((p - q) ** 2).sum(-1)
        k = np.argmin(d2all, axis=1)
        rows = np.arange(len(k))
        idx[s0:s0 + chunk] = k; sqr[s0:s0 + chunk] = d2all[rows, k]; close[s0:s0 + chunk] = q[rows, k]
    return sqr, idx, close


def closest_triangles(points, vertex, triangles, chunk=4096):
    """Index of the closest triangle to each point (point_mesh_squared_distance's second output, edittool/__init__.py:82)."""
    return point_mesh_squared_distance(points, vertex, triangles, chunk)[1]


class SingleObjectDeform(_TensorObject):
    """edittool.SingleObjectDeform(fg_path, mesh_path, name): a mesh-bound Gaussian PLY attached to its proxy mesh (OBJ)."""

    def __init__(self, fg_path, mesh_path, name=None, device="cuda"):
        self.device = torch.device(device)
        self.load_gaussian(fg_path)
        self.load_mesh(mesh_path)
        self.name = mesh_path if name is None else name

    def load_gaussian(self, gaussian_path):
        """:48-64.  The edit tool's loader fills _bc from the saved x, y, z (edittool/mesh_based_gaussian.py:183-184), so
        get_proj_xyz = softmax(xyz) . (v1, v2, v3); positions are the SAVED x, y, z (get_load_xyz)."""
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=self.device)
        m = gio.load_mesh_gaussians(gaussian_path, bc_from_xyz=True)
        self._loaded = m
        pos = t(m["load_xyz"])
        bc = torch.softmax(t(m["bc"]), dim=1)
        self.gaussian_proj_pos = bc[:, 0:1] * t(m["v1"]) + bc[:, 1:2] * t(m["v2"]) + bc[:, 2:3] * t(m["v3"])
        cov = _covariance(t(m["scaling"]), t(m["rotation"]))
        opacity = torch.sigmoid(t(m["opacity"]))
        feats = torch.cat([t(m["features_dc"]), t(m["features_rest"])], dim=1).contiguous()
        self.index_tri = np.asarray(m["fid"], np.int64) if m.get("fid") is not None else None
        self._pending = (pos, cov, opacity, feats)

    def load_mesh(self, mesh_path):
        """:65-102.  With face ids in the file (always, for files written by the training code): gaussian_triangles =
        F[fid] and the weights are taken at the projected positions; without: closest triangle of the projected position,
        weights at the foot of the perpendicular from the Gaussian onto that triangle's plane."""
        vertex, triangles = gio.read_obj(mesh_path)
        pos, cov, opacity, feats = self._pending
        if self.index_tri is None:
            normals = np.cross(vertex[triangles[:, 1]] - vertex[triangles[:, 0]], vertex[triangles[:, 2]] - vertex[triangles[:, 0]])
            normals /= np.linalg.norm(normals, axis=1)[:, None]
            bias = -(vertex[triangles[:, 0]] * normals).sum(axis=1)
            gpos = pos.cpu().numpy().astype(np.float64)
            index_tri = closest_triangles(self.gaussian_proj_pos.cpu().numpy(), vertex, triangles)
            n_g, b_g = normals[index_tri], bias[index_tri]
            distance = -((n_g * gpos).sum(axis=1) + b_g)
            intersection = gpos + distance[:, None] * n_g
            self.index_tri = index_tri[:, None]
        else:
            intersection = self.gaussian_proj_pos.cpu().numpy().astype(np.float64)
        tri = triangles[self.index_tri.reshape(-1)]
        coord = barycentric_weights(intersection, vertex[tri[:, 0]], vertex[tri[:, 1]], vertex[tri[:, 2]])
        t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=self.device)
        super().__init__(pos, cov, opacity, feats, t(tri, torch.int32), t(coord), t(vertex), name=None)
        self.faces = t(triangles, torch.int32)
        off, adj = vertex_face_adjacency(triangles, vertex.shape[0])
        self._adjacency = (t(off, torch.int32), t(adj, torch.int32))
        del self._pending

    def deform_gaussian(self, deform_mesh_path):
        """:103-131: read the deformed mesh, per-vertex (R, S) of the deformation (pyACAP.GetRS there, gm_mesh_rs here),
        then the Gaussian deformation; sets gaussian_deform_pos / gaussian_deform_cov / gaussian_deform_rot."""
        deform_vertex, _ = gio.read_obj(deform_mesh_path)
        return self.deform_vertices(torch.as_tensor(deform_vertex, dtype=torch.float32, device=self.device))

    def deform_vertices(self, deform_vertex):
        """the same from a [Vm,3] tensor of deformed vertices (an animation loop needs no file per frame)"""
        R, S = mesh_rs(self.vertex, deform_vertex, self.faces, adjacency=self._adjacency)
        return self.deform(deform_vertex, R, S)


class ObjectVisualTool:
    """edittool.ObjectVisualTool (:378-475): objects on a white background, colours from SH with the deformation-rotated
    view direction, colors_precomp + cov3D_precomp into NewGaussianRasterizer."""

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self.gaussians_list = []

    def add_gaussian(self, gaussian_path, mesh_path, name=None):
        self.gaussians_list.append(SingleObjectDeform(gaussian_path, mesh_path, name, device=self.device))

    def deform_one_gaussian(self, name, deform_mesh_path):
        for g in self.gaussians_list:
            if g.get_name() == name:
                g.deform_gaussian(deform_mesh_path)

    def get_camera(self, path):
        """cameras.json of a model directory -> cameras with the reference's attribute names (:547-584)"""
        return [Camera(c, self.device) for c in gio.load_cameras_json(os.path.join(path, "cameras.json"))]

    def get_single_camera(self, path, id=1):
        return self.get_camera(path)[id]

    def render_gaussian(self, viewpoint_camera):
        return render_deformed(viewpoint_camera, self.gaussians_list)


class SceneVisualTool(ObjectVisualTool):
    """edittool.SceneVisualTool (:133-231): a free-standing background cloud plus deformed objects.  As in the reference
    the concatenated covariances go through the eigen-decomposition route - (scale, quaternion) from eigh, here on the
    device (gm_cov_to_scale_rot) - and the colours come from the rasterizer's own SH evaluation (`shs`, :209-217)."""

    def __init__(self, bg_gaussian_path=None, device="cuda"):
        super().__init__(device)
        self.load_bg_gaussian(bg_gaussian_path)

    def load_bg_gaussian(self, path):
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=self.device)
        m = gio.load_plain_gaussians(path)
        self.bg_scale = torch.exp(t(m["scaling"]))
        self.bg_rot = torch.nn.functional.normalize(t(m["rotation"]))
        self.bg_cov3D = _covariance(t(m["scaling"]), t(m["rotation"]))
        self.bg_mean3D = t(m["xyz"])
        self.bg_shs = torch.cat([t(m["features_dc"]), t(m["features_rest"])], dim=1).contiguous()
        self.bg_opacity = torch.sigmoid(t(m["opacity"]))
        self.bg_deform_rot = torch.eye(3, device=self.device).repeat(self.bg_scale.shape[0], 1, 1)

    def render_gaussian(self, viewpoint_camera):
        import math
        c = viewpoint_camera
        objs = self.gaussians_list
        means3D = torch.cat([self.bg_mean3D] + [o.gaussian_deform_pos for o in objs], dim=0)
        shs = torch.cat([self.bg_shs] + [o.gaussian_feature for o in objs], dim=0)
        cov = torch.cat([self.bg_cov3D] + [o.gaussian_deform_cov for o in objs], dim=0)
        opacity = torch.cat([self.bg_opacity] + [o.gaussian_o for o in objs], dim=0)
        new_s, new_q = cov_to_scale_rot(cov)
        rs = GaussianRasterizationSettings(int(c.image_height), int(c.image_width), math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5),
                                           torch.ones(3, device=self.device), 1, c.world_view_transform, c.full_proj_transform, 3,
                                           c.camera_center, False, False, camera_work_hint(c, self.device))
        image, _ = NewGaussianRasterizer(rs)(means3D=means3D, means2D=torch.zeros_like(means3D), shs=shs, colors_precomp=None,
                                             opacities=opacity, scales=new_s, rotations=new_q, cov3D_precomp=None)
        return image
