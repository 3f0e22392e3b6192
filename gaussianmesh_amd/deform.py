"""Mesh-driven deformation of bound Gaussians (edit tool), tensor-in form.

Mirrors edittool/__init__.py of the reference:
  SingleObjectDeform (:40-131): attribute names gaussian_pos, gaussian_cov, gaussian_o, gaussian_feature,
  gaussian_deform_pos / gaussian_deform_cov / gaussian_deform_rot, gaussian_triangles, weight_g_pos, vertex.
  deform_gaussian(deform_mesh_path) of the reference reads an OBJ and calls pyACAP.GetRS; pyACAP is an external
  binary that is not in the reference tree, so here the per-vertex (R, S) are an explicit input:
      obj.deform(V1, R, S)   # V1 [Vm,3] deformed vertices, R/S [Vm,3,3]
The arithmetic (gather, barycentric blend, RS cov RS^T, x + dx) runs in one HIP kernel (csrc/gm_deform.hip).
"""
import torch

from . import _lib


def _f(t):
    if t.dtype is torch.float32 and t.is_contiguous():
        return t
    return t.detach().contiguous().float()


def barycentric_weights(points, p1, p2, p3):
    """Host-side (numpy, float64) barycentric weights from sub-triangle areas, as the reference computes them once
    per object when a mesh is attached (edittool/general_utils.py:73-88 get_barycentric_coordinate)."""
    import numpy as np
    e1, e2, e3 = points - p1, points - p2, points - p3
    s1 = np.linalg.norm(np.cross(e2, e3), axis=1)
    s2 = np.linalg.norm(np.cross(e1, e3), axis=1)
    s3 = np.linalg.norm(np.cross(e1, e2), axis=1)
    s = s1 + s2 + s3
    return np.stack([s1 / s, s2 / s, s3 / s], axis=1)


def deform_tensors(tri, w, dV, Rv, Sv, cov, pos):
    """gm_deform: returns (pos' [N,3], cov' [N,3,3], rot [N,3,3], cov6 [N,6])."""
    lib = _lib.lib()
    device = pos.device
    if device.type != "cuda":
        raise _lib.GmeshError("deform needs tensors on a HIP (cuda) device; there is no CPU path")
    tri = tri.detach().contiguous().to(torch.int32)
    w, dV, Rv, Sv, cov, pos = (_f(t) for t in (w, dV, Rv, Sv, cov, pos))
    N = pos.shape[0]
    f = dict(dtype=torch.float32, device=device)
    pos_o = torch.empty((N, 3), **f); cov_o = torch.empty((N, 3, 3), **f); rot_o = torch.empty((N, 3, 3), **f)
    cov6 = torch.empty((N, 6), **f)
    with torch.cuda.device(device):
        _lib.check(lib.gm_deform(N, tri.data_ptr(), w.data_ptr(), dV.data_ptr(), Rv.data_ptr(), Sv.data_ptr(), cov.data_ptr(),
                                 pos.data_ptr(), pos_o.data_ptr(), cov_o.data_ptr(), rot_o.data_ptr(), cov6.data_ptr(),
                                 torch.cuda.current_stream(device).cuda_stream))
    return pos_o, cov_o, rot_o, cov6


def sh_colors(pos, campos, shs, rot=None, deg=3):
    """gm_sh_colors: max(SH_deg(rot^T normalize(pos - campos)) + 0.5, 0)  (edittool/__init__.py:442-448)."""
    lib = _lib.lib()
    device = pos.device
    pos, campos, shs = _f(pos), _f(campos), _f(shs)
    rot = None if rot is None else _f(rot)
    N, M = shs.shape[0], shs.shape[1]
    rgb = torch.empty((N, 3), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.gm_sh_colors(N, int(deg), M, pos.data_ptr(), campos.data_ptr(), None if rot is None else rot.data_ptr(),
                                    shs.data_ptr(), rgb.data_ptr(), torch.cuda.current_stream(device).cuda_stream))
    return rgb


def deform_shade(tri, w, dV, Rv, Sv, cov, pos, shs, campos, deg=3, want_cov_rot=False):
    """gm_deform_shade: fused deform + rotated-direction SH colour.  Returns (pos' [N,3], cov6 [N,6], rgb [N,3]) and, with
    want_cov_rot, also (cov' [N,3,3], rot [N,3,3])."""
    lib = _lib.lib()
    device = pos.device
    if device.type != "cuda":
        raise _lib.GmeshError("deform_shade needs tensors on a HIP (cuda) device; there is no CPU path")
    tri = tri.detach().contiguous().to(torch.int32)
    w, dV, Rv, Sv, cov, pos, shs, campos = (_f(t) for t in (w, dV, Rv, Sv, cov, pos, shs, campos))
    N, M = pos.shape[0], shs.shape[1]
    f = dict(dtype=torch.float32, device=device)
    pos_o = torch.empty((N, 3), **f); cov6 = torch.empty((N, 6), **f); rgb = torch.empty((N, 3), **f)
    cov_o = torch.empty((N, 3, 3), **f) if want_cov_rot else None
    rot_o = torch.empty((N, 3, 3), **f) if want_cov_rot else None
    with torch.cuda.device(device):
        _lib.check(lib.gm_deform_shade(N, int(deg), M, tri.data_ptr(), w.data_ptr(), dV.data_ptr(), Rv.data_ptr(), Sv.data_ptr(),
                                       cov.data_ptr(), pos.data_ptr(), shs.data_ptr(), campos.data_ptr(), pos_o.data_ptr(),
                                       cov6.data_ptr(), rgb.data_ptr(), None if cov_o is None else cov_o.data_ptr(),
                                       None if rot_o is None else rot_o.data_ptr(), torch.cuda.current_stream(device).cuda_stream))
    return (pos_o, cov6, rgb, cov_o, rot_o) if want_cov_rot else (pos_o, cov6, rgb)


def pack_cov6(cov):
    """Rest covariances [N,3,3] (or [N,9]) -> [N,6] rows xx xy xz yy yz zz for forward_deformed_begin (GM_STREAM_COV6), or None when
    some matrix is not symmetric BIT FOR BIT: the fused pass reads the mirrored entries from the six, which reproduces the [N,9] call
    exactly only then (the reference multiplies with the full matrix, edittool/__init__.py:300-340)."""
    c = cov.detach().reshape(-1, 3, 3)
    if not (torch.equal(c[:, 0, 1], c[:, 1, 0]) and torch.equal(c[:, 0, 2], c[:, 2, 0]) and torch.equal(c[:, 1, 2], c[:, 2, 1])):
        return None
    return torch.stack((c[:, 0, 0], c[:, 0, 1], c[:, 0, 2], c[:, 1, 1], c[:, 1, 2], c[:, 2, 2]), dim=1).to(torch.float32).contiguous()


def pack_mesh_state(state, verts, out=None):
    """gm_pack_mesh_state: [Vm,21] frame state (V1 | R | S) and rest pose verts [Vm,3] -> gather table [Vm,24]."""
    lib = _lib.lib()
    device = state.device
    if device.type != "cuda":
        raise _lib.GmeshError("pack_mesh_state needs tensors on a HIP (cuda) device; there is no CPU path")
    state, verts = _f(state), _f(verts)
    Vm = state.shape[0]
    if state.shape[1] != 21 or verts.shape != (Vm, 3):
        raise ValueError("pack_mesh_state: state must be [Vm,21] and verts [Vm,3]")
    packed = out if out is not None else torch.empty((Vm, 24), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.gm_pack_mesh_state(Vm, state.data_ptr(), verts.data_ptr(), packed.data_ptr(),
                                          torch.cuda.current_stream(device).cuda_stream))
    return packed


def deform_shade_packed(tri, w, packed, cov, pos, shs, campos, deg=3, want_cov_rot=False):
    """gm_deform_shade_packed: deform_shade with the per-vertex (dV, R, S) read from pack_mesh_state()'s table."""
    lib = _lib.lib()
    device = pos.device
    if device.type != "cuda":
        raise _lib.GmeshError("deform_shade needs tensors on a HIP (cuda) device; there is no CPU path")
    tri = tri.detach().contiguous().to(torch.int32)
    w, packed, cov, pos, shs, campos = (_f(t) for t in (w, packed, cov, pos, shs, campos))
    N, M = pos.shape[0], shs.shape[1]
    f = dict(dtype=torch.float32, device=device)
    pos_o = torch.empty((N, 3), **f); cov6 = torch.empty((N, 6), **f); rgb = torch.empty((N, 3), **f)
    cov_o = torch.empty((N, 3, 3), **f) if want_cov_rot else None
    rot_o = torch.empty((N, 3, 3), **f) if want_cov_rot else None
    with torch.cuda.device(device):
        _lib.check(lib.gm_deform_shade_packed(N, int(deg), M, tri.data_ptr(), w.data_ptr(), packed.data_ptr(), cov.data_ptr(),
                                              pos.data_ptr(), shs.data_ptr(), campos.data_ptr(), pos_o.data_ptr(), cov6.data_ptr(),
                                              rgb.data_ptr(), None if cov_o is None else cov_o.data_ptr(),
                                              None if rot_o is None else rot_o.data_ptr(), torch.cuda.current_stream(device).cuda_stream))
    return (pos_o, cov6, rgb, cov_o, rot_o) if want_cov_rot else (pos_o, cov6, rgb)


def cov_to_scale_rot(cov):
    """gm_cov_to_scale_rot: (scales [N,3], rotations [N,4]) whose covariance R diag(s^2) R^T equals cov [N,3,3]
    (the SceneVisualTool route: edittool/__init__.py:204-207, rasterised with scales/rotations instead of cov3D_precomp)."""
    lib = _lib.lib()
    device = cov.device
    if device.type != "cuda":
        raise _lib.GmeshError("cov_to_scale_rot needs a tensor on a HIP (cuda) device; there is no CPU path")
    cov = _f(cov)
    N = cov.shape[0]
    scales = torch.empty((N, 3), dtype=torch.float32, device=device)
    rots = torch.empty((N, 4), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.gm_cov_to_scale_rot(N, cov.data_ptr(), scales.data_ptr(), rots.data_ptr(),
                                           torch.cuda.current_stream(device).cuda_stream))
    return scales, rots


def vertex_face_adjacency(faces, Vm):
    """CSR list of the faces incident to each vertex: (offsets int32 [Vm+1], face ids int32 [3F]), host side, once per mesh."""
    import numpy as np
    f = np.asarray(faces.detach().cpu() if hasattr(faces, "detach") else faces, dtype=np.int64).reshape(-1, 3)
    vid = f.reshape(-1)
    fid = np.repeat(np.arange(f.shape[0], dtype=np.int64), 3)
    order = np.argsort(vid, kind="stable")
    counts = np.bincount(vid, minlength=Vm)
    offsets = np.zeros(Vm + 1, np.int64)
    np.cumsum(counts, out=offsets[1:])
    return offsets.astype(np.int32), fid[order].astype(np.int32)


def mesh_rs_packed(rest_vertices, deformed_vertices, faces, adjacency, out=None):
    """gm_mesh_rs_packed: the per-vertex gather table [Vm,24] of the fused deformation kernels, straight from the deformed
    mesh (= pack_mesh_state(mesh_rs(..., want_state=True)[2], rest_vertices), bit for bit, in one launch)."""
    lib = _lib.lib()
    device = rest_vertices.device
    if device.type != "cuda":
        raise _lib.GmeshError("mesh_rs_packed needs tensors on a HIP (cuda) device; there is no CPU path")
    V0, V1 = _f(rest_vertices), _f(deformed_vertices)
    Vm = V0.shape[0]
    if faces.dtype is not torch.int32 or not faces.is_contiguous():
        faces = faces.detach().contiguous().to(torch.int32)
    off, adj = adjacency
    packed = out if out is not None else torch.empty((Vm, 24), dtype=torch.float32, device=device)
    from .rasterizer import _on, _stream
    with _on(device):
        _lib.check(lib.gm_mesh_rs_packed(Vm, faces.shape[0], V0.data_ptr(), V1.data_ptr(), faces.data_ptr(), off.data_ptr(), adj.data_ptr(),
                                         packed.data_ptr(), _stream(device)))
    return packed


def mesh_rs_packed_batch(rest_vertices, deformed_vertices_list, faces, adjacency, out=None):
    """gm_mesh_rs_packed_batch: the gather tables of up to GM_BATCH_MAX deformed meshes in ONE launch (the frames of a
    rasterizer.forward_deformed_batch); deformed_vertices_list: K tensors [Vm,3]; out: optional [K,Vm,24] tensor.  Returns the K tables
    (views of one [K,Vm,24] tensor), each bit for bit what mesh_rs_packed makes of its mesh."""
    lib = _lib.lib()
    device = rest_vertices.device
    if device.type != "cuda":
        raise _lib.GmeshError("mesh_rs_packed_batch needs tensors on a HIP (cuda) device; there is no CPU path")
    K = len(deformed_vertices_list)
    if not 1 <= K <= _lib.GM_BATCH_MAX:
        raise ValueError("mesh_rs_packed_batch: 1..%d frames" % _lib.GM_BATCH_MAX)
    V0 = _f(rest_vertices)
    V1 = [_f(v) for v in deformed_vertices_list]
    Vm = V0.shape[0]
    if faces.dtype is not torch.int32 or not faces.is_contiguous():
        faces = faces.detach().contiguous().to(torch.int32)
    off, adj = adjacency
    packed = out if out is not None else torch.empty((K, Vm, 24), dtype=torch.float32, device=device)
    import ctypes as C
    from .rasterizer import _on, _stream
    pv = (C.c_void_p * K)(*[v.data_ptr() for v in V1])
    pp = (C.c_void_p * K)(*[packed[k].data_ptr() for k in range(K)])
    with _on(device):
        _lib.check(lib.gm_mesh_rs_packed_batch(K, Vm, faces.shape[0], V0.data_ptr(), pv, faces.data_ptr(), off.data_ptr(), adj.data_ptr(), pp,
                                               _stream(device)))
    return [packed[k] for k in range(K)]


def mesh_rs(rest_vertices, deformed_vertices, faces, adjacency=None, want_state=False):
    """gm_mesh_rs: per-vertex (R, S) [Vm,3,3] of a deformed proxy mesh, the pair pyACAP.GetRS hands to
    SingleObjectDeform.deform_gaussian (edittool/__init__.py:109-113): cotangent-weighted one-ring least-squares
    deformation gradient per vertex (ACAP / ARAP), polar decomposition; R in pyACAP's row-vector convention (the transpose
    of the rotation).
    adjacency: (offsets, face ids) device int32 tensors from vertex_face_adjacency (built here when None).
    want_state: also return the frame record [Vm,21] = V1 | R | S that pack_mesh_state consumes."""
    lib = _lib.lib()
    device = rest_vertices.device
    if device.type != "cuda":
        raise _lib.GmeshError("mesh_rs needs tensors on a HIP (cuda) device; there is no CPU path")
    V0, V1 = _f(rest_vertices), _f(deformed_vertices)
    Vm = V0.shape[0]
    faces = faces.detach().contiguous().to(torch.int32)
    if adjacency is None:
        off, adj = vertex_face_adjacency(faces, Vm)
        adjacency = (torch.tensor(off, device=device), torch.tensor(adj, device=device))
    off, adj = adjacency
    R = torch.empty((Vm, 3, 3), dtype=torch.float32, device=device)
    S = torch.empty((Vm, 3, 3), dtype=torch.float32, device=device)
    state = torch.empty((Vm, 21), dtype=torch.float32, device=device) if want_state else None
    with torch.cuda.device(device):
        _lib.check(lib.gm_mesh_rs(Vm, faces.shape[0], V0.data_ptr(), V1.data_ptr(), faces.data_ptr(), off.data_ptr(), adj.data_ptr(),
                                  R.data_ptr(), S.data_ptr(), None if state is None else state.data_ptr(),
                                  torch.cuda.current_stream(device).cuda_stream))
    return (R, S, state) if want_state else (R, S)


class SingleObjectDeform:
    """Tensor-in counterpart of edittool.SingleObjectDeform.

    gaussian_pos [N,3], gaussian_cov [N,3,3], gaussian_o [N,1], gaussian_feature [N,16,3],
    gaussian_triangles int [N,3] (vertex ids of the bound face), weights [N,3] (barycentric, from
    get_barycentric_coordinate), vertex [Vm,3] rest vertices."""

    def __init__(self, gaussian_pos, gaussian_cov, gaussian_o, gaussian_feature, gaussian_triangles, weights, vertex,
                 name=None):
        self.name = name
        self.gaussian_pos = _f(gaussian_pos)
        self.gaussian_cov = _f(gaussian_cov)
        self.gaussian_o = _f(gaussian_o)
        self.gaussian_feature = _f(gaussian_feature)
        self.gaussian_triangles = gaussian_triangles.detach().contiguous().to(torch.int32)
        self.coord = _f(weights)
        self.weight_g_pos = self.coord.unsqueeze(2)
        self.weight_g_rs = self.coord.unsqueeze(2).unsqueeze(3)
        self.vertex = _f(vertex)
        self.number_gaussian = self.gaussian_pos.shape[0]
        self.gaussian_deform_pos = self.gaussian_pos
        self.gaussian_deform_cov = self.gaussian_cov
        self.gaussian_deform_rot = torch.eye(3, device=self.gaussian_pos.device).expand(self.number_gaussian, 3, 3).contiguous()
        self.gaussian_deform_cov6 = None

    def get_name(self):
        return self.name

    def deform(self, deform_vertex, cur_rot, cur_shear):
        dV = _f(deform_vertex) - self.vertex
        pos, cov, rot, cov6 = deform_tensors(self.gaussian_triangles, self.coord, dV, cur_rot.reshape(-1, 3, 3),
                                             cur_shear.reshape(-1, 3, 3), self.gaussian_cov, self.gaussian_pos)
        self.gaussian_deform_pos, self.gaussian_deform_cov, self.gaussian_deform_rot = pos, cov, rot
        self.gaussian_deform_cov6 = cov6
        return pos, cov, rot

    def deform_and_shade(self, deform_vertex, cur_rot, cur_shear, campos, deg=3):
        """One fused pass for the render loop: updates gaussian_deform_pos / gaussian_deform_cov6 and returns
        (means3D, colors_precomp, cov3D_precomp) for NewGaussianRasterizer (edittool/__init__.py:464-472)."""
        dV = _f(deform_vertex) - self.vertex
        pos, cov6, rgb = deform_shade(self.gaussian_triangles, self.coord, dV, cur_rot.reshape(-1, 3, 3), cur_shear.reshape(-1, 3, 3),
                                      self.gaussian_cov, self.gaussian_pos, self.gaussian_feature, campos, deg)
        self.gaussian_deform_pos, self.gaussian_deform_cov6 = pos, cov6
        return pos, rgb, cov6

    def deform_and_render(self, deform_vertex, cur_rot, cur_shear, viewpoint_camera, bg_color=None, workspace=None, begin_only=False):
        """The edit loop's frame in two enqueues (gm_forward_0_deformed_async + gm_forward_1_geom): deformation, rotated-
        direction SH colour and the rasterizer's preprocess run in one kernel, the deformed cloud is never written out.
        viewpoint_camera carries the reference's camera attributes (image_height/width, FoVx/FoVy, world_view_transform,
        full_proj_transform, camera_center).  Returns the image [3,H,W] (white background by default, as
        ObjectVisualTool.render_gaussian), or with begin_only the PendingForward handle for pipelined loops."""
        import math
        from . import rasterizer as Rz
        dev = self.gaussian_pos.device
        state = torch.cat([_f(deform_vertex), _f(cur_rot).reshape(-1, 9), _f(cur_shear).reshape(-1, 9)], dim=1)
        packed = pack_mesh_state(state, self.vertex)
        bg = torch.ones(3, device=dev) if bg_color is None else bg_color
        c = viewpoint_camera
        h = Rz.forward_deformed_begin(bg, self.gaussian_triangles, self.coord, packed, self.gaussian_cov, self.gaussian_pos,
                                      self.gaussian_feature, self.gaussian_o, c.world_view_transform, c.full_proj_transform,
                                      math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5), c.image_height, c.image_width, 3, c.camera_center,
                                      workspace=workspace)
        if begin_only:
            return h
        from .renderer import camera_work_hint
        return h.finish(image_only=True, work_hint=camera_work_hint(c, dev))[1]          # forward-only: no backward state
