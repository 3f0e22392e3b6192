"""On-disk formats either side of the hot path: the mesh-Gaussian PLY and cameras.json.

plyfile / igl are not available in this image, so this is a minimal reader/writer of exactly the format the reference
produces with plyfile: one `vertex` element, binary_little_endian 1.0, every property float32
(scene/mesh_based_gaussian_model.py:290-330).  Property order and names:
  x y z | nx ny nz | ca cb cc | v1x..v3z | dis | v_index1..3 | radius | face_id | f_dc_0..2 | f_rest_0..44 | opacity |
  scale_0..2 | rot_0..3
f_dc / f_rest are stored channel-major ([P,3,K] flattened) and transposed to [P,K,3] on load (:311-312, and
edittool/mesh_based_gaussian.py:160-186).  The edit tool's loader quirks are reproduced on request:
  * `_bc` is filled from x,y,z, not from ca,cb,cc (edittool/mesh_based_gaussian.py:183-184) -> bc_from_xyz=True
  * face_id is stored as float and cast to int (:196).
"""
import json
import math

import numpy as np

MESH_ATTRS = ['x', 'y', 'z', 'nx', 'ny', 'nz', 'ca', 'cb', 'cc', 'v1x', 'v1y', 'v1z', 'v2x', 'v2y', 'v2z', 'v3x', 'v3y', 'v3z',
              'dis', 'v_index1', 'v_index2', 'v_index3', 'radius', 'face_id']


def attribute_names(n_dc=3, n_rest=45, n_scale=3, n_rot=4):
    """construct_list_of_attributes, scene/mesh_based_gaussian_model.py:290-303."""
    return (MESH_ATTRS + ['f_dc_%d' % i for i in range(n_dc)] + ['f_rest_%d' % i for i in range(n_rest)] + ['opacity'] +
            ['scale_%d' % i for i in range(n_scale)] + ['rot_%d' % i for i in range(n_rot)])


def write_ply(path, columns):
    """columns: ordered dict name -> float array [P]; written as float32 little-endian."""
    names = list(columns)
    P = len(next(iter(columns.values())))
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    header += "".join("property float %s\n" % n for n in names) + "end_header\n"
    data = np.empty((P, len(names)), dtype="<f4")
    for j, n in enumerate(names):
        data[:, j] = np.asarray(columns[n], dtype=np.float32).reshape(P)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(data.tobytes())


def read_ply(path):
    """Returns (names, data[P, len(names)] float32) of the `vertex` element of a binary little-endian PLY with float props."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        names, P, fmt = [], None, None
        in_vertex = False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("unterminated PLY header")
            tok = line.decode("ascii").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    P = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] not in ("float", "float32"):
                    raise ValueError("only float32 vertex properties are supported (got %s)" % tok[1])
                names.append(tok[2])
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian":
            raise ValueError("only binary_little_endian PLY is supported (got %s)" % fmt)
        data = np.frombuffer(f.read(P * len(names) * 4), dtype="<f4").reshape(P, len(names))
    return names, np.array(data, dtype=np.float32)


def save_mesh_gaussians(path, m):
    """m: dict with xyz[P,3], normal[P,3], bc[P,3], v1,v2,v3[P,3], distance[P,1], vertex_index[P,3], radius[P,1], fid[P,1],
    features_dc[P,1,3], features_rest[P,15,3], opacity[P,1], scaling[P,3], rotation[P,4]  (save_ply, :305-328)."""
    f_dc = np.asarray(m["features_dc"]).transpose(0, 2, 1).reshape(len(m["xyz"]), -1)
    f_rest = np.asarray(m["features_rest"]).transpose(0, 2, 1).reshape(len(m["xyz"]), -1)
    attrs = np.concatenate([m["xyz"], m["normal"], m["bc"], m["v1"], m["v2"], m["v3"], m["distance"], m["vertex_index"], m["radius"],
                            m["fid"], f_dc, f_rest, m["opacity"], m["scaling"], m["rotation"]], axis=1)
    names = attribute_names(f_dc.shape[1], f_rest.shape[1], np.asarray(m["scaling"]).shape[1], np.asarray(m["rotation"]).shape[1])
    assert attrs.shape[1] == len(names)
    write_ply(path, {n: attrs[:, j] for j, n in enumerate(names)})


def load_mesh_gaussians(path, max_sh_degree=3, bc_from_xyz=False):
    """load_ply of the mesh-based model.  bc_from_xyz=True reproduces the edit tool (and train-time) loader, which
    fills _bc from the saved x,y,z (edittool/mesh_based_gaussian.py:183-184, scene/mesh_based_gaussian_model.py:392-393)."""
    names, d = read_ply(path)
    col = {n: d[:, j] for j, n in enumerate(names)}
    st = lambda *ks: np.stack([col[k] for k in ks], axis=1)
    xyz = st("x", "y", "z")
    rest = sorted([n for n in names if n.startswith("f_rest_")], key=lambda n: int(n.split("_")[-1]))
    assert len(rest) == 3 * (max_sh_degree + 1) ** 2 - 3
    f_rest = st(*rest).reshape(len(xyz), 3, (max_sh_degree + 1) ** 2 - 1).transpose(0, 2, 1)
    f_dc = st("f_dc_0", "f_dc_1", "f_dc_2").reshape(len(xyz), 3, 1).transpose(0, 2, 1)
    scales = st(*sorted([n for n in names if n.startswith("scale_")], key=lambda n: int(n.split("_")[-1])))
    rots = st(*sorted([n for n in names if n.startswith("rot")], key=lambda n: int(n.split("_")[-1])))
    return dict(xyz=xyz, load_xyz=xyz, bc=xyz.copy() if bc_from_xyz else st("ca", "cb", "cc"), normal=st("nx", "ny", "nz"),
                v1=st("v1x", "v1y", "v1z"), v2=st("v2x", "v2y", "v2z"), v3=st("v3x", "v3y", "v3z"), distance=col["dis"][:, None],
                vertex_index=st("v_index1", "v_index2", "v_index3"), radius=col["radius"][:, None],
                fid=col["face_id"][:, None].astype(np.int32), features_dc=np.ascontiguousarray(f_dc),
                features_rest=np.ascontiguousarray(f_rest), opacity=col["opacity"][:, None], scaling=scales, rotation=rots)


PLAIN_ATTRS = ['x', 'y', 'z', 'nx', 'ny', 'nz']


def save_plain_gaussians(path, m):
    """The free-standing (background) Gaussian PLY of the reference: x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_*
    (scene/gaussian_model.py save_ply; loaded by edittool/bg_gaussian.py:81-122).  m: xyz [P,3], features_dc [P,1,3],
    features_rest [P,15,3], opacity [P,1], scaling [P,3], rotation [P,4] (all raw, pre-activation)."""
    P = len(m["xyz"])
    f_dc = np.asarray(m["features_dc"]).transpose(0, 2, 1).reshape(P, -1)
    f_rest = np.asarray(m["features_rest"]).transpose(0, 2, 1).reshape(P, -1)
    attrs = np.concatenate([m["xyz"], np.zeros((P, 3)), f_dc, f_rest, m["opacity"], m["scaling"], m["rotation"]], axis=1)
    names = (PLAIN_ATTRS + ['f_dc_%d' % i for i in range(f_dc.shape[1])] + ['f_rest_%d' % i for i in range(f_rest.shape[1])] + ['opacity'] +
             ['scale_%d' % i for i in range(np.asarray(m["scaling"]).shape[1])] + ['rot_%d' % i for i in range(np.asarray(m["rotation"]).shape[1])])
    write_ply(path, {n: attrs[:, j] for j, n in enumerate(names)})


def load_plain_gaussians(path, max_sh_degree=3):
    """edittool/bg_gaussian.py:81-122 BGGaussianModel.load_ply."""
    names, d = read_ply(path)
    col = {n: d[:, j] for j, n in enumerate(names)}
    st = lambda *ks: np.stack([col[k] for k in ks], axis=1)
    xyz = st("x", "y", "z")
    rest = sorted([n for n in names if n.startswith("f_rest_")], key=lambda n: int(n.split("_")[-1]))
    assert len(rest) == 3 * (max_sh_degree + 1) ** 2 - 3
    f_rest = st(*rest).reshape(len(xyz), 3, (max_sh_degree + 1) ** 2 - 1).transpose(0, 2, 1)
    f_dc = st("f_dc_0", "f_dc_1", "f_dc_2").reshape(len(xyz), 3, 1).transpose(0, 2, 1)
    scales = st(*sorted([n for n in names if n.startswith("scale_")], key=lambda n: int(n.split("_")[-1])))
    rots = st(*sorted([n for n in names if n.startswith("rot")], key=lambda n: int(n.split("_")[-1])))
    return dict(xyz=xyz, features_dc=np.ascontiguousarray(f_dc), features_rest=np.ascontiguousarray(f_rest), opacity=col["opacity"][:, None],
                scaling=scales, rotation=rots)


# ----------------------------------------------------------------------------------------------
def read_obj(path):
    """Triangle mesh from a Wavefront OBJ (what igl.read_triangle_mesh returns for the reference's proxy meshes,
    edittool/__init__.py:66, 107): (vertices float64 [V,3], faces int32 [F,3]).  `v x y z` and `f a b c` records with
    optional /vt/vn suffixes and negative (relative) indices; polygons are fan-triangulated; everything else is ignored."""
    verts, faces = [], []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                verts.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif tok[0] == "f":
                idx = []
                for t in tok[1:]:
                    i = int(t.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    faces.append((idx[0], idx[k], idx[k + 1]))
    return np.asarray(verts, np.float64).reshape(-1, 3), np.asarray(faces, np.int32).reshape(-1, 3)


def write_obj(path, vertices, faces):
    with open(path, "w") as f:
        for v in np.asarray(vertices, np.float64):
            f.write("v %.17g %.17g %.17g\n" % (v[0], v[1], v[2]))
        for t in np.asarray(faces, np.int64):
            f.write("f %d %d %d\n" % (t[0] + 1, t[1] + 1, t[2] + 1))


def save_image(img, path):
    """[3,H,W] float image in [0,1] -> 8-bit PNG (the missing render_origin.save_image of edit.py:13)."""
    from PIL import Image
    a = np.asarray(img.detach().cpu() if hasattr(img, "detach") else img, np.float32)
    a = (np.clip(a, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8).transpose(1, 2, 0)
    Image.fromarray(a).save(path)


# ----------------------------------------------------------------------------------------------
def camera_to_json(cam_id, R, T, width, height, fovx, fovy, img_name=""):
    """utils/camera_utils.py:63-83 camera_to_JSON (R = camera-to-world rotation, T = world-to-camera translation)."""
    Rt = np.zeros((4, 4)); Rt[:3, :3] = np.asarray(R).transpose(); Rt[:3, 3] = T; Rt[3, 3] = 1.0
    W2C = np.linalg.inv(Rt)
    return {"id": cam_id, "img_name": img_name, "width": width, "height": height, "position": W2C[:3, 3].tolist(),
            "rotation": [x.tolist() for x in W2C[:3, :3]], "fy": height / (2 * math.tan(fovy / 2)), "fx": width / (2 * math.tan(fovx / 2))}


def load_cameras_json(path):
    """edittool/__init__.py:547-584 get_camera: cameras.json -> list of camera dicts (scenes.camera_from_RT layout)."""
    from . import scenes
    with open(path) as f:
        entries = json.load(f)
    cams = []
    for e in entries:
        W2C = np.zeros((4, 4)); W2C[:3, :3] = np.array(e["rotation"]); W2C[:3, 3] = np.array(e["position"]); W2C[3, 3] = 1
        Rt = np.linalg.inv(W2C)
        T, R = Rt[:3, 3], Rt[:3, :3].transpose()
        fovy = 2 * math.atan(e["height"] / (2 * e["fy"])); fovx = 2 * math.atan(e["width"] / (2 * e["fx"]))
        c = scenes.camera_from_RT(R, T, fovx, fovy, e["width"], e["height"])
        c["img_name"] = e.get("img_name", ""); c["id"] = e.get("id", len(cams))
        cams.append(c)
    return cams
