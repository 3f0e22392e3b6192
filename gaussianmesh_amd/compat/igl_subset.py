"""The four libigl python calls the in-scope reference code makes, on numpy:

    igl.read_triangle_mesh(path)                 edittool/__init__.py:66, 107; scene/mesh_based_gaussian_model.py:188
    igl.write_triangle_mesh(path, V, F)          scene/mesh_based_gaussian_model.py:594
    igl.per_face_normals(V, F, Z)                scene/mesh_based_gaussian_model.py:193
    igl.point_mesh_squared_distance(P, V, F)     edittool/__init__.py:80

libigl itself is a third-party dependency that is not installed in this image (pip package `libigl`; the reference pins no
version).  These follow its documented behaviour: meshes as (V float64 [n,3], F int [m,3]); per_face_normals returns the unit
normal of each face, and Z for degenerate faces; point_mesh_squared_distance returns (sqrD, I, C).  Registered as module `igl`
by compat.install(edit_tool=True) only when no real igl is importable.  Anything else raises AttributeError naming the symbol."""
import os

import numpy as np

__gaussianmesh_compat__ = True


def read_triangle_mesh(path, dtypef=np.float64):
    """(V, F) of an .obj or .off file (the formats of the reference's proxy meshes and of its mesh_preprocess tools)."""
    from .. import io as gio
    ext = os.path.splitext(str(path))[1].lower()
    if ext == ".obj":
        v, f = gio.read_obj(path)
    elif ext == ".off":
        with open(path) as fh:
            tok = fh.read().split()
        if not tok or tok[0] != "OFF":
            raise ValueError("%s: not an OFF file" % path)
        nv, nf = int(tok[1]), int(tok[2])
        v = np.asarray(tok[4:4 + 3 * nv], np.float64).reshape(nv, 3)
        faces, k = [], 4 + 3 * nv
        for _ in range(nf):
            n = int(tok[k]); idx = [int(t) for t in tok[k + 1:k + 1 + n]]; k += 1 + n
            faces += [(idx[0], idx[j], idx[j + 1]) for j in range(1, n - 1)]
        f = np.asarray(faces, np.int32).reshape(-1, 3)
    else:
        raise ValueError("igl subset: read_triangle_mesh supports .obj and .off (got %r)" % ext)
    return np.asarray(v, dtypef), f


def write_triangle_mesh(path, v, f, **_kw):
    from .. import io as gio
    ext = os.path.splitext(str(path))[1].lower()
    if ext == ".obj":
        gio.write_obj(path, v, f)
    elif ext == ".off":
        v = np.asarray(v, np.float64); f = np.asarray(f, np.int64)
        with open(path, "w") as fh:
            fh.write("OFF\n%d %d 0\n" % (len(v), len(f)))
            fh.writelines("%.17g %.17g %.17g\n" % tuple(r) for r in v)
            fh.writelines("3 %d %d %d\n" % tuple(r) for r in f)
    else:
        raise ValueError("igl subset: write_triangle_mesh supports .obj and .off (got %r)" % ext)
    return True


def per_face_normals(v, f, z):
    """Unit normal (v1 - v0) x (v2 - v0) of every face; the row `z` where the cross product vanishes."""
    v = np.asarray(v, np.float64); f = np.asarray(f, np.int64)
    n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    ok = ln[:, 0] > 0
    out = np.tile(np.asarray(z, np.float64).reshape(1, 3), (len(f), 1))
    out[ok] = n[ok] / ln[ok]
    return out


def point_mesh_squared_distance(p, v, f):
    from ..edittool import point_mesh_squared_distance as pmsd
    sqr, idx, close = pmsd(p, v, f)
    return sqr, idx.astype(np.int32), close


def __getattr__(name):
    raise AttributeError("gaussianmesh_amd's igl subset does not provide %r (only read_triangle_mesh, write_triangle_mesh, "
                         "per_face_normals, point_mesh_squared_distance: what the in-scope reference code calls)" % name)
