"""8f-2: mesh-Gaussian PLY schema and cameras.json (host side, no GPU)."""
import json

import numpy as np


def _model(P=37, seed=0):
    r = np.random.default_rng(seed)
    f = lambda *s: r.normal(size=s).astype(np.float32)
    return dict(xyz=f(P, 3), normal=f(P, 3), bc=f(P, 3), v1=f(P, 3), v2=f(P, 3), v3=f(P, 3), distance=f(P, 1),
                vertex_index=r.integers(0, 100, (P, 3)).astype(np.float32), radius=np.abs(f(P, 1)),
                fid=r.integers(0, 50, (P, 1)).astype(np.float32), features_dc=f(P, 1, 3), features_rest=f(P, 15, 3),
                opacity=f(P, 1), scaling=f(P, 3), rotation=f(P, 4))


def test_ply_schema_and_roundtrip(tmp_path):
    from gaussianmesh_amd import io
    m = _model()
    p = tmp_path / "point_cloud.ply"
    io.save_mesh_gaussians(str(p), m)
    head = p.read_bytes().split(b"end_header\n")[0].decode()
    lines = head.splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    props = [l.split()[-1] for l in lines if l.startswith("property")]
    assert all(l.startswith("property float ") for l in lines[3:])
    # scene/mesh_based_gaussian_model.py:290-303
    assert props[:24] == ['x', 'y', 'z', 'nx', 'ny', 'nz', 'ca', 'cb', 'cc', 'v1x', 'v1y', 'v1z', 'v2x', 'v2y', 'v2z', 'v3x', 'v3y',
                          'v3z', 'dis', 'v_index1', 'v_index2', 'v_index3', 'radius', 'face_id']
    assert props[24:27] == ['f_dc_0', 'f_dc_1', 'f_dc_2'] and props[27] == 'f_rest_0' and props[71] == 'f_rest_44'
    assert props[72:] == ['opacity', 'scale_0', 'scale_1', 'scale_2', 'rot_0', 'rot_1', 'rot_2', 'rot_3'] and len(props) == 80
    back = io.load_mesh_gaussians(str(p))
    for k in ("xyz", "bc", "normal", "v1", "v2", "v3", "distance", "radius", "features_dc", "features_rest", "opacity", "scaling", "rotation"):
        assert np.array_equal(back[k], m[k]), k
    assert back["fid"].dtype == np.int32 and np.array_equal(back["fid"], m["fid"].astype(np.int32))
    # f_rest is stored channel-major: f_rest_0..14 are channel 0 of coefficients 1..15
    names, d = io.read_ply(str(p))
    assert np.array_equal(d[:, names.index("f_rest_1")], m["features_rest"][:, 1, 0])
    assert np.array_equal(d[:, names.index("f_rest_15")], m["features_rest"][:, 0, 1])
    # loader quirk of the edit tool: _bc <- x,y,z
    q = io.load_mesh_gaussians(str(p), bc_from_xyz=True)
    assert np.array_equal(q["bc"], m["xyz"]) and not np.array_equal(q["bc"], m["bc"])


def test_cameras_json_roundtrip(tmp_path):
    from gaussianmesh_amd import io, scenes
    cams, entries = [], []
    for k in range(3):
        c = scenes.orbit_camera(k, 3, 640, 360)
        # recover (R, T) the way scene/cameras.py stores them: view = getWorld2View2(R, T).T
        Rt = c["view"].T.astype(np.float64)
        R, T = Rt[:3, :3].T, Rt[:3, 3]
        entries.append(io.camera_to_json(k, R, T, 640, 360, c["fovx"], c["fovy"], "img%d" % k))
        cams.append(c)
    p = tmp_path / "cameras.json"
    p.write_text(json.dumps(entries))
    assert set(entries[0]) == {"id", "img_name", "width", "height", "position", "rotation", "fy", "fx"}
    assert np.allclose(entries[1]["position"], cams[1]["campos"], atol=1e-5)        # position = camera centre
    back = io.load_cameras_json(str(p))
    for a, b in zip(back, cams):
        assert np.allclose(a["view"], b["view"], atol=1e-5) and np.allclose(a["proj"], b["proj"], atol=1e-4)
        assert np.allclose(a["campos"], b["campos"], atol=1e-5) and abs(a["tanx"] - b["tanx"]) < 1e-6


def test_obj_reader_writer(tmp_path):
    from gaussianmesh_amd import io as gio, scenes
    verts, faces = scenes.torus_mesh(12, 8)
    p = str(tmp_path / "m.obj")
    gio.write_obj(p, verts, faces)
    v, f = gio.read_obj(p)
    assert np.array_equal(v, verts) and np.array_equal(f, faces)                # %.17g round-trips float64
    q = tmp_path / "q.obj"
    q.write_text("# comment\nvn 0 0 1\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nf 1/1/1 2/1/1 3/1/1 4/1/1\nf -4//1 -3//1 -1//1\n")
    v, f = gio.read_obj(str(q))
    assert v.shape == (4, 3) and np.array_equal(f, [[0, 1, 2], [0, 2, 3], [0, 1, 3]])     # quad fanned, v/vt/vn and negative indices


def test_plain_gaussian_ply_roundtrip_and_png(tmp_path):
    from gaussianmesh_amd import io as gio
    rng = np.random.default_rng(0)
    P = 37
    m = dict(xyz=rng.normal(size=(P, 3)), features_dc=rng.normal(size=(P, 1, 3)), features_rest=rng.normal(size=(P, 15, 3)),
             opacity=rng.normal(size=(P, 1)), scaling=rng.normal(size=(P, 3)), rotation=rng.normal(size=(P, 4)))
    p = str(tmp_path / "bg.ply")
    gio.save_plain_gaussians(p, m)
    names, _ = gio.read_ply(p)
    assert names[:6] == ['x', 'y', 'z', 'nx', 'ny', 'nz'] and names[6] == 'f_dc_0' and names[-1] == 'rot_3' and len(names) == 62
    g = gio.load_plain_gaussians(p)
    for k in m:
        assert np.array_equal(g[k], np.asarray(m[k], np.float32)), k
    img = np.zeros((3, 5, 7), np.float32); img[0, 2, 3] = 1.0; img[1] = 0.5
    gio.save_image(img, str(tmp_path / "a.png"))
    from PIL import Image
    a = np.asarray(Image.open(str(tmp_path / "a.png")))
    assert a.shape == (5, 7, 3) and a[2, 3, 0] == 255 and a[0, 0, 1] == 128 and a[0, 0, 2] == 0


def test_closest_triangles_matches_point_by_point_search():
    """edittool.closest_triangles (stand-in for igl.point_mesh_squared_distance's face index) against an independent
    evaluation: squared distance to every triangle from a dense barycentric sampling + the exact plane/edge/vertex cases."""
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.edittool import closest_triangles
    rng = np.random.default_rng(1)
    verts, faces = scenes.torus_mesh(10, 7)
    pts = verts[rng.integers(len(verts), size=60)] * rng.uniform(0.6, 1.4, size=(60, 1)) + 0.05 * rng.normal(size=(60, 3))
    got = closest_triangles(pts, verts, faces, chunk=16)
    u = np.linspace(0, 1, 41)
    bu, bv = np.meshgrid(u, u, indexing="ij")
    keep = bu + bv <= 1 + 1e-12
    bu, bv = bu[keep], bv[keep]
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    samples = a[:, None] + bu[None, :, None] * (b - a)[:, None] + bv[None, :, None] * (c - a)[:, None]       # [F, S, 3]
    for i, p in enumerate(pts):
        d = ((samples - p) ** 2).sum(-1).min(1)                     # sampled distance to each triangle (upper bound, close)
        exact_got = d[got[i]]
        assert exact_got <= d.min() * 1.02 + 1e-4, i               # the chosen triangle is (one of) the closest


# ---- fixtures generated by EXECUTING the reference's own code (tests/golden/make_golden_model.py) -----------------------------
import os

_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_cameras_json_written_by_the_reference_serialiser():
    """tests/golden/cameras.json is the output of utils/camera_utils.py:63-83 camera_to_JSON on three scene/cameras.py Camera
    objects; cameras_expected.npz holds those cameras' (R, T, FoV) and the matrices the reference's Camera class derived from
    them.  io.camera_to_json must write the same entries, io.load_cameras_json (edittool/__init__.py:547-584) must recover the
    cameras."""
    from gaussianmesh_amd import io
    exp = np.load(os.path.join(_GOLD, "cameras_expected.npz"))
    with open(os.path.join(_GOLD, "cameras.json")) as f:
        entries = json.load(f)
    assert len(entries) == 3
    for k, e in enumerate(entries):
        mine = io.camera_to_json(k, exp["R"][k], exp["T"][k], int(exp["W"][k]), int(exp["H"][k]), float(exp["FoVx"][k]), float(exp["FoVy"][k]),
                                 e["img_name"])
        assert set(mine) == set(e) and mine["id"] == e["id"] and mine["width"] == e["width"] and mine["height"] == e["height"]
        assert mine["img_name"] == e["img_name"]
        assert np.allclose(mine["position"], e["position"], rtol=0, atol=1e-12) and np.allclose(mine["rotation"], e["rotation"], rtol=0, atol=1e-12)
        assert abs(mine["fx"] - e["fx"]) <= 1e-9 * e["fx"] and abs(mine["fy"] - e["fy"]) <= 1e-9 * e["fy"]
    cams = io.load_cameras_json(os.path.join(_GOLD, "cameras.json"))
    for k, c in enumerate(cams):
        assert (c["W"], c["H"]) == (int(exp["W"][k]), int(exp["H"][k])) and c["img_name"] == entries[k]["img_name"]
        assert abs(c["fovx"] - exp["FoVx"][k]) < 1e-9 and abs(c["fovy"] - exp["FoVy"][k]) < 1e-9
        # the matrices of the reference's Camera class (world_view_transform, full_proj_transform, camera_center), float32
        assert np.allclose(c["view"], exp["view"][k], atol=2e-6), k
        assert np.allclose(c["proj"], exp["proj"][k], rtol=1e-5, atol=1e-5), k
        assert np.allclose(c["campos"], exp["center"][k], atol=1e-5), k


def test_mesh_ply_rows_assembled_by_the_reference_writer(tmp_path):
    """tests/golden/mesh_ply.npz: `names` = construct_list_of_attributes() (scene/mesh_based_gaussian_model.py:290-303),
    `elements` = the structured rows save_ply (:305-330) hands to plyfile for the model stored next to them, `loaded_*` = what
    load_ply (:341-408) makes of those columns."""
    from gaussianmesh_amd import io
    fx = np.load(os.path.join(_GOLD, "mesh_ply.npz"))
    names = [str(n) for n in fx["names"]]
    assert io.attribute_names() == names and len(names) == 80
    m = {k: fx[k] for k in ("xyz", "normal", "bc", "v1", "v2", "v3", "distance", "vertex_index", "radius", "fid", "features_dc", "features_rest",
                             "opacity", "scaling", "rotation")}
    p = tmp_path / "point_cloud.ply"
    io.save_mesh_gaussians(str(p), m)
    got_names, data = io.read_ply(str(p))
    assert got_names == names and np.array_equal(data, fx["elements"])
    back = io.load_mesh_gaussians(str(p), bc_from_xyz=True)            # the reference loader fills _bc from x, y, z (:392-393)
    for k in ("bc", "features_dc", "features_rest", "opacity", "scaling", "rotation", "distance", "v1", "v2", "v3", "normal", "radius", "load_xyz"):
        assert np.array_equal(back[k], fx["loaded_" + k]), k
    assert np.array_equal(back["fid"].astype(np.float32).reshape(-1), fx["loaded_fid"].astype(np.float32).reshape(-1))
