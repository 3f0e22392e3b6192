"""-m gpu: gm_mesh_rs (the (R, S) producer that replaces pyACAP.GetRS, edittool/__init__.py:102, 109) against its numpy
oracle and the invariants of the construction; then through the Gaussian deformation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rs(V0, V1, faces, **kw):
    from gpu_utils import T
    from gaussianmesh_amd.deform import mesh_rs
    out = mesh_rs(T(V0), T(V1), T(faces, dtype=torch.int32), **kw)
    return [x.cpu().numpy().astype(np.float64) for x in out]


def _rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def test_invariants_identity_rigid_scale_affine():
    from gaussianmesh_amd import scenes
    verts, faces = scenes.torus_mesh(30, 20)
    I = np.eye(3)
    R, S = _rs(verts, verts, faces)
    assert np.abs(R - I).max() <= 1e-6 and np.abs(S - I).max() <= 1e-6                       # identity -> (I, I)
    Q = _rot([1, 2, -0.5], 0.9)
    R, S = _rs(verts, verts @ Q.T + np.array([0.3, -2.0, 5.0]), faces)
    assert np.abs(R - Q.T).max() <= 1e-5 and np.abs(S - I).max() <= 1e-5                     # rigid motion -> (rotation^T, I)
    R, S = _rs(verts, 1.7 * verts, faces)
    assert np.abs(R - I).max() <= 1e-5 and np.abs(S - 1.7 * I).max() <= 1e-5                 # uniform scale s -> (I, s I)
    # a general affine map A (rotation x shear x anisotropic scale): the one-ring fit reproduces it, R^T S = A
    A = _rot([0.2, 1, 0.4], 0.7) @ np.array([[1.4, 0.3, 0.1], [0.1, 0.8, -0.2], [0.0, 0.25, 1.1]])
    R, S = _rs(verts, verts @ A.T + 0.5, faces)
    Fm = np.einsum("nji,njk->nik", R, S)                                                      # R^T S
    assert np.abs(Fm - A[None]).max() <= 2e-4        # float32 vertices; the normal component rests on the one-ring's O(h^2) bulge
    assert np.abs(np.linalg.det(R) - 1).max() <= 1e-5 and np.abs(S - np.transpose(S, (0, 2, 1))).max() <= 1e-6
    # a FLAT patch (planar one-rings): in-plane exact, the normal direction follows the deformed normal
    g = np.stack(np.meshgrid(np.linspace(-1, 1, 12), np.linspace(-1, 1, 9), indexing="ij"), -1).reshape(-1, 2)
    flat = np.concatenate([g, np.zeros((g.shape[0], 1))], 1)
    idx = np.arange(12 * 9).reshape(12, 9)
    ff = np.concatenate([np.stack([idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:]], -1).reshape(-1, 3),
                         np.stack([idx[:-1, :-1], idx[1:, 1:], idx[:-1, 1:]], -1).reshape(-1, 3)], 0).astype(np.int32)
    R, S = _rs(flat, flat @ A.T, ff)
    Fm = np.einsum("nji,njk->nik", R, S)
    tang = np.array([[1, 0, 0], [0, 1, 0]], float).T
    assert np.abs(Fm @ tang - (A @ tang)[None]).max() <= 2e-5
    nA = np.cross(A[:, 0], A[:, 1]); nA /= np.linalg.norm(nA)
    got_n = Fm @ np.array([0, 0, 1.0])
    assert np.abs(got_n / np.linalg.norm(got_n, axis=1, keepdims=True) - nA).max() <= 1e-4
    assert np.isfinite(R).all() and np.isfinite(S).all()


def test_matches_numpy_oracle_on_random_deformations():
    from gaussianmesh_amd import scenes
    from oracle import mesh_oracle
    rng = np.random.default_rng(2)
    verts, faces = scenes.torus_mesh(40, 24)
    V1 = verts @ _rot([0, 1, 1], 0.4).T * np.array([1.3, 0.8, 1.1]) + 0.05 * rng.normal(size=verts.shape)
    verts32, V132 = verts.astype(np.float32), V1.astype(np.float32)
    R, S, state = _rs(verts32, V132, faces, want_state=True)
    Ro, So = mesh_oracle.mesh_rs(verts32, V132, faces)
    assert np.abs(R - Ro).max() <= 2e-5 and np.abs(S - So).max() <= 2e-5
    assert np.array_equal(state[:, :3], V132.astype(np.float64)) and np.array_equal(state[:, 3:12], R.reshape(-1, 9))
    assert np.array_equal(state[:, 12:], S.reshape(-1, 9))
    # the one-launch variant writes the deformation kernels' gather table directly: identical to state -> gm_pack_mesh_state
    from gpu_utils import T
    from gaussianmesh_amd.deform import mesh_rs, mesh_rs_packed, pack_mesh_state, vertex_face_adjacency
    off, adj = vertex_face_adjacency(faces, verts32.shape[0])
    adjacency = (torch.tensor(off, device="cuda"), torch.tensor(adj, device="cuda"))
    ft = T(faces, dtype=torch.int32)
    st = mesh_rs(T(verts32), T(V132), ft, adjacency=adjacency, want_state=True)[2]
    assert torch.equal(mesh_rs_packed(T(verts32), T(V132), ft, adjacency), pack_mesh_state(st, T(verts32)))
    # a reflected neighbourhood (det F < 0) still gives a proper rotation, and degenerate faces / isolated vertices are inert
    V2 = V132.copy(); V2[:, 0] *= -1
    R2, S2 = _rs(verts32, V2, faces)
    assert np.abs(np.linalg.det(R2) - 1).max() <= 1e-4
    assert np.abs(np.einsum("nji,njk->nik", R2, S2) - np.einsum("nji,njk->nik", *mesh_oracle.mesh_rs(verts32, V2, faces))).max() <= 5e-5
    f_deg = np.concatenate([faces, np.array([[0, 0, 1]], np.int32)], 0)
    v_iso = np.concatenate([verts32, np.array([[9, 9, 9]], np.float32)], 0)
    R3, S3 = _rs(v_iso, np.concatenate([V132, [[1, 2, 3]]], 0).astype(np.float32), f_deg)
    assert np.abs(R3[:-1] - R).max() <= 1e-6 and np.array_equal(R3[-1], np.eye(3)) and np.array_equal(S3[-1], np.eye(3))


def test_converges_to_the_analytic_twist_and_drives_the_deformation():
    """The per-vertex map from the mesh converges (first order in the edge length) to the analytic Jacobian of
    scenes.twist_bend_frame on tangent vectors as the mesh is refined; and the edit step from vertices alone (gm_mesh_rs
    feeding gm_deform) equals the oracle chain."""
    from gpu_utils import T
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.deform import deform_tensors
    errs = []
    for nu, nv in ((50, 38), (100, 75), (200, 150)):
        verts, faces = scenes.torus_mesh(nu, nv)
        V1, Ra, Sa = scenes.twist_bend_frame(verts, t=11)
        R, S = _rs(verts.astype(np.float32), V1.astype(np.float32), faces)
        # compare the action on tangent vectors: a surface cannot tell how the analytic map stretches along the normal
        tri = faces[:: max(1, faces.shape[0] // 4000)]
        v = tri[:, 0]
        e = verts[tri[:, 1]] - verts[v]
        e /= np.linalg.norm(e, axis=1, keepdims=True)
        got = np.einsum("nji,njk,nk->ni", R[v], S[v], e)            # R^T S: the one-ring affine map
        want = np.einsum("nji,njk,nk->ni", Ra[v], Sa[v], e)         # the analytic Jacobian
        errs.append(np.abs(got - want).max())
    assert errs[1] < 0.6 * errs[0] and errs[2] < 0.6 * errs[1] and errs[2] <= 2e-2, errs
    # the whole edit step from vertices alone: gm_mesh_rs -> gm_deform on the GPU == oracle mesh_rs -> oracle deform
    from oracle import mesh_oracle, oracle as orc
    verts, faces = scenes.torus_mesh(100, 75)
    V1, _, _ = scenes.twist_bend_frame(verts, t=11)
    verts32, V132 = verts.astype(np.float32), V1.astype(np.float32)
    cl = scenes.bind_cloud_to_mesh(4000, verts, faces, seed=3)
    cov = scenes.cov3d_from_scale_rot(cl["scales"], cl["rots"]).astype(np.float32)
    R, S = _rs(verts32, V132, faces)
    p1, c1, r1, _ = deform_tensors(T(cl["tri"], dtype=torch.int32), T(cl["weights"]), T(V132 - verts32), T(R), T(S), T(cov), T(cl["means"]))
    Ro, So = mesh_oracle.mesh_rs(verts32, V132, faces)
    p_ref, c_ref, r_ref = orc.deform(cl["tri"], cl["weights"], V132 - verts32, Ro.astype(np.float32), So.astype(np.float32), cov, cl["means"])
    assert np.abs(p1.cpu().numpy() - p_ref).max() <= 1e-5 * np.abs(p_ref).max()
    assert np.abs(c1.cpu().numpy() - c_ref).max() <= 1e-4 * np.abs(c_ref).max()
    assert np.abs(r1.cpu().numpy() - r_ref).max() <= 1e-4
