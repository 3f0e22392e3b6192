"""CPU restatement (numpy, float64) of gm_mesh_rs.  TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).

PARITY STATUS: "parity unpinned".  The reference obtains the per-vertex (R, S) pair from pyACAP.GetRS
(edittool/__init__.py:102, 109); pyACAP is a binary from ACAP/pyACAPv1.zip that is not in the reference tree
(.MISSING_LARGE_BLOBS), has no pinned version and no test vectors.  Restated from the published algorithm (Gao et al.,
"Sparse Data Driven Mesh Deformation": per vertex the cotangent-weighted least-squares affine map of the one-ring edges,
T_i = argmin sum_j c_ij |e'_ij - T e_ij|^2, then the polar decomposition T = Q S) and anchored on the call site
(deform_gaussian transposes the blended R and transforms covariances by R^T S, :118-129, so R = Q^T).
Written independently of the HIP kernel's algebra (per-edge scatter + numpy solve + SVD instead of the per-vertex CSR loop,
adjugate inverse and Jacobi eigen-solver); conditioning constants as documented in gm_mesh.hip."""
import numpy as np


def mesh_rs(V0, V1, faces):
    V0 = np.asarray(V0, np.float64); V1 = np.asarray(V1, np.float64); faces = np.asarray(faces, np.int64)
    Vm = V0.shape[0]
    M0 = np.zeros((Vm, 3, 3)); M1 = np.zeros((Vm, 3, 3)); nr = np.zeros((Vm, 3)); nd = np.zeros((Vm, 3)); wsum = np.zeros(Vm)
    for c0 in range(3):                                           # every face, seen from each of its corners
        v, a, b = faces[:, c0], faces[:, (c0 + 1) % 3], faces[:, (c0 + 2) % 3]
        ea, eb, da, db = V0[a] - V0[v], V0[b] - V0[v], V1[a] - V1[v], V1[b] - V1[v]
        ab = eb - ea
        n0, n1 = np.cross(ea, eb), np.cross(da, db)
        l0 = np.linalg.norm(n0, axis=1)
        ok = l0 > 1e-30
        with np.errstate(divide="ignore", invalid="ignore"):
            cot_a = -(ea * ab).sum(1) / l0                        # angle at a, opposite edge v-b
            cot_b = (eb * ab).sum(1) / l0                         # angle at b, opposite edge v-a
        wa = np.where(ok, np.maximum(0.5 * cot_b, 1e-3), 0.0); wb = np.where(ok, np.maximum(0.5 * cot_a, 1e-3), 0.0)
        np.add.at(M0, v, wa[:, None, None] * ea[:, :, None] * ea[:, None, :] + wb[:, None, None] * eb[:, :, None] * eb[:, None, :])
        np.add.at(M1, v, wa[:, None, None] * da[:, :, None] * ea[:, None, :] + wb[:, None, None] * db[:, :, None] * eb[:, None, :])
        np.add.at(nr, v, np.where(ok[:, None], n0, 0.0)); np.add.at(nd, v, np.where(ok[:, None], n1, 0.0))
        np.add.at(wsum, v, np.where(ok, l0, 0.0))
    R = np.tile(np.eye(3), (Vm, 1, 1)); S = np.tile(np.eye(3), (Vm, 1, 1))
    lr, ld = np.linalg.norm(nr, axis=1), np.linalg.norm(nd, axis=1)
    reg = (lr > 1e-30) & (ld > 1e-30)
    lam = 1e-9 * np.trace(M0, axis1=1, axis2=2)
    nru = nr / np.maximum(lr, 1e-300)[:, None]; ndu = nd / np.maximum(ld, 1e-300)[:, None]
    sc = np.sqrt(np.where(reg, ld / np.maximum(lr, 1e-300), 1.0))
    M0 = M0 + np.where(reg, lam, 0.0)[:, None, None] * nru[:, :, None] * nru[:, None, :]
    M1 = M1 + np.where(reg, lam * sc, 0.0)[:, None, None] * ndu[:, :, None] * nru[:, None, :]
    has = (wsum > 0) & (np.abs(np.linalg.det(M0)) > 1e-300)
    F = M1[has] @ np.linalg.inv(M0[has])
    U, sig, Vt = np.linalg.svd(F)
    neg = np.linalg.det(F) < 0
    sig = sig.copy(); sig[neg, 2] *= -1.0                         # the smallest stretch takes the sign of a reflection
    Q = np.einsum("nik,nk,nkj->nij", U, np.where(neg[:, None], np.array([1.0, 1.0, -1.0]), 1.0), Vt)
    S[has] = np.einsum("nki,nk,nkj->nij", Vt, sig, Vt)
    R[has] = Q.transpose(0, 2, 1)
    return R, S
