"""-m gpu: the drop-in boundary used the way the reference uses its own library - from C++, with no Python and no torch on the
path.  tests/native/abi_caller.cpp (HIP runtime for device memory + include/gmesh_hip.h, nothing else) is compiled on the box,
runs the reference bridge's call sequence (buffers sized by gm_*_bytes, gm_forward_0 -> num_rendered -> gm_forward_1 ->
gm_backward) on a scene written as raw files, and what it writes back is compared with the oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import assert_forward_gate, assert_grads_elementwise, small_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc on this box")
def test_cpp_host_program_through_the_c_abi(tmp_path, oracle):
    exe = tmp_path / "abi_caller"
    csrc = os.path.join(ROOT, "gaussianmesh_amd", "csrc")
    build = subprocess.run([HIPCC, "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "abi_caller.cpp"),
                            "-L", csrc, "-lgmesh_hip", "-Wl,-rpath," + csrc, "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    D, M = 3, 16
    sc, cam = small_scene(P=1500, W=160, H=96, seed=3, D=D)
    bg = np.array([0.1, 0.3, 0.6], np.float32)
    dpix = np.random.default_rng(4).normal(size=(3, cam["H"], cam["W"])).astype(np.float32)
    d = tmp_path / "scene"
    d.mkdir()
    w = lambda name, a, dt=np.float32: np.ascontiguousarray(a, dtype=dt).tofile(str(d / name))
    w("meta.bin", [sc["means"].shape[0], cam["W"], cam["H"], D, M], np.int32)
    w("camera.bin", np.concatenate([np.asarray(cam["view"], np.float32).ravel(), np.asarray(cam["proj"], np.float32).ravel(),
                                    np.asarray(cam["campos"], np.float32).ravel(), [cam["tanx"], cam["tany"]], bg]))
    for name in ("means", "shs", "opac", "scales", "rots"):
        w(name + ".bin", sc[name])
    w("dpix.bin", dpix)
    run = subprocess.run([str(exe), str(d)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    P, W, H = sc["means"].shape[0], cam["W"], cam["H"]
    r = lambda name, dt, shape: np.fromfile(str(d / name), dtype=dt).reshape(shape)
    fw = oracle.forward_full(sc, cam, bg, D=D)
    bw = oracle.backward_full(sc, cam, bg, fw, dpix, D=D)
    assert np.array_equal(r("radii.bin", np.int32, (P,)), fw["geo"]["radii"])
    assert int(r("num_rendered.bin", np.int32, (1,))[0]) > 0
    assert_forward_gate(fw, r("color.bin", np.float32, (3, H, W)), W, H, 1e-4, "C++ caller")
    for name, ref in (("d_means3D.bin", bw["dmean3D"]), ("d_opacity.bin", bw["dopacity"]), ("d_sh.bin", bw["dsh"]), ("d_scale.bin", bw["dscale"]),
                      ("d_rot.bin", bw["drot"])):
        got = np.fromfile(str(d / name), dtype=np.float32).astype(np.float64).reshape(np.shape(ref))
        ref = np.asarray(ref, np.float64)
        assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max(), name
        assert_grads_elementwise(got, ref, name)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc on this box")
def test_cpp_host_program_renders_a_batch_of_deformed_frames(tmp_path, oracle):
    """tests/native/batch_caller.cpp: gm_mesh_rs_packed_batch + gm_forward_deformed_batch_async from C++ - three frames (three deformed
    meshes, three cameras) of a torus-bound cloud in one launch chain.  The program compares every frame, byte for byte, with the
    single-frame calls it stands for (exit code 7 otherwise); here its images go through the oracle: deformation and rotated-direction SH
    colour of every frame from the C restatement (orc_deform, orc_sh_colors_rotated on the per-vertex (R, S) of oracle/mesh_oracle.py's
    float64 definition is NOT used - the tables are the device's own: this test is about the batch boundary), then orc forward."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from gaussianmesh_amd import scenes
    from gaussianmesh_amd.deform import vertex_face_adjacency
    exe = tmp_path / "batch_caller"
    csrc = os.path.join(ROOT, "gaussianmesh_amd", "csrc")
    build = subprocess.run([HIPCC, "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "batch_caller.cpp"),
                            "-L", csrc, "-lgmesh_hip", "-Wl,-rpath," + csrc, "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    P, W, H, F, K = 6000, 240, 136, 6, 3
    host = bench.build_scene(P, W, H, F)
    Vm, NF = host["verts"].shape[0], host["faces"].shape[0]
    off, adj = vertex_face_adjacency(host["faces"], Vm)
    frames, views = [1, 3, 4], [0, 2, 5]
    cams = [scenes.orbit_camera(v, F, W, H) for v in views]
    bg = np.array([0.7, 0.2, 0.4], np.float32)
    d = tmp_path / "scene"
    d.mkdir()
    w = lambda name, a, dt=np.float32: np.ascontiguousarray(a, dtype=dt).tofile(str(d / name))
    w("meta.bin", [P, W, H, K, Vm, NF, 400000], np.int32)
    w("tri.bin", host["tri"], np.int32); w("weights.bin", host["weights"]); w("cov.bin", host["cov"]); w("pos.bin", host["pos"])
    w("shs.bin", host["shs"]); w("opac.bin", host["opac"]); w("verts.bin", host["verts"]); w("faces.bin", host["faces"], np.int32)
    w("adj_offsets.bin", off, np.int32); w("adj_faces.bin", adj, np.int32)
    w("deformed.bin", np.stack([host["mesh"][t][:, 0:3] for t in frames]))
    w("cameras.bin", np.stack([np.concatenate([np.asarray(c["view"], np.float32).ravel(), np.asarray(c["proj"], np.float32).ravel(),
                                               np.asarray(c["campos"], np.float32).ravel(), [c["tanx"], c["tany"]]]) for c in cams]))
    w("background.bin", bg)
    run = subprocess.run([str(exe), str(d)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    colors = np.fromfile(str(d / "colors.bin"), np.float32).reshape(K, 3, H, W)
    radii = np.fromfile(str(d / "radii.bin"), np.int32).reshape(K, P)
    status = np.fromfile(str(d / "status.bin"), np.int32).reshape(K, 4)
    assert (status[:, 0] > 0).all() and (status[:, 3] == 0).all(), status           # every frame rendered, none refused
    # the oracle on each frame: per-vertex (R, S) from the device (gm_mesh_rs, the same kernel the tables come from), then C restatements
    import torch
    from gpu_utils import T
    from gaussianmesh_amd.deform import mesh_rs
    for k in range(K):
        state = mesh_rs(T(host["verts"]), T(np.ascontiguousarray(host["mesh"][frames[k]][:, 0:3])), T(host["faces"], dtype=torch.int32), want_state=True)[2].cpu().numpy()
        dV = state[:, 0:3] - host["verts"]
        p_ref, c_ref, r_ref = oracle.deform(host["tri"], host["weights"], dV, state[:, 3:12].reshape(-1, 3, 3), state[:, 12:21].reshape(-1, 3, 3), host["cov"], host["pos"])
        rgb_ref = oracle.sh_colors_rotated(p_ref, cams[k]["campos"], r_ref, host["shs"], deg=3)
        sc = dict(means=p_ref, opac=host["opac"], colors_precomp=rgb_ref, cov3D_precomp=scenes.strip_symmetric(c_ref))
        fw = oracle.forward_full(sc, cams[k], bg, D=3, use_precomp_cov=True, use_precomp_color=True)
        same = radii[k] == fw["geo"]["radii"]
        err = np.abs(colors[k].astype(np.float64) - fw["color"]).max(axis=0)
        print("C++ batch frame %d: %.4f %% of the radii equal the oracle's; image error vs the oracle on ITS deformed cloud: mean %.2e, 99.9 %% quantile %.2e, max %.2e"
              % (k, 100 * same.mean(), err.mean(), np.quantile(err, 0.999), err.max()))
        # the oracle deforms in float64 and rounds once, the device in float32 (positions agree to 1e-5 relative, tests/test_gpu_fullsize.py): a
        # radius on a rounding edge may differ and pixels move by what a 0.005-pixel shift of a splat moves them - the strict gate belongs to
        # the tests that hand the oracle the device's own deformed cloud; here: the same picture
        assert same.mean() >= 0.999, (k, float(same.mean()))
        if same.all():                                             # (measured: every radius equal, largest pixel error 3.3e-6)
            assert err.max() <= 1e-4, (k, float(err.max()))
        else:
            assert err.mean() <= 2e-4 and np.quantile(err, 0.999) <= 1e-2 and err.max() <= 0.1, k
