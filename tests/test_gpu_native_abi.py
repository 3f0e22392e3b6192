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
