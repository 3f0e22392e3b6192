"""The C-ABI library loads without a GPU and exports every symbol include/gmesh_hip.h declares; size queries and
argument validation (which run before any HIP call) behave; the product path has no CPU fallback."""
import ctypes as C

import numpy as np
import pytest
import torch


def test_library_exports_every_header_symbol():
    from gaussianmesh_amd import _lib
    l = _lib.lib()
    names = _lib.header_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(l, n), n
    assert set(names) == set(_lib.SIGNATURES), "python signatures out of sync with include/gmesh_hip.h"
    assert l.gm_abi_version() == 3


def test_scratch_sizes_scale_linearly():
    from gaussianmesh_amd import _lib
    l = _lib.lib()
    g1, g2 = l.gm_geom_bytes(1_000_000), l.gm_geom_bytes(2_000_000)
    assert 130e6 < g1 < 195e6 and abs(g2 - 2 * g1) < 2e6       # ~187 B per Gaussian (incl. 48 B backward accumulators, ordering scratch and histograms)
    b1 = l.gm_binning_bytes(8_000_000)
    assert 128e6 <= b1 < 150e6                                   # 16 B per instance + 2 B of histogram rows
    i1 = l.gm_image_bytes(1920, 1080)
    assert 16.5e6 < i1 < 17e6                                    # 8 B per pixel + tiles
    assert l.gm_geom_bytes(0) > 0 and l.gm_binning_bytes(0) > 0
    assert l.gm_knn_workspace_bytes(1000) > 16 * 1000


def test_argument_validation_happens_before_any_gpu_work():
    from gaussianmesh_amd import _lib
    l = _lib.lib()
    R = C.c_int(-1)
    one = 1   # bogus non-null "pointers": validation must reject before dereferencing anything
    rc = l.gm_forward_0(one, 10, 3, 16, one, 64, 64, one, one, one, one, one, 1.0, one, None, one, one, one, 0.5, 0.5, 0, None, 0, None, C.byref(R))
    assert rc == 1 and b"exactly one of shs / colors_precomp" in l.gm_last_error()
    rc = l.gm_forward_0(one, 10, 3, 16, one, 64, 64, one, one, None, one, one, 1.0, one, one, one, one, one, 0.5, 0.5, 0, None, 0, None, C.byref(R))
    assert rc == 1 and b"scales, rotations" in l.gm_last_error()
    rc = l.gm_forward_0(one, 10, 3, 9, one, 64, 64, one, one, None, one, one, 1.0, one, None, one, one, one, 0.5, 0.5, 0, None, 0, None, C.byref(R))
    assert rc == 1 and b"SH degree" in l.gm_last_error()
    rc = l.gm_forward_0(one, -1, 0, 0, one, 64, 64, one, None, one, one, None, 1.0, None, one, one, one, one, 0.5, 0.5, 0, None, 0, None, C.byref(R))
    assert rc == 1
    rc = l.gm_forward_0_async(7, one, 10, 3, 16, one, 64, 64, one, one, None, one, one, 1.0, one, None, one, one, one, 0.5, 0.5, 0, None, 0, None, None, None)
    assert rc == 1 and b"emission policy" in l.gm_last_error()
    assert l.gm_forward_1_geom(2, one, one, one, 10, -1, -5, one, 64, 64, one, 0, None, None, 0, None) == 1          # negative capacity
    assert l.gm_knn(5, None, None, None, 0, None) == 1
    assert l.gm_sh_colors(5, 4, 16, one, one, None, one, one, None) == 1


def test_round6_entry_points_validate_before_any_gpu_work():
    """gm_forward_deformed_batch_async / gm_mesh_rs_packed_batch / gm_backward_sh_step (ABI 3): every refusal below is decided on the
    arguments alone - no device is touched, so it holds on a box without one."""
    from gaussianmesh_amd import _lib
    l = _lib.lib()
    one = 1
    K = _lib.GM_BATCH_MAX
    frames = (_lib.BatchFrame * (K + 1))()
    call = lambda k, fr, P=10, deg=3, M=16, W=64, H=64, cap=1000, flags=1, pol=2: l.gm_forward_deformed_batch_async(
        pol, k, fr, P, deg, M, W, H, one, one, one, one, one, one, one, cap, flags, None, 0, None)
    assert call(K + 1, frames) == 1 and b"frames" in l.gm_last_error()
    assert call(0, frames) == 1
    assert call(2, None) == 1
    assert call(2, frames, flags=8) == 1 and b"unknown flags" in l.gm_last_error()
    assert call(2, frames, P=0) == 1 and b"single-frame calls" in l.gm_last_error()
    assert call(2, frames, M=9) == 1
    assert call(2, frames, cap=0) == 1 and b"binning_capacity" in l.gm_last_error()
    assert call(2, frames, W=3840, H=2160, pol=2) == 1 and b"list tiles" in l.gm_last_error()      # 4K under 32-px parents: two tile passes
    assert call(2, frames, pol=9) == 1 and b"emission policy" in l.gm_last_error()
    assert call(2, frames) == 1 and b"null pointer" in l.gm_last_error()                            # frames of nulls
    for k in range(2):                                                                              # unaligned scratch, then a shared buffer
        for name, _ in _lib.BatchFrame._fields_:
            if name not in ("tan_fovx", "tan_fovy"):
                setattr(frames[k], name, 4096 * (k + 1) + 8)
    assert call(2, frames) == 1 and b"256-byte aligned" in l.gm_last_error()
    for k in range(2):
        frames[k].geom_buffer, frames[k].binning_buffer, frames[k].image_buffer = 1 << 20, 2 << 20, 3 << 20
    assert call(2, frames) == 1 and b"share a buffer" in l.gm_last_error()
    ptrs = (C.c_void_p * (K + 1))(*([4096] * (K + 1)))
    assert l.gm_mesh_rs_packed_batch(K + 1, 10, 10, one, ptrs, one, one, one, ptrs, None) == 1
    assert l.gm_mesh_rs_packed_batch(0, 10, 10, one, ptrs, one, one, one, ptrs, None) == 1
    assert l.gm_mesh_rs_packed_batch(2, 10, 10, one, None, one, one, one, ptrs, None) == 1
    odd = (C.c_void_p * 2)(4096, 4100)
    assert l.gm_mesh_rs_packed_batch(2, 10, 10, one, ptrs, one, one, one, odd, None) == 1 and b"unaligned" in l.gm_last_error()
    step = lambda M=16, D=3, rows=10, P=10, st=1, m=one, shs=one: l.gm_backward_sh_step(
        2, P, D, M, 5, one, 64, 64, one, shs, one, 1.0, one, None, one, one, one, 0.5, 0.5, one, one, one, one, one, one, one, one, None, one, one, rows, m, one,
        1e-3, 1e-4, 0.9, 0.999, 1e-15, st, 0, None)
    assert step(M=9) == 1 and b"SH operand" in l.gm_last_error()
    assert step(shs=None) == 1
    assert step(D=4) == 1
    assert step(rows=11) == 1 and b"optimizer state" in l.gm_last_error()
    assert step(st=0) == 1
    assert step(m=None) == 1


def test_no_cpu_fallback():
    from gaussianmesh_amd import GaussianRasterizationSettings, GaussianRasterizer, distCUDA2
    from gaussianmesh_amd._lib import GmeshError
    from gaussianmesh_amd.deform import deform_tensors
    P = 8
    rs = GaussianRasterizationSettings(32, 32, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3), False, False)
    m = torch.zeros(P, 3)
    with pytest.raises(GmeshError):
        GaussianRasterizer(rs)(m, m, torch.ones(P, 1), colors_precomp=torch.ones(P, 3), cov3D_precomp=torch.ones(P, 6))
    with pytest.raises(GmeshError):
        distCUDA2(m)
    with pytest.raises(GmeshError):
        deform_tensors(torch.zeros(P, 3, dtype=torch.int32), m, m, torch.zeros(P, 3, 3), torch.zeros(P, 3, 3), torch.zeros(P, 3, 3), m)


def test_operator_argument_rules_match_reference():
    """diff_gaussian_rasterizater/__init__.py:146-150: exactly one of shs/colors and of (scales,rotations)/cov3D."""
    from gaussianmesh_amd import GaussianRasterizationSettings, GaussianRasterizer, NewGaussianRasterizer
    rs = GaussianRasterizationSettings(32, 32, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0, torch.zeros(3), False, False)
    m = torch.zeros(4, 3); o = torch.ones(4, 1); sh = torch.zeros(4, 16, 3); s = torch.ones(4, 3); q = torch.ones(4, 4)
    for cls in (GaussianRasterizer, NewGaussianRasterizer):
        r = cls(rs)
        assert r.execute.__func__ is r.forward.__func__            # Jittor spelling kept
        with pytest.raises(Exception, match="SHs or precomputed colors"):
            r(m, m, o, scales=s, rotations=q)
        with pytest.raises(Exception, match="SHs or precomputed colors"):
            r(m, m, o, shs=sh, colors_precomp=m, scales=s, rotations=q)
        with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
            r(m, m, o, shs=sh, scales=s)
        with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
            r(m, m, o, shs=sh, scales=s, rotations=q, cov3D_precomp=torch.ones(4, 6))
    # the reference's twelve fields in its order; one optional extension behind them (default None: reference call sites unchanged)
    assert GaussianRasterizationSettings._fields[:12] == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier",
                                                          "viewmatrix", "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
    assert GaussianRasterizationSettings._fields[12:] == ("work_hint",) and rs.work_hint is None


def test_import_has_no_side_effects_and_configure_runtime_is_opt_in():
    """HIP multiplexes streams onto four hardware queues unless GPU_MAX_HW_QUEUES says otherwise, and a four-stream frame loop loses
    17 % to that.  Importing the package leaves the environment alone (round 5; it used to export the variable);
    configure_runtime() exports 8 - read by the runtime at its first call, which comes later - leaves a caller's own setting alone and
    is switched off by GM_NO_RUNTIME_CONFIG=1."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, gaussianmesh_amd as g; a = os.environ.get('GPU_MAX_HW_QUEUES'); r = g.configure_runtime(ipc_dmabuf=True); "
            "print(a, os.environ.get('GPU_MAX_HW_QUEUES'), r['applied'], os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))")
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY", "GM_NO_RUNTIME_CONFIG")}
    run = lambda e: subprocess.run([sys.executable, "-c", code], cwd=root, env=e, capture_output=True, text=True, timeout=300)
    out = run(env)
    assert out.stdout.split() == ["None", "8", "True", "0"], out.stderr[-500:]
    out = run(dict(env, GPU_MAX_HW_QUEUES="4"))
    assert out.stdout.split() == ["4", "4", "False", "0"], out.stderr[-500:]
    out = run(dict(env, GM_NO_RUNTIME_CONFIG="1"))
    assert out.stdout.split() == ["None", "None", "False", "None"], out.stderr[-500:]
