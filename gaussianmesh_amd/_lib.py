"""ctypes binding of libgmesh_hip.so (C ABI: include/gmesh_hip.h).

The HIP library is the product path: there is NO CPU fallback.  If the shared object is missing or a
symbol declared in the header cannot be resolved, importing/using this module fails loudly.
"""
import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(CSRC, "libgmesh_hip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "gmesh_hip.h")
_lib = None

vp, i32, i64, f32, f64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t

GM_BATCH_MAX = 8


class BatchFrame(C.Structure):
    """gm_batch_frame (include/gmesh_hip.h): one frame of gm_forward_deformed_batch_async."""
    _fields_ = [("packed", vp), ("viewmatrix", vp), ("projmatrix", vp), ("cam_pos", vp), ("tan_fovx", f32), ("tan_fovy", f32),
                ("geom_buffer", vp), ("binning_buffer", vp), ("image_buffer", vp), ("out_color", vp), ("radii", vp), ("status_host", vp)]


# name -> (restype, argtypes); mirrors include/gmesh_hip.h one to one
SIGNATURES = {
    "gm_abi_version": (i32, []),
    "gm_last_error": (C.c_char_p, []),
    "gm_geom_bytes": (sz, [i32]),
    "gm_image_bytes": (sz, [i32, i32]),
    "gm_work_hint_bytes": (sz, [i32, i32]),
    "gm_binning_bytes": (sz, [i64]),
    "gm_forward_0": (i32, [vp, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, f32, f32, i32,
                           vp, i32, vp, C.POINTER(i32)]),
    "gm_forward_0_async": (i32, [i32, vp, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, f32, f32, i32,
                                 vp, i32, vp, vp, vp]),
    "gm_forward_1": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp,
                           f32, f32, i32, vp, vp, i32, vp]),
    "gm_backward": (i32, [i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp,
                          vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "gm_backward_p": (i32, [i32, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp,
                            vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "gm_backward_sh_step": (i32, [i32, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, f32, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                  vp, i32, vp, vp, f32, f32, f64, f64, f64, i32, i32, vp]),
    "gm_mark_visible": (i32, [i32, vp, vp, vp, vp, vp]),
    "gm_geom_field": (vp, [vp, i32, C.c_char_p]),
    "gm_splat_floats": (i32, []),
    "gm_image_field": (vp, [vp, i32, i32, C.c_char_p]),
    "gm_binning_field": (vp, [vp, i64, i32, i32, i32, C.c_char_p]),
    "gm_knn_workspace_bytes": (sz, [i32]),
    "gm_knn": (i32, [i32, vp, vp, vp, sz, vp]),
    "gm_deform": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gm_sh_colors": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "gm_deform_shade": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gm_pack_mesh_state": (i32, [i32, vp, vp, vp, vp]),
    "gm_forward_0_deformed_async": (i32, [i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp,
                                          i32, vp, vp, vp]),
    "gm_forward_0_deformed_stream_async": (i32, [i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp,
                                                 i32, vp, vp, vp, vp, vp, i32]),
    "gm_depth_plan_bytes": (sz, []),
    "gm_depth_slab_bytes": (sz, [i32]),
    "gm_forward_1_geom": (i32, [i32, vp, vp, vp, i32, i32, i64, vp, i32, i32, vp, i32, vp, vp, i32, vp]),
    "gm_forward_status_async": (i32, [vp, i32, vp, vp]),
    "gm_forward_deformed_batch_async": (i32, [i32, i32, C.POINTER(BatchFrame), i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, i32, vp]),
    "gm_mesh_rs_packed_batch": (i32, [i32, i32, i32, vp, C.POINTER(vp), vp, vp, vp, C.POINTER(vp), vp]),
    "gm_deform_shade_packed": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gm_cov_to_scale_rot": (i32, [i32, vp, vp, vp, vp]),
    "gm_mesh_rs": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "gm_mesh_rs_packed": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "gm_mesh_activate_fwd": (i32, [i32, f32] + [vp] * 10 + [vp] * 4 + [f32, vp] + [vp]),
    "gm_mesh_activate_bwd": (i32, [i32, f32] + [vp] * 10 + [vp] * 4 + [vp] * 5 + [f32, vp] + [vp]),
    "gm_adam_step": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, f64, f64, f64, i32, vp]),
    "gm_adam_step_active": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f64, f64, f64, i32, vp]),
    "gm_densify_stats": (i32, [i32, vp, vp, vp, vp, vp, vp]),
    "gm_ssim_partials": (i64, [i32, i32, i32]),
    "gm_ssim_fwd": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "gm_ssim_bwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "gm_loss_combine": (i32, [vp, i64, f64, f64, f64, vp, vp]),
    "gm_profile_enable": (None, [i32]),
    "gm_profile_reset": (None, []),
    "gm_profile_read": (i32, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(i64)]),
    "gm_debug_render_trace": (None, [vp]),
    "gm_debug_bucket_trace": (None, [vp]),
}


def header_symbols():
    """Every function name declared in include/gmesh_hip.h."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gm_[a-z0-9_]+)\s*\(", txt)))


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into csrc/libgmesh_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libgmesh_hip.so failed")
    return SO_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                "gaussianmesh_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C gaussianmesh_amd/csrc`). There is no CPU fallback." % SO_PATH)
        l = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the symbol is absent -> loud failure
            fn.restype = res
            fn.argtypes = args
        if l.gm_abi_version() != 3:
            raise ImportError("libgmesh_hip.so ABI version mismatch")
        _lib = l
    return _lib


class GmeshError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise GmeshError("libgmesh_hip: error %d: %s" % (rc, lib().gm_last_error().decode()))
