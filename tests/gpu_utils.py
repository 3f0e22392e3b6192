"""Helpers for the -m gpu tests: run the HIP path through the python operator API / C ABI and expose the
opaque scratch state as numpy arrays (via gm_*_field)."""
import numpy as np
import torch

from gaussianmesh_amd import _lib
from gaussianmesh_amd import rasterizer as R


def dev():
    return torch.device("cuda:0")


def T(a, rg=False, dtype=torch.float32):
    t = torch.tensor(np.asarray(a), dtype=dtype, device=dev())
    if rg:
        t.requires_grad_(True)
    return t


def settings(cam, bg, D, mod=1.0, debug=False):
    return R.GaussianRasterizationSettings(
        image_height=cam["H"], image_width=cam["W"], tanfovx=cam["tanx"], tanfovy=cam["tany"], bg=T(bg),
        scale_modifier=mod, viewmatrix=T(cam["view"]), projmatrix=T(cam["proj"]), sh_degree=D, campos=T(cam["campos"]),
        prefiltered=False, debug=debug)


DEFAULT_MODE = 2


def set_policy(mode):
    """emission policy of the following forwards (0 = the reference's lists ... 3), see gm_common.h"""
    R.set_default_emission_policy(int(mode))


def _view(buf, ptr, count, dtype):
    off = ptr - buf.data_ptr()
    nbytes = count * torch.tensor([], dtype=dtype).element_size()
    return buf[off:off + nbytes].view(dtype).cpu().numpy()


def forward_state(scene, cam, bg, D=3, use_precomp_cov=False, use_precomp_color=False, mod=1.0, debug=False, tile_cull=False):
    """Low-level forward returning colour + every intermediate the oracle also exposes.
    tile_cull=False: emit every tile of the rectangle like the reference (lists comparable 1:1 with the oracle);
    tile_cull=True: the product default (instances the Gaussian cannot reach are not emitted)."""
    lib = _lib.lib()
    mode = int(tile_cull)
    P = scene["means"].shape[0]
    W, H = cam["W"], cam["H"]
    sh = None if use_precomp_color else T(scene["shs"])
    col = T(scene["colors_precomp"]) if use_precomp_color else None
    sc = None if use_precomp_cov else T(scene["scales"])
    rot = None if use_precomp_cov else T(scene["rots"])
    cov = T(scene["cov3D_precomp"]) if use_precomp_cov else None
    nr, color, radii, geom, binning, img = R.rasterize_forward(
        T(bg), T(scene["means"]), col, T(scene["opac"]), sc, rot, mod, cov, T(cam["view"]), T(cam["proj"]), cam["tanx"],
        cam["tany"], H, W, sh, D, T(cam["campos"]), False, debug, emission_policy=mode)
    torch.cuda.synchronize()
    sh_ = max(mode - 1, 0)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    tiles = ((gx + (1 << sh_) - 1) >> sh_) * ((gy + (1 << sh_) - 1) >> sh_)       # lists are per parent tile
    out = dict(R=nr, color=color.cpu().numpy(), radii=radii.cpu().numpy(), geom=geom, binning=binning, img=img)
    if P > 0:
        gp = lambda n: lib.gm_geom_field(geom.data_ptr(), P, n.encode())
        SF = lib.gm_splat_floats()                       # 9: x, y, conic xyz.., opacity, rgb (helpers below index it by name)
        out["splat"] = _view(geom, gp("splat"), P * SF, torch.float32).reshape(P, SF)
        out["depth_key"] = _view(geom, gp("depth_key"), P, torch.int32).astype(np.uint32)
        out["tiles"] = _view(geom, gp("tiles_touched"), P, torch.int32).astype(np.uint32)
        out["cov3D"] = _view(geom, gp("cov3D"), P * 6, torch.float32).reshape(P, 6)
        out["clamped"] = _view(geom, gp("clamped"), P, torch.uint8)
        V = int(_view(geom, gp("bucket_start"), 2049, torch.int32)[2048])
        out["order"] = _view(geom, gp("order"), P, torch.int32).astype(np.uint32)[:V]      # visible Gaussians in (depth, id) order
    ip = lambda n: lib.gm_image_field(img.data_ptr(), W, H, n.encode())
    out["final_T"] = _view(img, ip("final_T"), W * H, torch.float32)
    out["n_contrib"] = _view(img, ip("n_contrib"), W * H, torch.int32).astype(np.uint32)
    out["ranges"] = _view(img, ip("ranges"), tiles * 2, torch.int32).astype(np.uint32).reshape(tiles, 2)
    if nr > 0:
        pp = lib.gm_binning_field(binning.data_ptr(), nr, W, H, mode, b"pairs")
        pairs = _view(binning, pp, 2 * nr, torch.int32).astype(np.uint32).reshape(nr, 2)       # (key, Gaussian id) per instance, tile-sorted
        out["point_list"] = np.ascontiguousarray(pairs[:, 1])
        out["tile_keys"] = np.ascontiguousarray(pairs[:, 0])
    else:
        out["point_list"] = np.zeros(0, np.uint32)
        out["tile_keys"] = np.zeros(0, np.uint32)
    out["child_mask"] = out["tile_keys"] >> 16
    out["tile_keys"] = out["tile_keys"] & 0xFFFF
    return out
