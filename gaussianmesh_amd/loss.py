"""Photometric losses of the training loop, mirroring utils/loss_utils.py of the reference.

  l1_loss(network_output, gt)                       loss_utils.py:17-18
  mesh_restrict_loss(scale, v1, v2, v3, weight)     loss_utils.py:83-108 (torch elementwise; once per iteration on [N,3])
  l2_loss(network_output, gt)                       loss_utils.py:20-21
  ssim(img1, img2, window_size=11, size_average=True)   loss_utils.py:35-81
  photometric_loss(image, gt, lambda_dssim)         train_mesh_gaussian.py:92-94:
                                                    (1 - l) * l1_loss + l * (1 - ssim), one fused pass

ssim and photometric_loss run csrc/gm_loss.hip (gm_ssim_fwd / gm_ssim_bwd) and are differentiable with respect to
the first image (the rendered one); the ground truth gets no gradient, as in the training loop.  There is no CPU path.
"""
import torch

from . import _lib


def l1_loss(network_output, gt):
    return (network_output - gt).abs().mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


def _planes(img):
    if img.dim() == 3:
        return img.shape[0], 1
    if img.dim() == 4:
        return img.shape[0] * img.shape[1], img.shape[0]
    raise ValueError("ssim expects [C,H,W] or [B,C,H,W] images")


def _fwd(img1, img2, want_grad):
    lib = _lib.lib()
    if img1.device.type != "cuda":
        raise _lib.GmeshError("ssim needs tensors on a HIP (cuda) device; there is no CPU path")
    if img1.shape != img2.shape:
        raise ValueError("ssim: image shapes differ: %s vs %s" % (tuple(img1.shape), tuple(img2.shape)))
    a = img1.detach().contiguous().float()
    b = img2.detach().contiguous().float()
    planes, _ = _planes(a)
    H, W = a.shape[-2], a.shape[-1]
    dev = a.device
    n = int(lib.gm_ssim_partials(planes, H, W))
    partial = torch.empty((n, 2), dtype=torch.float32, device=dev)
    maps = torch.empty((3,) + tuple(a.shape), dtype=torch.float32, device=dev) if want_grad else None
    mp = [maps[i].data_ptr() for i in range(3)] if want_grad else [None, None, None]
    with torch.cuda.device(dev):
        _lib.check(lib.gm_ssim_fwd(a.data_ptr(), b.data_ptr(), planes, H, W, mp[0], mp[1], mp[2], partial.data_ptr(),
                                   torch.cuda.current_stream(dev).cuda_stream))
    return a, b, maps, partial.view(planes, -1, 2)


def _bwd(a, b, maps, g_ssim_planes, g_l1):
    lib = _lib.lib()
    planes, _ = _planes(a)
    H, W = a.shape[-2], a.shape[-1]
    grad = torch.empty_like(a)
    g_ssim_planes = g_ssim_planes.contiguous().float()          # views of one small tensor are contiguous already
    g_l1 = None if g_l1 is None else g_l1.reshape(1).contiguous().float()
    with torch.cuda.device(a.device):
        _lib.check(lib.gm_ssim_bwd(a.data_ptr(), b.data_ptr(), maps[0].data_ptr(), maps[1].data_ptr(), maps[2].data_ptr(), planes, H, W,
                                   g_ssim_planes.data_ptr(), None if g_l1 is None else g_l1.data_ptr(), grad.data_ptr(),
                                   torch.cuda.current_stream(a.device).cuda_stream))
    return grad


class _Ssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, size_average):
        a, b, maps, partial = _fwd(img1, img2, ctx.needs_input_grad[0])
        planes, batch = _planes(a)
        per_plane = partial[:, :, 0].double().sum(dim=1)                     # [planes]
        count = a.shape[-2] * a.shape[-1]
        ctx.size_average, ctx.shape, ctx.batch = size_average, img1.shape, batch
        if maps is not None:
            ctx.save_for_backward(a, b, maps)
        if size_average or a.dim() == 3:
            out = (per_plane.sum() / (planes * count)).float()
            return out if size_average else out.reshape(1)
        return (per_plane.view(batch, -1).sum(dim=1) / (planes // batch * count)).float()

    @staticmethod
    def backward(ctx, g):
        a, b, maps = ctx.saved_tensors
        planes, batch = _planes(a)
        count = a.shape[-2] * a.shape[-1]
        if ctx.size_average or a.dim() == 3:
            gp = (g.reshape(1) / (planes * count)).expand(planes)
        else:
            gp = (g.reshape(batch, 1) / (planes // batch * count)).expand(batch, planes // batch).reshape(planes)
        return _bwd(a, b, maps, gp, None).view(ctx.shape), None, None


def ssim(img1, img2, window_size=11, size_average=True):
    """Mean structural similarity; size_average=False returns one mean per image of a [B,C,H,W] batch
    (loss_utils.py:78-81).  Only the reference's window (11, sigma 1.5) is built."""
    if window_size != 11:
        raise NotImplementedError("ssim: only window_size=11 (the reference's only use) is implemented")
    return _Ssim.apply(img1, img2, bool(size_average))


_COEF_CACHE = {}


def _coefs(lam, n, planes, device):
    """device constants of the fused loss: value = lam + <coef, (sum ssim, sum |a-b|)>; per-plane / L1 gradient scales"""
    key = (float(lam), int(n), int(planes), device)
    c = _COEF_CACHE.get(key)
    if c is None:
        val = torch.tensor([-lam / n, (1.0 - lam) / n], dtype=torch.float64, device=device)
        grad = torch.tensor([-lam / n] * planes + [(1.0 - lam) / n], dtype=torch.float32, device=device)
        c = _COEF_CACHE[key] = (val, grad)
    return c


class _Photometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim):
        a, b, maps, partial = _fwd(image, gt, ctx.needs_input_grad[0])
        lam, n = float(lambda_dssim), a.numel()
        planes, _ = _planes(a)
        _, grad = _coefs(lam, n, planes, a.device)
        ctx.shape, ctx.planes = image.shape, planes
        if maps is not None:
            ctx.save_for_backward(a, b, maps, grad)
        out = torch.empty((), dtype=torch.float32, device=a.device)
        flat = partial.reshape(-1, 2)
        with torch.cuda.device(a.device):                                      # lam + <(-lam/n, (1-lam)/n), (sum ssim, sum |a-b|)>, one launch
            _lib.check(_lib.lib().gm_loss_combine(flat.data_ptr(), flat.shape[0], -lam / n, (1.0 - lam) / n, lam, out.data_ptr(),
                                                  torch.cuda.current_stream(a.device).cuda_stream))
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, maps, grad = ctx.saved_tensors
        gv = grad * g.reshape(1)                                               # [planes] ssim scales | [1] L1 scale
        return _bwd(a, b, maps, gv[:ctx.planes], gv[ctx.planes:]).view(ctx.shape), None, None


def photometric_loss(image, gt, lambda_dssim=0.2):
    """(1 - lambda) * l1_loss(image, gt) + lambda * (1 - ssim(image, gt)) in one forward and one backward kernel."""
    return _Photometric.apply(image, gt, float(lambda_dssim))


def distance(point1, point2):
    """loss_utils.py:83-84"""
    return torch.linalg.vector_norm(point1 - point2, dim=1)


def circumradius(point1, point2, point3):
    """loss_utils.py:86-101: despite the name, sqrt(|AB x AC|) (the square root of twice the triangle's area) - kept as
    the reference computes it."""
    areas = torch.linalg.vector_norm(torch.cross(point2 - point1, point3 - point1, dim=1), dim=1)
    return torch.sqrt(areas)


def mesh_restrict_loss(scale, point1, point2, point3, weight=10):
    """loss_utils.py:103-108: sum over Gaussians of max(0, max_axis(scale) - weight * circumradius(face)); keeps a bound
    Gaussian from growing past its triangle (train_mesh_gaussian.py:93)."""
    max_s = scale.max(dim=1).values
    return torch.clamp(max_s - weight * circumradius(point1, point2, point3), min=0).sum()
