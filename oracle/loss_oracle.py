"""CPU restatement of the reference's photometric loss -- TEST INFRASTRUCTURE ONLY (see oracle/gm_oracle.c header):
nothing under gaussianmesh_amd/ imports this file.

Follows utils/loss_utils.py of the reference:
  gaussian()/create_window()  :23-32   1-D Gaussian (sigma 1.5, 11 taps) in Python doubles -> float32 array, divided by its
                                        float32 sum; 2-D window = outer product (float32), one copy per channel
  _ssim()                     :44-81   five depthwise conv2d (zero padding 5) -> mu, sigma, ssim map -> mean
  l1_loss()                   :17-18
Parity status: the reference implementation needs Jittor (absent) to run, so no fixture of its outputs exists:
"parity unpinned" for this row as well.  What pins this file: two independent evaluations of the same definition
(scipy.signal.correlate2d with the 2-D window in float64 below, torch.nn.functional.conv2d in float64 with autograd)
agree to 1e-12 (tests/test_loss_oracle.py), plus known answers (ssim(x, x) = 1, symmetry, constant images).
"""
import numpy as np


def window_1d(window_size=11, sigma=1.5):
    g = np.array([np.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)], dtype=np.float32)
    return (g / g.sum(dtype=np.float32)).astype(np.float32)


def window_2d(window_size=11):
    g = window_1d(window_size)
    return (g[:, None] * g[None, :]).astype(np.float32)          # float32 outer product, as _1D.matmul(_1D.t()).float()


def ssim_map(img1, img2):
    """[C,H,W] float arrays -> ssim map [C,H,W] (float64 arithmetic, float32 window values)."""
    from scipy.signal import correlate2d
    w = window_2d().astype(np.float64)
    a = np.asarray(img1, np.float64); b = np.asarray(img2, np.float64)
    conv = lambda x: np.stack([correlate2d(x[c], w, mode="same", boundary="fill", fillvalue=0.0) for c in range(x.shape[0])])
    mu1, mu2 = conv(a), conv(b)
    s1 = conv(a * a) - mu1 * mu1
    s2 = conv(b * b) - mu2 * mu2
    s12 = conv(a * b) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))


def ssim(img1, img2):
    return float(ssim_map(img1, img2).mean())


def l1(img1, img2):
    return float(np.abs(np.asarray(img1, np.float64) - np.asarray(img2, np.float64)).mean())


def ssim_torch(img1, img2, size_average=True):
    """Differentiable float64 evaluation with torch conv2d (independent of ssim_map); img [C,H,W] or [B,C,H,W] tensors."""
    import torch
    import torch.nn.functional as F
    a = img1 if img1.dim() == 4 else img1[None]
    b = img2 if img2.dim() == 4 else img2[None]
    a = a.double(); b = b.double()
    C = a.shape[1]
    w = torch.tensor(window_2d(), dtype=torch.float64)[None, None].expand(C, 1, 11, 11).contiguous()
    conv = lambda x: F.conv2d(x, w, padding=5, groups=C)
    mu1, mu2 = conv(a), conv(b)
    s1 = conv(a * a) - mu1 * mu1
    s2 = conv(b * b) - mu2 * mu2
    s12 = conv(a * b) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


def photometric_torch(image, gt, lam):
    return (1.0 - lam) * (image.double() - gt.double()).abs().mean() + lam * (1.0 - ssim_torch(image, gt))
