"""Independent dense restatement of the rasterizer in torch float64 + autograd.  TEST INFRASTRUCTURE.

Purpose: cross-check oracle/gm_oracle.c (forward AND every backward formula) from a second,
differently-structured derivation: textbook matrix algebra (cov2D = J W S W^T J^T with ordinary
row/column conventions instead of the GLM-transposed forms of the reference), one big
[pixels x gaussians] tensor instead of tile lists, and gradients from autograd instead of the
hand-derived chain rule of cuda_rasterizer/backward.cu.  Discrete decisions (tile rectangle
membership, power>0 / alpha<1/255 skips, the T<1e-4 stop) are reproduced as constant masks.
Only usable for small scenes (memory ~ H*W*P doubles).

Reference semantics that are NOT plain autograd and are reproduced on purpose:
  * frustum clamp of t.xy/t.z: the clamped value is treated as a constant w.r.t. t.z and gets
    zero gradient w.r.t. t.xy  (backward.cu:172-176, 262-264)
  * min(0.99, .) is not masked in the gradient (backward.cu:499, 503-554)
  * dL/dmeans2D is d(loss)/d(ndc-scaled pixel position): pixel-space gradient x (0.5W, 0.5H)
    (backward.cu:460-461, 545-546) and flows on to means3D through the projection
  * dL/dconic -> dL/dcov2D uses 1/(det^2 + 1e-7)  (backward.cu:203) - autograd uses the exact
    inverse; the difference is O(1e-7/det^2) and covered by the tolerance.
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def eval_sh(deg, sh, d):
    """sh [P,M,3], d [P,3] unit."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = SH_C0 * sh[:, 0]
    if deg > 0:
        r = r - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
             + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        r = (r + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
             + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
             + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
             + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return r


def quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)


def render(means, opac, view, proj, campos, W, H, tanx, tany, bg, D=3, shs=None, colors_precomp=None,
           scales=None, rots=None, cov3D_precomp=None, mod=1.0, means2D=None):
    """All tensor args float64 torch tensors (requires_grad as desired).  view/proj are the
    reference's transposed 4x4 (row-vector convention).  Returns (color[3,H,W], aux dict)."""
    P = means.shape[0]
    dt = means.dtype
    fx, fy = W / (2.0 * tanx), H / (2.0 * tany)
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means, ones], 1)
    hom = ph @ proj                                  # row-vector convention
    p_w = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :3] * p_w[:, None]
    t = (ph @ view)[:, :3]
    vis = t[:, 2] > 0.2
    # 3D covariance
    if cov3D_precomp is not None:
        c = cov3D_precomp
        Sig = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(-1, 3, 3)
    else:
        Rm = quat_to_rot(rots)                        # standard rotation matrix (no normalisation)
        L = Rm * (mod * scales)[:, None, :]
        Sig = L @ L.transpose(1, 2)
    # EWA projection with the reference's clamp semantics
    limx, limy = 1.3 * tanx, 1.3 * tany
    tz = t[:, 2]
    txtz, tytz = t[:, 0] / tz, t[:, 1] / tz
    inx = (txtz >= -limx) & (txtz <= limx)
    iny = (tytz >= -limy) & (tytz <= limy)
    tx = torch.where(inx, t[:, 0], (txtz.clamp(-limx, limx) * tz).detach())
    ty = torch.where(iny, t[:, 1], (tytz.clamp(-limy, limy) * tz).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * tx / (tz * tz), zero, fy / tz, -fy * ty / (tz * tz)], 1).reshape(-1, 2, 3)
    Rw2c = view[:3, :3].transpose(0, 1)               # world->view rotation (view is stored transposed)
    JW = J @ Rw2c
    cov2 = JW @ Sig @ JW.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    cc = cov2[:, 1, 1] + 0.3
    det = a * cc - b * b
    conic = torch.stack([cc / det, -b / det, a / det], 1)
    mid = 0.5 * (a + cc)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    if means2D is not None:                           # screen-space probe: d(loss)/d(means2D) = pixel grad * 0.5*(W,H)
        px = px + means2D[:, 0] * (0.5 * W)
        py = py + means2D[:, 1] * (0.5 * H)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ri = radius.to(torch.int64).to(dt)
    pxd, pyd = px.detach().to(torch.float32).to(dt), py.detach().to(torch.float32).to(dt)
    x0 = torch.trunc((pxd - ri) / 16).clamp(0, gx); x1 = torch.trunc((pxd + ri + 15) / 16).clamp(0, gx)
    y0 = torch.trunc((pyd - ri) / 16).clamp(0, gy); y1 = torch.trunc((pyd + ri + 15) / 16).clamp(0, gy)
    vis = vis & (det != 0) & ((x1 - x0) * (y1 - y0) > 0)
    # colours
    if colors_precomp is not None:
        col = colors_precomp
    else:
        d = means - campos[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        col = torch.clamp(eval_sh(D, shs, d) + 0.5, min=0.0)
    # order: (depth as float32 bits, index) ascending
    depth32 = t[:, 2].detach().to(torch.float32)
    order = sorted(range(P), key=lambda i: (float(depth32[i]), i))
    order = torch.tensor([i for i in order if bool(vis[i])], dtype=torch.long)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pixx, pixy = xs.reshape(-1), ys.reshape(-1)
    tX, tY = torch.floor(pixx / 16), torch.floor(pixy / 16)
    T = torch.ones(H * W, dtype=dt)
    C = torch.zeros(H * W, 3, dtype=dt)
    done = torch.zeros(H * W, dtype=torch.bool)
    ncontrib = torch.zeros(H * W, dtype=torch.long)
    count = torch.zeros(H * W, dtype=torch.long)
    op = opac.reshape(-1)
    for g in order.tolist():
        member = (tX >= x0[g]) & (tX < x1[g]) & (tY >= y0[g]) & (tY < y1[g])
        count = count + (member & ~done).long()
        dx = px[g] - pixx
        dy = py[g] - pixy
        power = -0.5 * (conic[g, 0] * dx * dx + conic[g, 2] * dy * dy) - conic[g, 1] * dx * dy
        G = torch.exp(power)
        alpha_raw = op[g] * G
        alpha = torch.where(alpha_raw > 0.99, alpha_raw * 0 + 0.99 + (alpha_raw - alpha_raw.detach()), alpha_raw)
        ok = member & ~done & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
        testT = T * (1 - alpha)
        stop = ok & (testT.detach() < 1e-4)
        done = done | stop
        ok = ok & ~stop
        okf = ok.to(dt)
        C = C + (col[g][None, :] * (alpha * T * okf)[:, None])
        T = torch.where(ok, testT, T)
        ncontrib = torch.where(ok, count, ncontrib)
    out = (C + T[:, None] * bg[None, :]).transpose(0, 1).reshape(3, H, W)
    aux = dict(radii=torch.where(vis, radius, torch.zeros_like(radius)).to(torch.int32), final_T=T.detach(),
               n_contrib=ncontrib, conic=conic.detach(), xy=torch.stack([px, py], 1).detach(), rgb=col.detach(),
               depth=t[:, 2].detach(), cov2=torch.stack([a, b, cc], 1).detach(),
               live=dict(conic=conic, px=px, py=py))    # non-detached intermediates: retain_grad() them before backward to get the
    return out, aux                                     # float64 truth of dL/dconic and dL/d(pixel position) (tools/needle_stages.py)
