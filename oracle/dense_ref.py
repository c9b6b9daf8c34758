"""Dense, differentiable float64 PyTorch restatement of the rasterizer (no tiling in the math: every
pixel looks at every Gaussian whose tile rectangle covers the pixel's tile).

TEST INFRASTRUCTURE ONLY.  Purpose: an independent derivation of every analytic backward formula
in oracle/gs_oracle.c through autograd, at tiny sizes (<= a few hundred Gaussians, <= 64x64).
Follows SURVEY.md 8a rows A4/A6 and the same reference files as gs_oracle.c.  Discrete decisions
(cull, tile rectangle, alpha / transmittance thresholds, the 1.3*tanfov clamp mask, SH clamp)
are taken on the forward values and are not differentiated through, as in the reference kernels.
"""
import math

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def _sh_rgb(deg, sh, dirs):
    """sh (N,M,3), dirs (N,3) unit -> (N,3); utils/sh_utils.py:58-101."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    r = C0 * sh[:, 0]
    if deg > 0:
        r = r - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            r = (r + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                 + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                r = (r + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
                     + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                     + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
                     + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return r


def render(W, H, tanfovx, tanfovy, bg, viewmatrix, projmatrix, campos, means3D, means2D, opacities,
           shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, sh_degree=0,
           scale_modifier=1.0):
    """All tensor arguments float64.  `means2D` (N,3) is the zero 'screenspace_points' tensor of
    gaussian_renderer/__init__.py:76: its gradient is defined as dL/d(NDC position of the centre),
    which is what the reference wrapper returns for it.  Returns (color[3,H,W], radii[N], aux)."""
    dt = torch.float64
    N = means3D.shape[0]
    V = viewmatrix.to(dt).reshape(4, 4)
    PV = projmatrix.to(dt).reshape(4, 4)
    ones = torch.ones(N, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    p_view = ph @ V  # row-vector convention
    p_hom = ph @ PV
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    ndc = p_hom[:, :2] * p_w[:, None]
    depth = p_view[:, 2]
    if cov3D_precomp is not None:
        c6 = cov3D_precomp
        Sig = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4], c6[:, 2], c6[:, 4], c6[:, 5]],
                          1).view(N, 3, 3)
    else:
        r, x, y, z = rotations.unbind(1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(N, 3, 3)
        # The reference kernel's scale gradient omits the scale_modifier factor:
        # differentiate w.r.t. s = mod*scale and hand that back for `scales`.
        s = scales * scale_modifier if scale_modifier == 1.0 else (scales.detach() * (scale_modifier - 1.0) + scales)
        L = R * s.unsqueeze(1)
        Sig = L @ L.transpose(1, 2)
    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz = p_view[:, 0] / tz
    tytz = p_view[:, 1] / tz
    inx = (txtz >= -limx) & (txtz <= limx)
    iny = (tytz >= -limy) & (tytz <= limy)
    # clamped axis: the value lim*tz is used as a constant (its gradient is dropped in the kernel)
    tx = torch.where(inx, p_view[:, 0], (txtz.clamp(-limx, limx) * tz).detach())
    ty = torch.where(iny, p_view[:, 1], (tytz.clamp(-limy, limy) * tz).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], 1).view(N, 2, 3)
    Rv = V[:3, :3].t()  # W2C rotation (viewmatrix holds its transpose)
    Mx = J @ Rv
    cov = Mx @ Sig @ Mx.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + 0.3
    det = a * c - b * b
    conA, conB, conC = c / det, -b / det, a / det
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16
    pxd, pyd = px.detach(), py.detach()
    minx = torch.trunc((pxd - radius) / 16).clamp(0, gx)
    miny = torch.trunc((pyd - radius) / 16).clamp(0, gy)
    maxx = torch.trunc((pxd + radius + 15) / 16).clamp(0, gx)
    maxy = torch.trunc((pyd + radius + 15) / 16).clamp(0, gy)
    visible = (depth.detach() > 0.2) & (det.detach() != 0) & (((maxx - minx) * (maxy - miny)) > 0)
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)
    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - campos.to(dt)[None]
        dirs = d / d.norm(dim=1, keepdim=True)
        raw = _sh_rgb(sh_degree, shs, dirs) + 0.5
        rgb = torch.clamp_min(raw, 0.0)
    # screen-space position seen by the blend; means2D enters in NDC units
    bx = px + means2D[:, 0] * (0.5 * W)
    by = py + means2D[:, 1] * (0.5 * H)
    # depth order, ties by index (stable sort of the reference)
    order = torch.argsort(depth.detach(), stable=True)
    order = order[visible[order]]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pixx, pixy = xs.reshape(-1), ys.reshape(-1)
    tilex, tiley = torch.floor(pixx / 16), torch.floor(pixy / 16)
    o = order
    dx = bx[o][None, :] - pixx[:, None]
    dy = by[o][None, :] - pixy[:, None]
    power = -0.5 * (conA[o][None] * dx * dx + conC[o][None] * dy * dy) - conB[o][None] * dx * dy
    alpha = torch.clamp_max(opacities.reshape(-1)[o][None] * torch.exp(power), 0.99)
    in_rect = ((tilex[:, None] >= minx[o][None]) & (tilex[:, None] < maxx[o][None]) &
               (tiley[:, None] >= miny[o][None]) & (tiley[:, None] < maxy[o][None]))
    valid = in_rect & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_minus = 1.0 - a_eff
    T_incl = torch.cumprod(one_minus, dim=1)
    T_excl = torch.cat([torch.ones(T_incl.shape[0], 1, dtype=dt), T_incl[:, :-1]], 1)
    stop = valid & ((T_excl * (1 - alpha)).detach() < 0.0001)
    stopped = torch.cumsum(stop.to(torch.int64), dim=1) > 0
    include = valid & ~stopped
    a_inc = torch.where(include, alpha, torch.zeros_like(alpha))
    T_i = torch.cumprod(1.0 - a_inc, dim=1)
    T_e = torch.cat([torch.ones(T_i.shape[0], 1, dtype=dt), T_i[:, :-1]], 1)
    w = a_inc * T_e
    Cpix = w @ rgb[o]
    T_final = T_i[:, -1] if T_i.shape[1] > 0 else torch.ones(H * W, dtype=dt)
    color = Cpix + T_final[:, None] * bg.to(dt)[None]
    aux = dict(depth=depth, px=px, py=py, conic=torch.stack([conA, conB, conC], 1), rgb=rgb,
               T_final=T_final.view(H, W), include=include, order=order,
               rect=torch.stack([minx, miny, maxx, maxy], 1))
    return color.t().reshape(3, H, W), radii, aux
