"""Seeded synthetic Gaussian scenes.  Shapes and activations follow
scene/gaussian_model.py:132-157,172-199 of the reference: xyz uniform in the AABB
(dataset/zjumocap.py:419-427), scale = sqrt(clamp_min(distCUDA2, 1e-7)) (gaussian_model.py:186-187),
opacity = sigmoid(inverse_sigmoid(0.1) + noise) (:191), SH DC = RGB2SH(colour) (:175-180).
"""
import math

import torch

C0 = 0.28209479177387814  # utils/sh_utils.py:27


def rgb_to_sh(rgb):
    return (rgb - 0.5) / C0  # utils/sh_utils.py:163


def inverse_sigmoid(x):
    return math.log(x / (1 - x))


class GaussianCloud(object):
    """Post-activation Gaussian state: the tensors `render()` hands to the rasterizer."""

    FIELDS = ("xyz", "scales", "rotations", "opacity", "shs")

    def __init__(self, xyz, scales, rotations, opacity, shs, sh_degree):
        self.xyz, self.scales, self.rotations, self.opacity, self.shs = xyz, scales, rotations, opacity, shs
        self.sh_degree = sh_degree

    @property
    def num(self):
        return self.xyz.shape[0]

    def to(self, device):
        return GaussianCloud(*[getattr(self, f).to(device) for f in self.FIELDS], self.sh_degree)

    def covariance6(self, scale_modifier=1.0):
        """cov3D_precomp (N,6) = strip_symmetric((R S)(R S)^T): GaussianModel.get_covariance
        (scene/gaussian_model.py:154-157) -- with the deformer's `rotation_precomp` (N,3,3) when the cloud carries
        one, else with the quaternions -- through the fused HIP op (gsplat_mi355.prepass, row N3)."""
        from .prepass import build_covariance_from_scaling_rotation
        rot = getattr(self, "rotation_precomp", None)
        return build_covariance_from_scaling_rotation(self.scales, scale_modifier, rot if rot is not None else self.rotations)

    def pack(self):
        """One flat fp32 buffer [xyz | scales | rotations | opacity | shs]: the broadcast payload of
        SURVEY.md 8e (tensors 1-6 of GaussianModel.capture(), scene/gaussian_model.py:98-112)."""
        return torch.cat([getattr(self, f).reshape(-1) for f in self.FIELDS])

    @staticmethod
    def packed_numel(n, sh_degree):
        m = (sh_degree + 1) ** 2
        return n * (3 + 3 + 4 + 1 + 3 * m)

    @staticmethod
    def unpack(flat, n, sh_degree):
        m = (sh_degree + 1) ** 2
        sizes = [3 * n, 3 * n, 4 * n, n, 3 * m * n]
        parts = torch.split(flat, sizes)
        return GaussianCloud(parts[0].view(n, 3).clone(), parts[1].view(n, 3).clone(), parts[2].view(n, 4).clone(),
                             parts[3].view(n, 1).clone(), parts[4].view(n, m, 3).clone(), sh_degree)


# a standing figure of capsules (segment end points, radius) inside the [-1, 1]^3 box, y up: torso, head, arms, legs
_BODY = [((0.0, -0.10, 0.0), (0.0, 0.50, 0.0), 0.17), ((0.0, 0.64, 0.0), (0.0, 0.74, 0.0), 0.11),
         ((-0.22, 0.45, 0.0), (-0.55, 0.00, 0.05), 0.05), ((0.22, 0.45, 0.0), (0.55, 0.00, 0.05), 0.05),
         ((-0.10, -0.15, 0.0), (-0.16, -0.92, 0.0), 0.08), ((0.10, -0.15, 0.0), (0.16, -0.92, 0.0), 0.08)]


def _body_points(n, g):
    """Points on the surface of the capsule figure (+ 2 mm of noise): the shape of a TRAINED avatar cloud -- the
    Gaussians on a thin shell that covers a fraction of the image -- where synthetic_cloud's default is the reference's
    initial state (uniform in the box)."""
    p0 = torch.tensor([c[0] for c in _BODY])
    p1 = torch.tensor([c[1] for c in _BODY])
    r = torch.tensor([c[2] for c in _BODY])
    length = (p1 - p0).norm(dim=1)
    area = 2 * math.pi * r * (length + 2 * r)  # cylinder + the two half-sphere caps
    which = torch.multinomial(area / area.sum(), n, replacement=True, generator=g)
    u = torch.rand(n, generator=g) * (length + 2 * r)[which]  # position along the capsule, caps unrolled
    axis = ((p1 - p0) / length[:, None])[which]
    helper = torch.where(axis[:, 1:2].abs() < 0.9, torch.tensor([[0.0, 1.0, 0.0]]), torch.tensor([[1.0, 0.0, 0.0]]))
    e1 = torch.linalg.cross(axis, helper.expand_as(axis))
    e1 = e1 / e1.norm(dim=1, keepdim=True)
    e2 = torch.linalg.cross(axis, e1)
    phi = torch.rand(n, generator=g) * 2 * math.pi
    ring = torch.cos(phi)[:, None] * e1 + torch.sin(phi)[:, None] * e2
    rr, ll = r[which], length[which]
    t = (u - rr).clamp(min=0.0) .clamp(max=1e9)
    t = torch.minimum(t, ll)                      # foot point on the segment
    over = torch.where(u < rr, u - rr, torch.where(u > rr + ll, u - rr - ll, torch.zeros_like(u)))  # signed run onto a cap
    theta = over / rr                             # angle from the rim towards the pole, within +-1 rad of the rim
    normal = torch.cos(theta)[:, None] * ring + torch.sin(theta)[:, None] * axis
    pts = p0[which] + t[:, None] * axis + rr[:, None] * normal
    return pts + 0.002 * torch.randn(n, 3, generator=g)


def synthetic_cloud(n, sh_degree=3, seed=0, dist2_fn=None, heavy_tail=0.0, device="cpu", layout="box"):
    """SURVEY.md 8d scene.  `dist2_fn(points[N,3]) -> [N]` is the distCUDA2 implementation to use
    (the HIP one on a GPU; tests on CPU pass the oracle's).  `heavy_tail` > 0 multiplies the scales
    of that fraction of the Gaussians by 4 (BASELINE config 5: tile-overflow / sort stress).  `layout`: "box" = the
    reference's initial state (uniform in the box, opacity around 0.1); "body" = the shape of a trained avatar (a thin
    shell on a capsule figure, mostly opaque)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    if layout == "body":
        xyz = _body_points(n, g)
    else:
        xyz = torch.rand(n, 3, generator=g) * 2 - 1
    aniso = torch.exp(torch.randn(n, 3, generator=g) * 0.3)
    rot = torch.randn(n, 4, generator=g)
    rot = rot / rot.norm(dim=1, keepdim=True)
    opacity = torch.sigmoid((2.0 if layout == "body" else inverse_sigmoid(0.1)) + torch.randn(n, 1, generator=g) * (1.5 if layout == "body" else 1.0))
    m = (sh_degree + 1) ** 2
    shs = torch.randn(n, m, 3, generator=g) * 0.05
    shs[:, 0, :] = rgb_to_sh(torch.rand(n, 3, generator=g))
    tail = torch.rand(n, generator=g)
    xyz = xyz.to(device)
    if dist2_fn is None:
        from simple_knn._C import distCUDA2 as dist2_fn
    d2 = torch.as_tensor(dist2_fn(xyz)).to(device=device, dtype=torch.float32)
    base = torch.sqrt(torch.clamp_min(d2, 1e-7))[:, None]
    scales = base * aniso.to(device)
    if heavy_tail > 0:
        scales = torch.where((tail < heavy_tail).to(device)[:, None], scales * 4.0, scales)
    return GaussianCloud(xyz.contiguous(), scales.contiguous(), rot.to(device).contiguous(),
                         opacity.to(device).contiguous(), shs.to(device).contiguous(), sh_degree)
