"""The stage-0/1 field: two hash-grid encoders + three tiny bias-free MLPs, as nerf/network.py:57-208 defines it.

Parameter and buffer names/shapes are the reference's (`encoder.embeddings [6119864,1]`,
`sigma_net.net.{0,1}.weight [32,19],[1,32]`, `encoder_color.embeddings [6119864,2]`,
`color_net.net.{0,1,2}.weight [64,35],[64,64],[6,64]`, `specular_net.net.{0,1}.weight [32,6],[3,32]`,
SURVEY.md section 8b), so a reference checkpoint's model state_dict loads with strict=True.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .activation import trunc_exp
from .encoding import get_encoder
from .renderer import NeRFRenderer


def x_is_cuda(net):
    return net.encoder.embeddings.is_cuda


class MLP(nn.Module):
    """Linear(+ReLU) stack, `bias=False` everywhere in nerf2mesh (nerf/network.py:10-54; the geom_init /
    weight_norm variants are only reachable from commented-out code there and are not carried over)."""

    def __init__(self, dim_in, dim_out, dim_hidden, num_layers, bias=True):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        self.net = nn.ModuleList([
            nn.Linear(dim_in if l == 0 else dim_hidden, dim_out if l == num_layers - 1 else dim_hidden, bias=bias)
            for l in range(num_layers)])

    def forward(self, x):
        for l, layer in enumerate(self.net):
            x = layer(x)
            if l != self.num_layers - 1:
                x = F.relu(x, inplace=True)
        return x


class NeRFNetwork(NeRFRenderer):
    def __init__(self, opt, specular_dim=3):
        super().__init__(opt)
        if getattr(opt, "tcnn", False):
            raise NotImplementedError("--tcnn is out of scope (north_star: no tiny-cuda-nn)")
        # density branch: C=1 table (fp32 even under autocast, grid.py:45) -> 19 -> 32 -> 1
        self.encoder, self.in_dim_density = get_encoder("hashgrid", level_dim=1, desired_resolution=2048 * self.bound, interpolation="linear")
        self.sigma_net = MLP(3 + self.in_dim_density, 1, 32, 2, bias=False)
        # colour branch: C=2 table (fp16 under autocast) -> 35 -> 64 -> 64 -> 3 diffuse + 3 specular features
        self.encoder_color, self.in_dim_color = get_encoder("hashgrid", level_dim=2, desired_resolution=2048 * self.bound, interpolation="linear")
        self.color_net = MLP(3 + self.in_dim_color + self.individual_dim, 3 + specular_dim, 64, 3, bias=False)
        # view-dependent branch: raw direction (no encoding, nerf/network.py:74) + features -> 32 -> 3
        self.encoder_dir, self.in_dim_dir = get_encoder("None")
        self.specular_net = MLP(specular_dim + self.in_dim_dir, 3, 32, 2, bias=False)
        if self.opt.sdf:
            self.register_parameter("variance", nn.Parameter(torch.tensor(0.3, dtype=torch.float32)))

    def packed_tables(self):
        """[rows, 2] fp32 tensor whose 8-byte rows are {density feature fp32, colour features 2 x fp16}: the layout
        n2m_grid_encode_forward_packed gathers from (one L2 line per vertex pair for both encoders).  Rebuilt when either table
        changed through torch (version counters); optim.FusedAdamAMP refreshes it in its own update pass.  None when the two
        encoders do not share their geometry."""
        e1, e2 = self.encoder, self.encoder_color
        a, b = e1.embeddings, e2.embeddings
        if a.shape[1] != 1 or b.shape[1] != 2 or a.shape[0] != b.shape[0] or not a.is_cuda:
            return None
        key = (a._version, b._version, a.data_ptr(), b.data_ptr())
        if getattr(self, "_packed_key", None) != key or getattr(self, "_packed", None) is None:
            from .gridencoder import same_geometry
            if not same_geometry(e1, e2):
                return None
            with torch.no_grad():
                # `_packed_buffer`: memory the copy must LIVE in (engine.Stage0Engine's peer-store mode exports it to the other ranks, who
                # store their rows into it: a rebuild -- load_state_dict, an in-place edit of a table -- must not move it)
                pk = getattr(self, "_packed_buffer", None)
                if pk is None or pk.shape != (a.shape[0], 2) or pk.device != a.device:
                    pk = torch.empty(a.shape[0], 2, dtype=torch.float32, device=a.device)
                pk[:, 0] = a.detach()[:, 0]
                pk.view(torch.float16)[:, 2:] = b.detach().half()
            self._packed, self._packed_key = pk, key
        return self._packed

    def _can_fuse(self, c=None):
        # SDF: the fused kernels return the raw fp16 sigma_net output (flag bit 1); progressive levels go through max_level
        return bool(getattr(self.opt, "fused_mlp", False)) and c is None and x_is_cuda(self)

    def forward(self, x, d, c=None, shading="full", raw_dirs=False):
        """raw_dirs: `d` are un-normalised ray directions (only valid when _can_fuse(c): the kernel normalises on load)."""
        if self._can_fuse(c):
            from .fused import fused_field
            return fused_field(self, x.view(-1, 3), d.view(-1, 3), shading, normalize_dirs=raw_dirs)
        assert not raw_dirs
        sigma = self.density(x)["sigma"]
        color, specular = self.rgb(x, d, c, shading)
        return sigma, color, specular

    def density(self, x):
        if self._can_fuse() and not x.requires_grad:
            from .fused import fused_density
            return {"sigma": fused_density(self, x.reshape(-1, 3)).view(x.shape[:-1])}     # trunc_exp(.) or, SDF, the raw output
        h = self.encoder(x, bound=self.bound, max_level=self.max_level)
        h = self.sigma_net(torch.cat([x, h], dim=-1))
        sigma = h[..., 0].float() if self.opt.sdf else trunc_exp(h[..., 0])
        return {"sigma": sigma}

    def init_double_sphere(self, r1=0.5, r2=1.5, iters=8192, batch_size=8192):
        """SDF pre-training towards two nested spheres (nerf/network.py:111-132); SDF mode only."""
        if not self.opt.sdf:
            return
        opt = torch.optim.Adam(list(self.parameters()), lr=1e-3)
        dev = self.embeddings_device()
        for _ in range(iters):
            xyzs = torch.rand(batch_size, 3, device=dev) * 2 * self.bound - self.bound
            d = torch.norm(xyzs, p=2, dim=-1)
            gt = torch.where(d < (r1 + r2) / 2, d - r1, r2 - d)
            loss = F.mse_loss(self.density(xyzs)["sigma"], gt)
            opt.zero_grad()
            loss.backward()
            opt.step()

    def embeddings_device(self):
        return self.encoder.embeddings.device

    def normal(self, x, epsilon=1e-4):
        """Central finite differences of the density/SDF: 6 extra density() evaluations (nerf/network.py:143-154) -- here ONE call on
        the six offset copies stacked into a [6M, 3] batch (same points, same clamp, same arithmetic per point; one encode + one MLP
        launch forward and backward instead of six)."""
        M = x.shape[0]
        off = torch.zeros(6, 1, 3, device=x.device, dtype=x.dtype)
        for axis in range(3):
            off[2 * axis, 0, axis] = epsilon
            off[2 * axis + 1, 0, axis] = -epsilon
        pts = (x.unsqueeze(0) + off).clamp(-self.bound, self.bound).reshape(6 * M, 3)
        s = self.density(pts)["sigma"].view(6, M)
        return torch.stack([0.5 * (s[0] - s[1]) / epsilon, 0.5 * (s[2] - s[3]) / epsilon, 0.5 * (s[4] - s[5]) / epsilon], dim=-1)

    def geo_feat(self, x, c=None):
        h = self.encoder_color(x, bound=self.bound, max_level=self.max_level)
        h = torch.cat([x, h], dim=-1)
        if c is not None:
            h = torch.cat([h, c.repeat(x.shape[0], 1) if c.shape[0] == 1 else c], dim=-1)
        return torch.sigmoid(self.color_net(h))

    def rgb(self, x, d, c=None, shading="full"):
        if self._can_fuse(c) and not x.requires_grad:
            from .fused import fused_color
            return fused_color(self, x.reshape(-1, 3), d.reshape(-1, 3), shading)
        geo_feat = self.geo_feat(x, c)
        diffuse = geo_feat[..., :3]
        if shading == "diffuse":
            return diffuse, None
        d = self.encoder_dir(d)
        specular = torch.sigmoid(self.specular_net(torch.cat([d, geo_feat[..., 3:]], dim=-1)))
        color = specular if shading == "specular" else (specular + diffuse).clamp(0, 1)
        return color, specular

    def get_params(self, lr):
        params = super().get_params(lr)
        params.extend([{"params": m.parameters(), "lr": lr}
                       for m in (self.encoder, self.encoder_color, self.sigma_net, self.color_net, self.specular_net)])
        if self.opt.sdf:
            params.append({"params": self.variance, "lr": lr * 0.1})
        return params
