"""Stage-0 volume renderer on the HIP operators: the caller-side logic of nerf/renderer.py (render :676-813,
update_extra_state :1074-1149, mark_untrained_grid :985-1071) restated on top of nerf2mesh_amd.raymarching.

Buffer names/shapes follow the reference (`density_grid [cascade, H^3]`, `density_bitfield [cascade*H^3/8]`,
`aabb_train/aabb_infer [6]`) so its checkpoints load.  What changed for ROCm:
* the occupancy refresh walks the grid in Morton order (cell i -> morton3D_invert(i)), so densities land in
  `tmp_grid` without the index scatter of :1118; the cell coordinates are cached across refreshes;
* the inference loop compacts the alive list on the device (raymarching.compact_alive) instead of boolean
  indexing (:798).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import raymarching


def safe_normalize(x, eps=1e-20):
    return x / torch.sqrt(torch.clamp(torch.sum(x * x, -1, keepdim=True), min=eps))   # nerf/utils.py:safe_normalize


def contract(xyzs):
    mag = torch.amax(torch.abs(xyzs), dim=1, keepdim=True)
    return torch.where(mag <= 1, xyzs, xyzs * (2 - 1 / mag) / mag)


class NeRFRenderer(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.real_bound = opt.bound                        # marching bound (world)
        self.bound = 2 if opt.contract else opt.bound      # grid / hashing bound
        self.cascade = 1 + math.ceil(math.log2(self.bound))
        self.grid_size = opt.grid_size
        self.min_near = opt.min_near
        self.density_thresh = opt.density_thresh
        self.max_level = 16
        rb = self.real_bound
        self.register_buffer("aabb_train", torch.tensor([-rb, -rb, -rb, rb, rb, rb], dtype=torch.float32))
        self.register_buffer("aabb_infer", self.aabb_train.clone())
        self.individual_num, self.individual_dim = opt.ind_num, opt.ind_dim
        self.individual_codes = nn.Parameter(torch.randn(opt.ind_num, opt.ind_dim) * 0.1) if opt.ind_dim > 0 else None
        self.cuda_ray = True
        self.register_buffer("density_grid", torch.zeros(self.cascade, self.grid_size ** 3))
        self.register_buffer("density_bitfield", torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
        self.mean_density = 0
        self.iter_density = 0
        self._cell_unit = None      # cached [H^3,3] cell centres in [-1,1], Morton order
        self.glctx = None

    def get_params(self, lr):
        params = []
        if self.individual_codes is not None:
            params.append({"params": self.individual_codes, "lr": self.opt.lr * 0.1, "weight_decay": 0})
        return params

    def density(self, x):
        raise NotImplementedError()

    def reset_extra_state(self):
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0

    def update_aabb(self, aabb):
        if not torch.is_tensor(aabb):
            aabb = torch.as_tensor(aabb).float()
        self.aabb_train = aabb.clamp(-self.real_bound, self.real_bound).to(self.aabb_train.device)
        self.aabb_infer = self.aabb_train.clone()

    # ------------------------------------------------------------------------------------------ stage 0
    def render(self, rays_o, rays_d, index=None, dt_gamma=0, bg_color=None, perturb=False, max_steps=1024, T_thresh=1e-4,
               cam_near_far=None, shading="full", **kwargs):
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, device = rays_o.shape[0], rays_o.device

        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train if self.training else self.aabb_infer, self.min_near)
        if cam_near_far is not None:
            nears = torch.maximum(nears, cam_near_far[:, 0])
            fars = torch.minimum(fars, cam_near_far[:, 1])
        if bg_color is None:
            bg_color = 1
        ind_code = None
        if self.individual_dim > 0:
            ind_code = self.individual_codes[index] if self.training else self.individual_codes[[0]]
        results = {}
        amp = torch.autocast(device_type="cuda", dtype=torch.float16, enabled=bool(self.opt.fp16))

        if self.training:
            xyzs, dirs, ts, rays = raymarching.march_rays_train(rays_o, rays_d, self.real_bound, self.opt.contract, self.density_bitfield,
                                                                self.cascade, self.grid_size, nears, fars, perturb, dt_gamma, max_steps)
            if ind_code is not None and ind_code.shape[0] > 1:
                ind_code = ind_code[raymarching.flatten_rays(rays, xyzs.shape[0]).long()]
            dirs = safe_normalize(dirs)
            with amp:
                sigmas, rgbs, speculars = self(xyzs, dirs, ind_code, shading)
            if self.opt.sdf:
                raw_normal = self.normal(xyzs, self.opt.normal_anneal_epsilon)
                results["normal"] = raw_normal
                true_cos = (dirs * safe_normalize(raw_normal)).sum(-1)
                car = self.opt.cos_anneal_ratio
                iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - car) + F.relu(-true_cos) * car)
                sigmas = self._sdf_to_alpha(sigmas, iter_cos, ts[:, 1])
            weights, weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, ts, rays, T_thresh, self.opt.sdf)
            results.update(num_points=xyzs.shape[0], xyzs=xyzs, speculars=speculars, weights=weights, weights_sum=weights_sum)
        else:
            weights_sum = torch.zeros(N, dtype=torch.float32, device=device)
            depth = torch.zeros(N, dtype=torch.float32, device=device)
            image = torch.zeros(N, 3, dtype=torch.float32, device=device)
            rays_alive = torch.arange(N, dtype=torch.int32, device=device)
            rays_t = nears.clone()
            step = 0
            while step < max_steps:
                n_alive = rays_alive.shape[0]
                if n_alive <= 0:
                    break
                n_step = max(min(N // n_alive, 8), 1)
                xyzs, dirs, ts = raymarching.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.real_bound,
                                                        self.opt.contract, self.density_bitfield, self.cascade, self.grid_size, nears,
                                                        fars, perturb if step == 0 else False, dt_gamma, max_steps)
                dirs = safe_normalize(dirs)
                with amp:
                    sigmas, rgbs, speculars = self(xyzs, dirs, ind_code, shading)
                if self.opt.sdf:
                    true_cos = -F.relu(-(dirs * safe_normalize(self.normal(xyzs))).sum(-1))
                    sigmas = self._sdf_to_alpha(sigmas, true_cos, ts[:, 1])
                raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image, T_thresh,
                                           self.opt.sdf)
                rays_alive = raymarching.compact_alive(rays_alive)
                step += n_step

        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        results["depth"] = depth.view(*prefix)
        results["image"] = image.view(*prefix, 3)
        return results

    def _sdf_to_alpha(self, sdf, cos, dt):
        """NeuS-style alpha from sdf samples (nerf/renderer.py:724-739)."""
        inv_s = torch.exp(self.variance * 10.0).clip(1e-6, 1e6)
        prev_cdf = torch.sigmoid((sdf - cos * dt * 0.5) * inv_s)
        next_cdf = torch.sigmoid((sdf + cos * dt * 0.5) * inv_s)
        return ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).view(-1).clip(0, 1)

    # ------------------------------------------------------------------------------- occupancy-grid upkeep
    def _cells(self):
        """[H^3,3] fp32 cell coordinates 2*c/(H-1)-1 in Morton order (cell i = morton3D_invert(i))."""
        dev = self.density_bitfield.device
        if self._cell_unit is None or self._cell_unit.device != dev:
            H = self.grid_size
            coords = raymarching.morton3D_invert(torch.arange(H ** 3, dtype=torch.int32, device=dev))
            self._cell_unit = 2 * coords.float() / (H - 1) - 1
        return self._cell_unit

    @torch.no_grad()
    def update_extra_state(self, decay=0.95):
        """Occupancy refresh (nerf/renderer.py:1074-1149): one jittered density sample per cell and cascade,
        grid = max(grid*decay, sample) where both are valid, threshold = min(mean, density_thresh), packbits."""
        if self.opt.stage > 0:
            return
        cells = self._cells()
        tmp_grid = torch.empty_like(self.density_grid)
        for cas in range(self.cascade):
            bound = min(2 ** cas, self.bound)
            hgs = bound / self.grid_size
            xyzs = cells * (bound - hgs) + (torch.rand_like(cells) * 2 - 1) * hgs
            with torch.autocast(device_type="cuda", dtype=torch.float16, enabled=bool(self.opt.fp16)):
                sigmas = self.density(xyzs)["sigma"].reshape(-1).detach()
                if self.opt.sdf:
                    inv_s = torch.exp(self.variance * 10.0).clip(1e-6, 1e6)
                    sigmas = torch.sigmoid(-sigmas * inv_s) * inv_s
            tmp_grid[cas] = sigmas.float()
        valid = (self.density_grid >= 0) & (tmp_grid >= 0)
        self.density_grid.copy_(torch.where(valid, torch.maximum(self.density_grid * decay, tmp_grid), self.density_grid))
        self.mean_density = torch.mean(self.density_grid.clamp(min=0)).item()
        self.iter_density += 1
        density_thresh = min(self.mean_density, self.density_thresh)
        self.density_bitfield = raymarching.packbits(self.density_grid, density_thresh, self.density_bitfield)

    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsics, cam_near_far=None, S=64):
        """Marks cells no training camera sees, or that lie outside aabb_train, with -1 (nerf/renderer.py:985-1071).
        poses [B,4,4] cam2world, intrinsics (fx, fy, cx, cy)."""
        fx, fy, cx, cy = intrinsics
        dev = self.density_grid.device
        poses = poses.to(dev)
        B = poses.shape[0]
        cells = self._cells().unsqueeze(0)                    # [1,N,3]; already Morton-ordered
        mask_cam = torch.zeros_like(self.density_grid)
        mask_aabb = torch.zeros_like(self.density_grid)
        for cas in range(self.cascade):
            bound = min(2 ** cas, self.bound)
            hgs = bound / self.grid_size
            pts = cells * (bound - hgs)
            inside = ((pts >= (self.aabb_train[:3] - hgs)).all(-1) & (pts <= (self.aabb_train[3:] + hgs)).all(-1)).reshape(-1)
            mask_aabb[cas] += inside
            for head in range(0, B, S):
                tail = min(head + S, B)
                cam = pts - poses[head:tail, :3, 3].unsqueeze(1)
                cam = cam @ poses[head:tail, :3, :3]
                z = -cam[:, :, 2]
                near = self.opt.min_near if cam_near_far is None else cam_near_far[head:tail, 0].unsqueeze(1)
                seen = (z > near) & (cam[:, :, 0].abs() < cx / fx * z + hgs * 2) & (cam[:, :, 1].abs() < cy / fy * z + hgs * 2)
                mask_cam[cas] += seen.any(0)
        self.density_grid[(mask_cam == 0) | (mask_aabb == 0)] = -1
