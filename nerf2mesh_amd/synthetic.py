"""Synthetic stand-in for nerf_synthetic/lego (no dataset ships with the container).

* cameras: 100 poses on the upper hemisphere of a sphere of radius 4.0311289 * 0.8 (lego's camera distance
  times `--scale 0.8`, scripts/runall_syn.sh:1), OpenGL convention (-z forward), 800x800, focal 1111.111
  (camera_angle_x = 0.6911112, nerf/provider.py:251-263).
* rays: exactly the arithmetic of get_rays (nerf/utils.py:242-290): pixel centres at +0.5, un-normalised
  directions ((i-cx)/fx, -(j-cy)/fy, -1) @ R^T, origin = pose[:3,3].
* scene: an analytic "lego-like" union of axis-aligned boxes inside [-0.6,0.6]^3 (z up), used both as the
  ground truth (first-hit shading -> RGBA pixels) and to rasterise an occupancy grid in Morton order.

Everything is plain torch and runs on CPU or on the GPU (device of the inputs).
"""
import math

import torch

LEGO_RADIUS = 4.0311289 * 0.8
LEGO_HW = 800
LEGO_FOCAL = 0.5 * LEGO_HW / math.tan(0.5 * 0.6911112070083618)

# (xmin, ymin, zmin, xmax, ymax, zmax, r, g, b)
_BOXES = [
    (-0.55, -0.35, -0.45, 0.55, 0.35, -0.37, 0.55, 0.55, 0.58),   # base plate
    (-0.40, -0.22, -0.37, 0.30, 0.22, -0.12, 0.95, 0.75, 0.10),   # chassis
    (-0.05, -0.18, -0.12, 0.28, 0.18, 0.16, 0.92, 0.70, 0.08),    # cab
    (-0.40, -0.06, -0.12, -0.05, 0.06, 0.02, 0.25, 0.25, 0.28),   # boom base
    (-0.58, -0.04, 0.02, -0.20, 0.04, 0.10, 0.90, 0.72, 0.10),    # boom
    (-0.60, -0.16, -0.20, -0.52, 0.16, 0.10, 0.35, 0.35, 0.38),   # bucket
    (-0.46, -0.30, -0.45, -0.22, -0.22, -0.25, 0.12, 0.12, 0.12), # wheel fl
    (-0.46, 0.22, -0.45, -0.22, 0.30, -0.25, 0.12, 0.12, 0.12),   # wheel fr
    (0.08, -0.30, -0.45, 0.32, -0.22, -0.25, 0.12, 0.12, 0.12),   # wheel rl
    (0.08, 0.22, -0.45, 0.32, 0.30, -0.25, 0.12, 0.12, 0.12),     # wheel rr
    (0.02, -0.10, 0.16, 0.10, -0.02, 0.22, 0.80, 0.10, 0.10),     # beacon
    (0.30, -0.20, -0.37, 0.50, 0.20, -0.30, 0.30, 0.30, 0.75),    # rear step
]


# "garden": the same object on a lawn inside a hedged yard that reaches |x| = 8 -- a stand-in for an unbounded mip-NeRF-360 capture
# (BASELINE config 4: --bound 16, 5 cascades): content in cascades 0..3, a sparse-point AABB much tighter than the bound, per-camera depth
# ranges (colmap_provider.py:223,270).  Same cameras as the lego stand-in (they circle the centre piece like the garden's table).
_GARDEN_EXTRA = [
    (-8.00, -8.00, -0.62, 8.00, 8.00, -0.45, 0.30, 0.46, 0.20),   # lawn (its top carries the base plate)
    (-7.60, -7.60, -0.45, -6.80, 7.60, 1.20, 0.16, 0.36, 0.14),   # hedge west
    (6.80, -7.60, -0.45, 7.60, 7.60, 1.00, 0.18, 0.38, 0.15),     # hedge east
    (-6.80, 6.80, -0.45, 6.80, 7.60, 1.40, 0.15, 0.34, 0.13),     # hedge north
    (-6.80, -7.60, -0.45, 6.80, -6.80, 0.90, 0.17, 0.37, 0.16),   # hedge south
    (-1.30, -1.10, -0.45, 1.30, 1.10, -0.42, 0.62, 0.48, 0.32),   # table top under the object
    (2.60, 1.70, -0.45, 3.10, 2.20, 1.80, 0.36, 0.25, 0.15),      # tree trunk
    (1.90, 1.00, 1.80, 3.80, 2.90, 3.20, 0.20, 0.42, 0.18),       # crown
    (-3.90, -2.40, -0.45, -3.10, -1.60, 0.35, 0.55, 0.52, 0.50),  # planter
    (-2.20, 3.40, -0.45, -0.60, 4.10, 0.05, 0.45, 0.30, 0.20),    # bench
    (4.40, -3.60, -0.45, 5.20, -2.80, 0.60, 0.58, 0.56, 0.52),    # stone
]
SCENES = {"lego": _BOXES, "garden": _BOXES + _GARDEN_EXTRA}


def boxes(device="cpu", scene="lego"):
    return torch.tensor(SCENES[scene], dtype=torch.float32, device=device)


def scene_points(scene="lego", per_edge=5):
    """A sparse point cloud of the scene (box surface lattice points) -- the stand-in for colmap's points3D."""
    bx = boxes("cpu", scene)
    lin = torch.linspace(0, 1, per_edge)
    a, b, c = torch.meshgrid(lin, lin, lin, indexing="ij")
    t = torch.stack([a.reshape(-1), b.reshape(-1), c.reshape(-1)], -1)
    t = t[((t == 0) | (t == 1)).any(-1)]                                           # surface lattice only
    return (bx[:, None, 0:3] + t[None] * (bx[:, None, 3:6] - bx[:, None, 0:3])).reshape(-1, 3)


def pts_aabb(scene="lego"):
    """[6] min / max of the sparse points (colmap_provider.py:223), what main.py:234-235 hands to update_aabb."""
    p = scene_points(scene)
    return torch.cat([p.min(0).values, p.max(0).values])


def cam_near_far(poses, scene="lego", H=LEGO_HW, W=LEGO_HW, focal=LEGO_FOCAL):
    """[V,2] per-camera (min, max) depth of the sparse points the camera observes, depth = (P[:3,3] - pts) @ P[:3,2]
    (colmap_provider.py:243-270: the points with a key point inside the image; here: the lattice points that project into the frame)."""
    p = scene_points(scene, per_edge=9).to(poses.device)
    rel = p[None] - poses[:, None, :3, 3]                                          # [V, P, 3]
    xc, yc = (rel * poses[:, None, :3, 0]).sum(-1), (rel * poses[:, None, :3, 1]).sum(-1)
    depth = -(rel * poses[:, None, :3, 2]).sum(-1)
    front = (depth > 0) & (xc.abs() <= 0.5 * W / focal * depth) & (yc.abs() <= 0.5 * H / focal * depth)
    big = torch.finfo(torch.float32).max
    near = torch.where(front, depth, torch.full_like(depth, big)).min(1).values
    far = torch.where(front, depth, torch.zeros_like(depth)).max(1).values
    return torch.stack([near, far], 1).float().contiguous()


def make_cameras(n=100, radius=LEGO_RADIUS, seed=0, device="cpu"):
    """[n,4,4] camera-to-world poses looking at the origin from the upper hemisphere (z up)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, generator=g)
    v = torch.rand(n, generator=g)
    theta = 2 * math.pi * u
    elev = torch.deg2rad(5.0 + 75.0 * v)                       # 5..80 degrees above the horizon
    c = torch.stack([torch.cos(elev) * torch.cos(theta), torch.cos(elev) * torch.sin(theta), torch.sin(elev)], -1) * radius
    fwd = -c / c.norm(dim=-1, keepdim=True)                    # camera looks along -z_cam
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = torch.cross(fwd, up, dim=-1)
    right = right / right.norm(dim=-1, keepdim=True)
    true_up = torch.cross(right, fwd, dim=-1)
    poses = torch.eye(4).repeat(n, 1, 1)
    poses[:, :3, 0] = right
    poses[:, :3, 1] = true_up
    poses[:, :3, 2] = -fwd
    poses[:, :3, 3] = c
    return poses.to(device)


def rays_from_pixels(poses, cam_idx, pix, H=LEGO_HW, W=LEGO_HW, focal=LEGO_FOCAL):
    """get_rays restated (nerf/utils.py:242-290). pix = flat pixel index j*W + i; returns rays_o, rays_d [N,3]."""
    i = (pix % W).float() + 0.5
    j = torch.div(pix, W, rounding_mode="floor").float() + 0.5
    cx, cy = W / 2, H / 2
    dirs = torch.stack([(i - cx) / focal, -(j - cy) / focal, -torch.ones_like(i)], -1)          # [N,3]
    P = poses[cam_idx]                                                                            # [N,4,4], one gather
    # d = R @ dir as three broadcast FMAs (a [N,1,3]x[N,3,3] bmm goes through the batched-GEMM library path)
    rays_d = dirs[:, 0:1] * P[:, :3, 0] + dirs[:, 1:2] * P[:, :3, 1] + dirs[:, 2:3] * P[:, :3, 2]
    rays_o = P[:, :3, 3]
    return rays_o.contiguous(), rays_d.contiguous()


def random_rays(poses, N, generator=None, H=LEGO_HW, W=LEGO_HW, focal=LEGO_FOCAL):
    """N random pixels over all images (random_image_batch, nerf/provider.py:302-303)."""
    dev = poses.device
    cam = torch.randint(0, poses.shape[0], (N,), device=dev, generator=generator)
    pix = torch.randint(0, H * W, (N,), device=dev, generator=generator)
    o, d = rays_from_pixels(poses, cam, pix, H, W, focal)
    return o, d


def crop_rays(poses, cam=0, size=64, H=LEGO_HW, W=LEGO_HW, focal=LEGO_FOCAL):
    """Central size x size crop of one view (BASELINE config 1)."""
    dev = poses.device
    j0, i0 = (H - size) // 2, (W - size) // 2
    jj, ii = torch.meshgrid(torch.arange(j0, j0 + size, device=dev), torch.arange(i0, i0 + size, device=dev), indexing="ij")
    pix = (jj * W + ii).reshape(-1)
    return rays_from_pixels(poses, torch.full_like(pix, cam), pix, H, W, focal)


def scene_inside(xyz, bx=None):
    """[...,3] -> bool: inside the union of boxes."""
    bx = boxes(xyz.device) if bx is None else bx
    p = xyz.unsqueeze(-2)                                           # [...,1,3]
    return ((p >= bx[:, 0:3]) & (p <= bx[:, 3:6])).all(-1).any(-1)


_SHADE = {}


def _shade(device):
    if device not in _SHADE:          # built once: torch.tensor(..., device=cuda) is a blocking host-to-device copy
        _SHADE[device] = torch.tensor([0.80, 0.65, 1.00], device=device)
    return _SHADE[device]


def preload_images(poses, bx=None, H=LEGO_HW, W=LEGO_HW, focal=LEGO_FOCAL, chunk=1 << 21):
    """All views' ground-truth RGBA, [n_views, H*W, 4] fp32 resident on the device -- the stand-in for the reference's
    preloaded `self.images` (nerf/provider.py:224-233 keeps the whole training set on the GPU and gathers pixels from it)."""
    dev = poses.device
    n = poses.shape[0]
    out = torch.empty(n, H * W, 4, dtype=torch.float32, device=dev)
    pix = torch.arange(H * W, device=dev)
    for v in range(n):
        for s in range(0, H * W, chunk):
            p = pix[s:s + chunk]
            o, d = rays_from_pixels(poses, torch.full_like(p, v), p, H, W, focal)
            out[v, s:s + chunk] = render_gt(o, d, bx)
    return out


def random_batch(poses, images, N, generator=None, H=LEGO_HW, W=LEGO_HW, focal=LEGO_FOCAL):
    """N random pixels over all images with their ground truth gathered from the preloaded set
    (random_image_batch, nerf/provider.py:302-303 + :330 `torch.gather(images, 1, ...)`).  On the GPU the ray construction and
    the gather are one kernel (n2m_get_rays); rays_from_pixels is the torch statement of the same arithmetic."""
    dev = poses.device
    cam = torch.randint(0, poses.shape[0], (N,), device=dev, generator=generator)
    pix = torch.randint(0, H * W, (N,), device=dev, generator=generator)
    if dev.type == "cuda" and poses.dtype == torch.float32 and poses.is_contiguous() and images.is_contiguous():
        from . import _lib as L
        o = torch.empty(N, 3, device=dev)
        d = torch.empty(N, 3, device=dev)
        rgba = torch.empty(N, 4, device=dev)
        L.call("n2m_get_rays", L.ptr(poses), L.ptr(cam), L.ptr(pix), N, H, W, float(focal), float(focal), W / 2, H / 2, L.ptr(images), L.ptr(o),
               L.ptr(d), L.ptr(rgba), L.stream())
        return o, d, rgba
    o, d = rays_from_pixels(poses, cam, pix, H, W, focal)
    return o, d, images[cam, pix]


def batch_from_uniforms(poses, images, u, aabb, min_near, H=LEGO_HW, W=LEGO_HW, focal=LEGO_FOCAL, out=None, counter=None, cam_near_far=None):
    """A training batch from ONE tensor of uniforms u [N,6] in [0,1): view = floor(u0 V), pixel = floor(u1 H W) (N random pixels over
    random views: random_image_batch, nerf/provider.py:302-303 + nerf/utils.py:271), rays and ground truth like random_batch, near/far
    of the aabb, march jitter = u2, random background = u3..u5 (nerf/utils.py:649-652).  Returns (rays_o, rays_d, rgba, nears, fars,
    noises, bg).  On the GPU this is one kernel (n2m_batch_rays) writing into `out` (a tuple of preallocated tensors) when given; the
    torch statement below is the same arithmetic.  Both training drivers draw their batches through this function, so they consume
    identical numbers whatever their scheduling."""
    dev = poses.device
    N, V = u.shape[0], poses.shape[0]
    if dev.type == "cuda":
        from . import _lib as L
        if out is None:
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            out = (f(N, 3), f(N, 3), f(N, 4), f(N), f(N), f(N), f(N, 3))
        o, d, rgba, nears, fars, noises, bg = out
        L.call("n2m_batch_rays_cnf", L.ptr(poses), L.ptr(u), V, N, H, W, float(focal), float(focal), W / 2, H / 2, L.ptr(images), L.ptr(aabb),
               float(min_near), L.ptr(o), L.ptr(d), L.ptr(rgba), L.ptr(nears), L.ptr(fars), L.ptr(noises), L.ptr(bg), L.ptr(counter),
               L.ptr(cam_near_far), L.stream())
        return o, d, rgba, nears, fars, noises, bg
    cam = (u[:, 0] * V).long().clamp(max=V - 1)
    pix = (u[:, 1] * (H * W)).long().clamp(max=H * W - 1)
    o, d = rays_from_pixels(poses, cam, pix, H, W, focal)
    inv = 1.0 / d
    lo, hi = (aabb[:3] - o) * inv, (aabb[3:] - o) * inv
    tn, tf = torch.minimum(lo, hi).amax(-1), torch.maximum(lo, hi).amin(-1)
    miss = tn > tf
    big = torch.finfo(torch.float32).max
    nears = torch.where(miss, torch.full_like(tn, big), tn.clamp(min=min_near))
    fars = torch.where(miss, torch.full_like(tf, big), tf)
    if cam_near_far is not None:                       # nerf/renderer.py:689-691 with the per-ray pairs of colmap_provider.py:563-565
        nears = torch.maximum(nears, cam_near_far[cam, 0])
        fars = torch.minimum(fars, cam_near_far[cam, 1])
    return o, d, images[cam, pix], nears, fars, u[:, 2].contiguous(), u[:, 3:6].contiguous()


def render_gt(rays_o, rays_d, bx=None):
    """First-hit shading of the box scene. Returns rgba [N,4] (alpha 0 = background)."""
    bx = boxes(rays_o.device) if bx is None else bx
    o, d = rays_o.unsqueeze(1), rays_d.unsqueeze(1)                 # [N,1,3]
    inv = 1.0 / torch.where(d.abs() < 1e-12, torch.full_like(d, 1e-12), d)
    t0, t1 = (bx[:, 0:3] - o) * inv, (bx[:, 3:6] - o) * inv         # [N,K,3]
    tn, tf = torch.minimum(t0, t1), torch.maximum(t0, t1)
    tnear, axis = tn.max(-1)
    tfar = tf.min(-1).values
    hit = (tnear <= tfar) & (tfar > 0)
    tnear = torch.where(hit, tnear, torch.full_like(tnear, float("inf")))
    t, k = tnear.min(-1)                                            # first box
    any_hit = torch.isfinite(t)
    ax = axis.gather(1, k.unsqueeze(1)).squeeze(1)
    shade = bx.new_tensor([0.80, 0.65, 1.00]) if rays_o.device.type == "cpu" else _shade(rays_o.device)
    shade = shade[ax]                                               # per-axis "lambert"
    rgb = bx[k, 6:9] * shade.unsqueeze(-1)
    rgba = torch.cat([rgb, torch.ones_like(rgb[:, :1])], -1)
    return torch.where(any_hit.unsqueeze(-1), rgba, torch.zeros_like(rgba))


def _part1by2(v):
    v = v & 0x3FF
    v = (v | (v << 16)) & 0xFF0000FF
    v = (v | (v << 8)) & 0x0F00F00F
    v = (v | (v << 4)) & 0xC30C30C3
    v = (v | (v << 2)) & 0x49249249
    return v


def morton3D_torch(coords):
    """int64 [N,3] -> int64 [N]; x in bit 0 (same code as raymarching.morton3D, used here for CPU-side setup)."""
    c = coords.long()
    return _part1by2(c[:, 0]) | (_part1by2(c[:, 1]) << 1) | (_part1by2(c[:, 2]) << 2)


def scene_density_grid(H=128, cascade=1, bound=1.0, sigma=50.0, device="cpu"):
    """density_grid [cascade, H^3] fp32 in Morton order: sigma where the cell centre region touches the scene.

    Cell (x,y,z) of cascade c covers [-b,b]^3 with b = min(2^c, bound), like update_extra_state
    (nerf/renderer.py:1094-1118) lays the grid out.  A cell is occupied if any of its 8 corners or its centre
    is inside a box (conservative enough for a marcher test scene)."""
    ar = torch.arange(H, device=device)
    xx, yy, zz = torch.meshgrid(ar, ar, ar, indexing="ij")
    coords = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)
    idx = morton3D_torch(coords)
    grid = torch.zeros(cascade, H ** 3, dtype=torch.float32, device=device)
    bx = boxes(device)
    offs = torch.tensor([[0.5, 0.5, 0.5]] + [[a, b, c] for a in (0.0, 1.0) for b in (0.0, 1.0) for c in (0.0, 1.0)], device=device)
    for cas in range(cascade):
        b = min(2.0 ** cas, bound)
        occ = torch.zeros(H ** 3, dtype=torch.bool, device=device)
        for o in offs:
            p = ((coords.float() + o) / H * 2 - 1) * b
            occ |= scene_inside(p, bx)
        grid[cas, idx] = occ.float() * sigma
    return grid


def scene_mesh(target_faces=300000, device="cpu"):
    """Triangle mesh of the box scene (every box face tessellated into k x k quads so that the total is close to
    `target_faces`, the reference's decimate_target, main.py:101).  Returns vertices [V,3] f32, faces [F,3] i32.
    Boxes keep their own vertices (the union is not remeshed); each box is individually watertight."""
    bx = torch.tensor(_BOXES, dtype=torch.float32)
    n_box = bx.shape[0]
    k = max(1, int(round(math.sqrt(target_faces / (n_box * 6 * 2)))))
    lin = torch.linspace(0, 1, k + 1)
    uu, vv = torch.meshgrid(lin, lin, indexing="ij")
    uv = torch.stack([uu.reshape(-1), vv.reshape(-1)], -1)                     # [(k+1)^2, 2]
    ii, jj = torch.meshgrid(torch.arange(k), torch.arange(k), indexing="ij")
    q = (ii * (k + 1) + jj).reshape(-1)
    quad = torch.stack([q, q + (k + 1), q + (k + 2), q, q + (k + 2), q + 1], -1).reshape(-1, 3)   # two triangles per cell
    verts, faces, base = [], [], 0
    for b in range(n_box):
        lo, hi = bx[b, 0:3], bx[b, 3:6]
        for axis in range(3):
            a1, a2 = (axis + 1) % 3, (axis + 2) % 3
            for side in (0, 1):
                p = torch.zeros(uv.shape[0], 3)
                p[:, axis] = hi[axis] if side else lo[axis]
                p[:, a1] = lo[a1] + uv[:, 0] * (hi[a1] - lo[a1])
                p[:, a2] = lo[a2] + uv[:, 1] * (hi[a2] - lo[a2])
                verts.append(p)
                f = quad + base
                faces.append(f if side else f[:, [0, 2, 1]])                   # outward winding
                base += uv.shape[0]
    return torch.cat(verts).to(device), torch.cat(faces).int().to(device)


def mvp_matrix(pose, H=LEGO_HW, W=LEGO_HW, focal=LEGO_FOCAL, near=0.05, far=100.0):
    """projection @ inverse(pose) with the reference's projection (nerf/provider.py:266-276: y flipped, OpenGL z)."""
    proj = torch.tensor([[2 * focal / W, 0, 0, 0],
                         [0, -2 * focal / H, 0, 0],
                         [0, 0, -(far + near) / (far - near), -(2 * far * near) / (far - near)],
                         [0, 0, -1, 0]], dtype=torch.float32, device=pose.device)
    return proj @ torch.inverse(pose)
