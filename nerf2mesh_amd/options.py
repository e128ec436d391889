"""The reference's `opt` namespace: defaults of main.py:12-125 and its derived overrides (main.py:127-181)."""
from types import SimpleNamespace

_DEFAULTS = dict(
    O=False, workspace="workspace", seed=0, stage=0, ckpt="latest", fp16=False, sdf=False, tcnn=False, progressive_level=False,
    bound=2.0, scale=-1, offset=[0, 0, 0], mesh="", min_near=0.05,
    iters=30000, lr=1e-2, lr_vert=1e-4, pos_gradient_boost=1, cuda_ray=True, max_steps=1024, update_extra_interval=16,
    max_ray_batch=4096, grid_size=128, mark_untrained=False, dt_gamma=1 / 256, density_thresh=10, diffuse_step=1000,
    diffuse_only=False, background="random", enable_offset_nerf_grad=False,
    num_rays=4096, adaptive_num_rays=False, num_points=2 ** 18,
    lambda_density=0, lambda_entropy=0, lambda_tv=1e-8, lambda_depth=0.1, lambda_specular=1e-5, lambda_eikonal=0.1,
    lambda_rgb=1, lambda_mask=0.1,
    lambda_lpips=0, lambda_offsets=0.1, lambda_lap=0.001, lambda_normal=0, lambda_edgelen=0,
    contract=False, patch_size=1, trainable_density_grid=False, color_space="srgb", ind_dim=0, ind_num=500,
    ssaa=2, texture_size=4096, refine=False, gui=False,
    cos_anneal_ratio=1.0, normal_anneal_epsilon=1e-4,
    fused_mlp=False,     # opt-in: fused MFMA field kernels (nerf2mesh_amd/fused.py) instead of nn.Linear calls
    enable_cam_near_far=False,     # main.py:40 (colmap mode): clamp every ray to its camera's sparse-point depth range
    scene="lego",        # not a reference option: which synthetic stand-in the drivers render (nerf2mesh_amd/synthetic.py: "lego" | "garden")
)


def make_options(**overrides):
    """opt = make_options(O=True, bound=1, dt_gamma=0) reproduces `main.py -O --bound 1 --dt_gamma 0`."""
    unknown = set(overrides) - set(_DEFAULTS)
    if unknown:
        raise TypeError(f"unknown option(s): {sorted(unknown)}")
    o = SimpleNamespace(**{**_DEFAULTS, **overrides})
    o.cuda_ray = True                                   # main.py:127
    if o.O:                                             # main.py:129-136
        o.fp16 = True
        o.mark_untrained = True
        o.adaptive_num_rays = True
        o.refine = True
    if o.sdf:                                           # main.py:138-153
        o.density_thresh = 0.001
        if o.stage == 0:
            o.progressive_level = True
        if o.bound > 1:
            o.contract = True
        o.enable_offset_nerf_grad = True
    if o.contract:                                      # main.py:155-157
        o.mark_untrained = False
    return o
