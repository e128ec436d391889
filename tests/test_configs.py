"""End-to-end training steps on the GPU for the BASELINE configurations beyond the lego -O recipe: large-bound cascades
(config 4: `--bound 16`, 5 cascades, inner/outer TV split) and the SDF variant (config 5: NeuS alpha, 7 density evaluations per
sample for the normals, eikonal loss, progressive levels, contraction).  They assert that the whole path runs through the HIP
kernels, stays finite and learns; parity of the individual kernels is covered elsewhere."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _trainer(**kw):
    import torch
    from nerf2mesh_amd import synthetic
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    from nerf2mesh_amd.trainer import Stage0Trainer
    torch.manual_seed(0)
    opt = make_options(O=True, iters=2000, **kw)
    tr = Stage0Trainer(NeRFNetwork(opt), opt, synthetic.make_cameras(12, seed=0), torch.device("cuda", 0), seed=0)
    tr.mark_untrained()
    return tr


def _run(tr, steps):
    import torch
    losses = [float(tr.train_step().detach()) for _ in range(steps)]
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)), losses
    return losses


def test_bound16_cascades_fused_path_learns():
    tr = _trainer(bound=16, dt_gamma=1 / 256, fused_mlp=True)
    assert tr.model.cascade == 5 and tr.amp_adam
    losses = _run(tr, 120)
    assert np.mean(losses[-20:]) < 0.6 * np.mean(losses[:10]), (losses[:10], losses[-20:])
    assert tr.last_num_points > 0 and float(tr.model.density_bitfield.float().sum()) > 0


def test_bound16_unfused_reference_graph_learns():
    tr = _trainer(bound=16, dt_gamma=1 / 256, fused_mlp=False)
    assert not tr.amp_adam
    losses = _run(tr, 120)                                               # the LR ramps up over the first 500 steps (main.py:239)
    assert np.mean(losses[-10:]) < 0.8 * np.mean(losses[:10]), (losses[:10], losses[-10:])


@pytest.mark.parametrize("fused", [True, False])
def test_sdf_stage0_steps_run_and_learn(fused):
    """SDF recipe (config 5).  fused: the field kernels with the raw-sdf head (flag bit 1 of n2m_field_forward/backward), the six
    finite-difference density evaluations as ONE stacked call, FusedAdamAMP (the `variance` scalar rides along as a plain tensor);
    unfused: the nn.Linear graph + torch Adam + GradScaler.  Progressive levels keep TV on its stand-alone kernel for the first half."""
    tr = _trainer(bound=1, dt_gamma=0, sdf=True, fused_mlp=fused, num_rays=512, num_points=2 ** 15)     # small batches: the unfused path does 7 autograd encodes per step
    assert tr.opt.progressive_level and tr.amp_adam == fused
    losses = _run(tr, 60)
    assert tr.model.max_level < 16                                      # progressive levels active (nerf/utils.py:654-655)
    assert np.mean(losses[-10:]) < np.mean(losses[:10]), (losses[:10], losses[-10:])
    assert tr.model.variance.grad is not None or fused                  # the fused optimiser consumes grads without touching .grad


def test_sdf_fused_and_unfused_paths_agree():
    """Same seed, same rays, 40 steps: the two SDF paths end within 15 % of each other's loss (fp16 rounding points differ; the
    finite-difference normals amplify them)."""
    kw = dict(bound=1, dt_gamma=0, sdf=True, num_rays=512, num_points=2 ** 15)
    a = np.mean(_run(_trainer(fused_mlp=True, **kw), 40)[-10:])
    b = np.mean(_run(_trainer(fused_mlp=False, **kw), 40)[-10:])
    assert abs(a - b) <= 0.15 * max(a, b), (a, b)
