"""Driver-visible statement of tools/train_check.py: the lego recipe on the synthetic scene converges, and the step executor, the
fused autograd path and the unfused nn.Linear graph reach the same quality.  (PSNR against the analytic ground truth of the synthetic
scene, quarter-resolution held-in view; BASELINE's "final PSNR within 0.1 dB" is a statement about full runs on the real dataset --
here the bar is what 1 500 iterations of a 20-view synthetic scene support: a floor and agreement between the paths.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _train(kind, steps):
    from nerf2mesh_amd import synthetic
    from nerf2mesh_amd.engine import Stage0Engine
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    from nerf2mesh_amd.trainer import Stage0Trainer
    torch.manual_seed(0)
    opt = make_options(O=True, bound=1, dt_gamma=0, iters=steps, fused_mlp=kind != "unfused")
    cls = Stage0Engine if kind == "engine" else Stage0Trainer
    tr = cls(NeRFNetwork(opt), opt, synthetic.make_cameras(20, seed=0), torch.device("cuda", 0), seed=0)
    tr.mark_untrained()
    for _ in range(steps):
        tr.train_step()
    torch.cuda.synchronize()
    return sum(tr.eval_psnr(cam=c) for c in (0, 7)) / 2


def test_psnr_floor_and_agreement_between_the_paths():
    steps = 1500
    psnr = {k: _train(k, steps) for k in ("engine", "fused", "unfused")}
    print("\nPSNR after", steps, "iterations:", {k: round(v, 2) for k, v in psnr.items()})
    assert min(psnr.values()) >= 33.0, psnr          # measured: 36.0 / 36.5 / 36.0 dB
    # same recipe, same draws; the paths differ in fp16 rounding points and summation order only -- and 1 500 Adam steps amplify that
    # into a run-to-run spread of ~0.5 dB (two runs of ONE path differ as much)
    assert max(psnr.values()) - min(psnr.values()) <= 1.2, psnr
