"""Driver-visible statement of tools/train_check.py: the lego recipe on the synthetic scene converges on every driver of the kernels -- the
step executor, the fused autograd path and the unfused nn.Linear graph.  (PSNR against the analytic ground truth of the synthetic scene,
quarter-resolution held-in views.)  The parity statement -- executor against the REFERENCE's own loop, held-out views, EMA weights, the bar
derived from the reference's own run-to-run spread -- is tests/test_run_parity.py; the executor's own run-to-run spread is zero (asserted there)."""
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
    # Same recipe, same draws; the three differ in fp16 rounding points and summation order only.  500 steps behind the switch to full
    # shading (diffuse_step = 1 000) the run is in its chaotic recovery: two runs of the REFERENCE loop that differ only in their draws sit
    # 1.5 dB rms apart at 2 000 steps and 0.09 dB at 5 000 (profiles/r06_run_parity.txt), so agreement between drivers is asserted where it
    # means something -- tests/test_run_parity.py, 5 000 steps -- and here only that no driver falls out of the others' reach.
    assert max(psnr.values()) - min(psnr.values()) <= 1.2, psnr
