"""EMA of the parameters (nerf2mesh_amd/ema.py): the reference's Trainer keeps torch_ema.ExponentialMovingAverage(decay 0.95) over
model.parameters(), updates it once per epoch and evaluates with it (nerf/utils.py:544-545,1213-1214,1250-1252; main.py:241)."""
import pytest
import torch


def _torch_ema_update(shadows, params, decay, num_updates):
    """The library's update(), literally (torch-ema 0.3)."""
    num_updates += 1
    d = min(decay, (1 + num_updates) / (10 + num_updates))
    omd = 1.0 - d
    for s, p in zip(shadows, params):
        tmp = s - p
        tmp.mul_(omd)
        s.sub_(tmp)
    return num_updates


def test_decay_warm_up_schedule():
    from nerf2mesh_amd.ema import decay_at
    assert decay_at(0.95, 1) == pytest.approx(2 / 11)
    assert decay_at(0.95, 20) == pytest.approx(21 / 30)
    assert decay_at(0.95, 170) == pytest.approx(171 / 180)
    assert decay_at(0.95, 171) == 0.95                       # (1 + n) / (10 + n) > 0.95 from n = 171 on
    assert decay_at(0.5, 100) == 0.5


def test_cpu_tensors_are_refused():
    """No CPU / PyTorch fallback: the update is a HIP kernel."""
    from nerf2mesh_amd.ema import ExponentialMovingAverage
    p = torch.nn.Parameter(torch.zeros(8))
    ema = ExponentialMovingAverage([p], 0.95)
    with pytest.raises(RuntimeError, match="no CPU"):
        ema.update()


@pytest.mark.gpu
def test_update_is_bit_identical_to_the_library_ops():
    from nerf2mesh_amd.ema import ExponentialMovingAverage
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    shapes = [(6119864, 1), (100003, 2), (32, 19), (1, 32), (7,), (1,), ()]      # aligned, odd, tiny, scalar (the SDF variance)
    params = [torch.nn.Parameter(torch.randn(s, device=dev, generator=g)) for s in shapes]
    frozen = torch.nn.Parameter(torch.randn(5, device=dev), requires_grad=False)
    ema = ExponentialMovingAverage(params + [frozen], 0.95)
    assert len(ema.shadow_params) == len(params)
    ref = [p.detach().clone() for p in params]
    n = 0
    for it in range(12):
        with torch.no_grad():
            for p in params:
                p.add_(torch.randn(p.shape, device=dev, generator=g) * 0.1)
        ema.update()
        n = _torch_ema_update(ref, [p.detach() for p in params], 0.95, n)
        for a, b in zip(ema.shadow_params, ref):
            assert torch.equal(a, b), (it, a.shape)
    assert ema.num_updates == n == 12
    # store / copy_to / restore, state_dict round trip
    before = [p.detach().clone() for p in params]
    with ema.average_parameters():
        for p, s in zip(params, ema.shadow_params):
            assert torch.equal(p.detach(), s)
    for p, b in zip(params, before):
        assert torch.equal(p.detach(), b)
    ema2 = ExponentialMovingAverage(params, 0.5)
    ema2.load_state_dict(ema.state_dict())
    assert ema2.decay == 0.95 and ema2.num_updates == 12
    for a, b in zip(ema2.shadow_params, ema.shadow_params):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_engine_and_trainer_update_once_per_epoch_and_evaluate_the_average():
    from nerf2mesh_amd import synthetic
    from nerf2mesh_amd.engine import Stage0Engine
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    from nerf2mesh_amd.trainer import Stage0Trainer
    dev = torch.device("cuda", 0)
    for cls in (Stage0Engine, Stage0Trainer):
        torch.manual_seed(0)
        opt = make_options(O=True, bound=1, dt_gamma=0, iters=300, fused_mlp=True)
        opt.num_rays, opt.num_points = 1024, 1 << 14
        tr = cls(NeRFNetwork(opt), opt, synthetic.make_cameras(6, seed=0), dev, seed=0)        # epoch = 6 steps
        tr.mark_untrained()
        assert tr.ema is not None and tr.epoch_len == 6
        names = [n for n, p in tr.model.named_parameters() if p.requires_grad]
        ref = [p.detach().clone() for p in tr.model.parameters() if p.requires_grad]
        n = 0
        for it in range(1, 20):
            tr.train_step()
            if it % 6 == 0:
                n = _torch_ema_update(ref, [p.detach() for p in tr.model.parameters() if p.requires_grad], 0.95, n)
            assert tr.ema.num_updates == it // 6
        torch.cuda.synchronize()
        for name, a, b in zip(names, tr.ema.shadow_params, ref):
            assert torch.equal(a, b), (cls.__name__, name)
        raw = [p.detach().clone() for p in tr.model.parameters()]
        p_ema, p_raw = tr.eval_psnr(cam=0, use_ema=True), tr.eval_psnr(cam=0)
        assert p_ema == p_ema and p_raw == p_raw and p_ema != p_raw            # finite, and the averaged weights really were rendered
        for p, b in zip(tr.model.parameters(), raw):                              # ... and taken out again
            assert torch.equal(p.detach(), b)
        tr.train_step()                                                           # the step still runs on the restored weights
