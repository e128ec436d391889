"""GPU: stage-1 rendering path end to end (rasterise -> interpolate -> colour networks -> antialias -> ssaa downscale) and one
optimisation step; gradients must reach the colour networks AND the vertex offsets (through antialias only, since the
surface points are detached by default, nerf/renderer.py:878)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused", [False, True])
def test_render_stage1_and_step(fused):
    import torch
    from nerf2mesh_amd import synthetic as S
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    from nerf2mesh_amd.trainer import Stage1Trainer
    torch.manual_seed(0)
    opt = make_options(O=True, bound=1, dt_gamma=0, stage=1, fused_mlp=fused)
    dev = torch.device("cuda")
    v, f = S.scene_mesh(20000)
    tr = Stage1Trainer(NeRFNetwork(opt), opt, S.make_cameras(6, seed=0), v, f, dev, H=200, W=200)
    assert abs(tr.optimizer.param_groups[0]["lr"] - 0.01 * opt.lr) < 1e-12          # main.py:239: the schedule starts at 1 % of lr
    for _ in range(500):
        tr.scheduler.step()               # past the warm-up, so that a dozen steps are enough to see the loss move
    assert abs(tr.optimizer.param_groups[0]["lr"] - opt.lr) < 1e-9
    losses = [float(tr.train_step().detach()) for _ in range(12)]
    m = tr.model
    assert m.vertices_offsets.grad is not None and m.vertices_offsets.grad.abs().sum() > 0, "no gradient reached the vertices"
    assert m.encoder_color.embeddings.grad.abs().sum() > 0 and m.color_net.net[0].weight.grad.abs().sum() > 0
    assert m.encoder.embeddings.grad is None or m.encoder.embeddings.grad.abs().sum() == 0     # density branch is not used in stage 1
    assert m.triangles_errors_cnt.sum() > 0
    assert sum(losses[-4:]) < sum(losses[:4]), losses
    unseen = m.mark_unseen_triangles(m.vertices, m.triangles, tr.mvps[:3], 200, 200)
    assert 0 < int(unseen.sum()) < f.shape[0]


def test_fused_color_matches_unfused():
    import torch
    from test_mlp_parity import make_nets, samples
    ref, fused = make_nets()
    x, d = samples(5000)
    with torch.no_grad():
        with torch.autocast("cuda", dtype=torch.float16):
            c0, p0 = ref.rgb(x, d, None, "full")
        c1, p1 = fused.rgb(x, d, None, "full")
    assert (c0.float() - c1).abs().max().item() < 6e-3 and (p0.float() - p1).abs().max().item() < 6e-3
