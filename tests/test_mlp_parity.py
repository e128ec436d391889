"""Fused MFMA field kernels (include/n2m_mlp.h) against the unfused autocast graph (nn.Linear + F.relu as in
nerf/network.py:10-54,92-108,159-189) on the same weights and samples.

Both sides use fp16 operands with fp32 accumulation and round every layer output to fp16; they differ only in the
accumulation order inside a dot product, so values agree up to occasional 1-ulp fp16 flips that propagate through the
following layers.  Tolerances below are stated in those terms."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_nets(seed=0, emb_scale=0.5):
    import torch
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    torch.manual_seed(seed)
    ref = NeRFNetwork(make_options(O=True, bound=1, dt_gamma=0)).cuda()
    with torch.no_grad():
        ref.encoder.embeddings.uniform_(-emb_scale, emb_scale)
        ref.encoder_color.embeddings.uniform_(-emb_scale, emb_scale)
    fused = NeRFNetwork(make_options(O=True, bound=1, dt_gamma=0, fused_mlp=True)).cuda()
    fused.load_state_dict(ref.state_dict())
    return ref, fused


def samples(M, seed=1):
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.rand(M, 3, device="cuda", generator=g) * 1.9 - 0.95
    d = torch.nn.functional.normalize(torch.randn(M, 3, device="cuda", generator=g), dim=-1)
    return x, d


@pytest.mark.parametrize("shading", ["diffuse", "full", "specular"])
@pytest.mark.parametrize("M", [1, 31, 4096 + 17])
def test_fused_forward(shading, M):
    import torch
    ref, fused = make_nets()
    x, d = samples(M)
    with torch.no_grad():
        with torch.autocast("cuda", dtype=torch.float16):
            s0, c0, p0 = ref(x, d, None, shading)
        s1, c1, p1 = fused(x, d, None, shading)
    # rgb/specular are fp16-quantised sigmoids in [0,1]: a handful of fp16 ulps (2^-11 each)
    assert (c0.float() - c1).abs().max().item() < 6e-3
    assert (c0.float() - c1).abs().mean().item() < 3e-4
    if shading != "diffuse":
        assert (p0.float() - p1).abs().max().item() < 6e-3
    else:
        assert p1 is None
    # sigma = exp(fp16 pre-activation): one fp16 ulp of the exponent is up to 0.8 % of sigma
    rel = ((s0.float() - s1).abs() / s0.float().abs().clamp(min=1e-3))
    assert rel.max().item() < 3e-2 and rel.mean().item() < 2e-3


def test_in_kernel_dir_normalisation_is_bit_identical():
    """raw_dirs=True (safe_normalize on load, nerf/renderer.py:704) == normalising in torch first: forward and gradients."""
    import torch
    from nerf2mesh_amd.renderer import safe_normalize
    _, fused = make_nets()
    x, d = samples(20011)
    raw = d * (torch.rand(d.shape[0], 1, device="cuda") * 3 + 0.2)
    raw[5] = 0.0                                                       # the eps clamp of safe_normalize
    outs = []
    for mode in (False, True):
        fused.zero_grad(set_to_none=True)
        s, c, p = fused(x, raw if mode else safe_normalize(raw), None, "full", raw_dirs=mode)
        (s.sum() + (c * c).sum() + p.sum()).backward()
        outs.append((s.detach(), c.detach(), p.detach(), [q.grad.clone() for q in fused.parameters() if q.grad is not None]))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)
    assert len(outs[0][3]) == len(outs[1][3]) > 0
    for a, b in zip(outs[0][3], outs[1][3]):
        # the dW reduction uses atomics at its final flush: order noise only
        np.testing.assert_allclose(a.float().cpu().numpy(), b.float().cpu().numpy(), rtol=1e-3, atol=1e-5 * float(a.abs().max()) + 1e-12)


def test_fused_density_only():
    import torch
    ref, fused = make_nets()
    x, _ = samples(50000)
    with torch.no_grad():
        with torch.autocast("cuda", dtype=torch.float16):
            s0 = ref.density(x)["sigma"]
        s1 = fused.density(x)["sigma"]
    rel = ((s0.float() - s1).abs() / s0.float().abs().clamp(min=1e-3))
    assert rel.max().item() < 3e-2 and rel.mean().item() < 2e-3


@pytest.mark.parametrize("shading", ["diffuse", "full"])
def test_fused_backward(shading):
    """Gradients of a scalar loss w.r.t. every MLP weight and both hash tables."""
    import torch
    ref, fused = make_nets()
    M = 8192 + 5
    x, d = samples(M, seed=3)
    g = torch.Generator(device="cuda").manual_seed(5)
    cs, cc, cp = torch.randn(M, device="cuda", generator=g), torch.randn(M, 3, device="cuda", generator=g), torch.randn(M, 3, device="cuda", generator=g)
    scale = 128.0                                               # a GradScaler-like loss scale keeps fp16 grads in range

    def loss_of(net, use_autocast):
        if use_autocast:
            with torch.autocast("cuda", dtype=torch.float16):
                s, c, p = net(x, d, None, shading)
        else:
            s, c, p = net(x, d, None, shading)
        l = (torch.log1p(s.float()) * cs).sum() + (c.float() * cc).sum()
        if p is not None:
            l = l + 0.3 * (p.float() * cp).sum()
        return l * scale / M

    loss_of(ref, True).backward()
    loss_of(fused, False).backward()
    names = ["sigma_net.net.0.weight", "sigma_net.net.1.weight", "color_net.net.0.weight", "color_net.net.1.weight", "color_net.net.2.weight"]
    if shading != "diffuse":
        names += ["specular_net.net.0.weight", "specular_net.net.1.weight"]
    pr, pf = dict(ref.named_parameters()), dict(fused.named_parameters())
    for nme in names:
        a, b = pr[nme].grad.float(), pf[nme].grad.float()
        denom = a.abs().max().item() + 1e-12
        err = (a - b).abs().max().item() / denom
        # reference dW is an fp16-rounded GEMM output, ours an fp32 sum of the same fp16 products; activation-gradient
        # ulp flips add a little on top
        assert err < 2e-2, f"{nme}: rel err {err}"
    for nme in ("encoder.embeddings", "encoder_color.embeddings"):
        a, b = pr[nme].grad.float(), pf[nme].grad.float()
        assert torch.isfinite(b).all()
        denom = a.abs().max().item() + 1e-12
        # tables: compare in aggregate (per-level sums) and pointwise relative to the largest entry
        assert (a - b).abs().max().item() / denom < 5e-2, nme
        assert abs(a.sum().item() - b.sum().item()) <= 2e-2 * a.abs().sum().item() + 1e-6, nme
    if shading == "diffuse":
        assert pf["specular_net.net.0.weight"].grad is None


def test_fused_training_matches_unfused_loss_curve():
    """Same seed, same rays: 60 optimisation steps with and without fusion end at the same loss (within 5 %)."""
    import torch
    from nerf2mesh_amd import synthetic
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    from nerf2mesh_amd.trainer import Stage0Trainer
    losses = []
    for fused in (False, True):
        torch.manual_seed(0)
        opt = make_options(O=True, bound=1, dt_gamma=0, fused_mlp=fused)
        tr = Stage0Trainer(NeRFNetwork(opt), opt, synthetic.make_cameras(20, seed=0), torch.device("cuda"), seed=0)
        tr.mark_untrained()
        acc = []
        for i in range(60):
            acc.append(float(tr.train_step()))
        losses.append(np.mean(acc[-10:]))
    assert abs(losses[0] - losses[1]) / losses[0] < 0.10, losses


@pytest.mark.parametrize("shading", ["diffuse", "full"])
def test_fused_field_is_no_farther_from_fp32_truth_than_the_reference_fp16_graph(shading):
    """Ground truth = the same network in fp32 (autocast off: fp32 tables, fp32 nn.Linear).  The reference's `-O` graph (fp16 autocast,
    nerf/renderer.py:721-722) is one fp16 approximation of it, the fused MFMA kernels another (fp16 operands, fp32 accumulate, fp32
    weight-gradient sums).  Per output and per gradient tensor the fused error must not exceed 1.5 x the autocast graph's error (+ a
    floor of a few fp16 ulps of the tensor's scale): a biased gradient or a wrong rounding point would show up here, where a
    comparison of the two fp16 versions with wide bars does not see it."""
    import torch
    ref, fused = make_nets()
    M = 16384 + 3
    x, d = samples(M, seed=11)
    g = torch.Generator(device="cuda").manual_seed(12)
    cs, cc, cp = torch.randn(M, device="cuda", generator=g), torch.randn(M, 3, device="cuda", generator=g), torch.randn(M, 3, device="cuda", generator=g)
    scale = 128.0

    def run(net, mode):
        for p in net.parameters():
            p.grad = None
        if mode == "autocast":
            with torch.autocast("cuda", dtype=torch.float16):
                s, c, p = net(x, d, None, shading)
        else:
            s, c, p = net(x, d, None, shading)            # fp32 graph (ref net), or the fused kernels (fused net)
        l = (torch.log1p(s.float()) * cs).sum() + (c.float() * cc).sum()
        if p is not None:
            l = l + 0.3 * (p.float() * cp).sum()
        (l * scale / M).backward()
        outs = {"sigma": s.detach().float(), "rgb": c.detach().float()}
        if p is not None:
            outs["specular"] = p.detach().float()
        grads = {n: q.grad.detach().float().clone() for n, q in net.named_parameters() if q.grad is not None}
        return outs, grads

    truth_o, truth_g = run(ref, "fp32")
    auto_o, auto_g = run(ref, "autocast")
    fuse_o, fuse_g = run(fused, "fused")
    rows = []
    for name in truth_o:
        t = truth_o[name]
        if name == "sigma":                      # exp(): compare relative errors
            ea = ((auto_o[name] - t).abs() / t.abs().clamp(min=1e-3)).mean().item()
            ef = ((fuse_o[name] - t).abs() / t.abs().clamp(min=1e-3)).mean().item()
            floor = 2.0 ** -11
        else:
            ea, ef = (auto_o[name] - t).abs().mean().item(), (fuse_o[name] - t).abs().mean().item()
            floor = 2.0 ** -12
        rows.append((name, ea, ef, floor))
    for name, t in truth_g.items():
        assert name in fuse_g and name in auto_g, name
        den = t.abs().max().item() + 1e-30
        if "embeddings" in name:                 # tables: mean error over the touched entries, relative to the largest entry
            m = t != 0
            ea = ((auto_g[name] - t).abs()[m].mean() / den).item()
            ef = ((fuse_g[name] - t).abs()[m].mean() / den).item()
        else:
            ea, ef = ((auto_g[name] - t).abs().max() / den).item(), ((fuse_g[name] - t).abs().max() / den).item()
        rows.append(("d " + name, ea, ef, 2.0 ** -11))
    print()
    bad = []
    for name, ea, ef, floor in rows:
        print(f"  {name:34s} autocast-vs-fp32 {ea:.3e}   fused-vs-fp32 {ef:.3e}   ratio {ef / max(ea, 1e-30):.2f}")
        if not ef <= 1.5 * ea + floor:
            bad.append(name)
    assert not bad, f"fused field farther from the fp32 truth than 1.5 x the reference's fp16 graph: {bad}"


@pytest.mark.parametrize("M", [77, 262144 + 13])
def test_train_entry_points_fold_the_specular_regulariser(M):
    """n2m_field_forward_train / n2m_field_backward_train (include/n2m_mlp.h) against the plain entry points + the torch statement of
    nerf/utils.py:733-737: loss += lambda * (specular ** 2).sum(-1).mean(), whose gradient 2 lambda / M * specular (times the seed
    gradient) autograd hands to the specular output.  Forward: sigma / rgb bit-identical, sum of the partials == the torch sum within
    fp32 summation order, unused slots zero.  Backward: d_h1 / d_h2 and the seven dW BIT-identical to the plain backward that is given
    the materialised d_specular = specular * (seed * 2 lambda / M) -- the kernel forms the same product from its recomputed activation."""
    import torch
    from nerf2mesh_amd import _lib as L
    _, net = make_nets()
    p = L.ptr
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cuda").manual_seed(11)
    x, d = samples(M, seed=9)
    h1 = (torch.rand(16, M, device=dev, generator=g) - 0.5).contiguous()
    h2 = (torch.rand(16, M, 2, device=dev, generator=g) - 0.5).half().contiguous()
    w = [q.detach().contiguous() for m in (net.sigma_net, net.color_net, net.specular_net) for q in m.parameters()]
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    n_slots = int(L.lib().n2m_field_spec_partials())
    s = L.stream()
    sig0, rgb0, spec0 = f(M), f(M, 3), f(M, 3)
    L.call("n2m_field_forward", p(x), p(d), p(h1), p(h2), *[p(t) for t in w], M, 1, 1, p(sig0), p(rgb0), p(spec0), s)
    sig1, rgb1 = f(M), f(M, 3)
    part = torch.full((n_slots,), 7.0, device=dev)
    L.call("n2m_field_forward_train", p(x), p(d), p(h1), p(h2), *[p(t) for t in w], M, 1, 1, p(sig1), p(rgb1), None, p(part), s)
    assert torch.equal(sig0, sig1) and torch.equal(rgb0, rgb1)
    want = (spec0.double() ** 2).sum().item()
    assert abs(part.double().sum().item() - want) <= 2e-6 * want
    grid = min(256, max(1, ((M + 31) // 32 + 3) // 4)) * 2
    assert float(part[grid:].abs().max()) == 0.0 if grid < n_slots else True
    # backward
    lam, seed = 1e-5, torch.tensor(4096.0, device=dev)
    d_sigma, d_rgb = torch.randn(M, device=dev, generator=g), torch.randn(M, 3, device=dev, generator=g)
    d_spec = spec0 * (seed * (2.0 * lam / M))
    outs = []
    for fused in (False, True):
        dh1, dh2 = f(16, M), torch.empty(16, M, 2, dtype=torch.float16, device=dev)
        dws = [torch.zeros_like(t) for t in w]
        finf = torch.zeros((), device=dev)
        if fused:
            L.call("n2m_field_backward_train", p(x), p(d), p(h1), p(h2), *[p(t) for t in w], M, 1, 1, p(d_sigma), p(d_rgb), None, p(dh1), p(dh2),
                   *[p(t) for t in dws], p(finf), float(2.0 * lam / M), p(seed), s)
        else:
            L.call("n2m_field_backward", p(x), p(d), p(h1), p(h2), *[p(t) for t in w], M, 1, 1, p(d_sigma), p(d_rgb), p(d_spec), p(dh1), p(dh2),
                   *[p(t) for t in dws], p(finf), s)
        outs.append((dh1, dh2, dws))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for a, b in zip(outs[0][2], outs[1][2]):
        assert torch.equal(a, b)
    assert outs[1][2][5].abs().sum() > 0
    # and the regulariser does reach the gradients: without it the specular head's dW differs
    dws = [torch.zeros_like(t) for t in w]
    L.call("n2m_field_backward", p(x), p(d), p(h1), p(h2), *[p(t) for t in w], M, 1, 1, p(d_sigma), p(d_rgb), None, f(16, M).data_ptr(),
           torch.empty(16, M, 2, dtype=torch.float16, device=dev).data_ptr(), *[p(t) for t in dws], None, s)
    torch.cuda.synchronize()
    assert not torch.equal(dws[6], outs[1][2][6])
