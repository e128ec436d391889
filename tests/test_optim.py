"""FusedAdamAMP (n2m_adam_step + n2m_scaler_update) against torch.optim.Adam + torch.amp.GradScaler on the same gradients."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fused_adam_amp_matches_torch_adam_and_gradscaler():
    import torch
    from nerf2mesh_amd.optim import FusedAdamAMP
    torch.manual_seed(0)
    shapes = [(1000, 2), (37,), (64, 35), (5, 1)]                       # an odd tail, a tiny tensor
    ref = [torch.randn(s, device="cuda").requires_grad_() for s in shapes]
    mine = [p.detach().clone().requires_grad_() for p in ref]
    lrs = [1e-2, 1e-2, 1e-3, 1e-2]
    opt_ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref, lrs)], eps=1e-15)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=3)
    opt = FusedAdamAMP([{"params": [p], "lr": lr} for p, lr in zip(mine, lrs)], eps=1e-15, init_scale=1024.0, growth_interval=3)
    shadow = mine[0].detach().half()
    opt.shadows[mine[0]] = lambda: shadow
    half_grad = {}
    opt.half_grads[mine[0]] = lambda: half_grad.get("g")
    for step in range(9):
        gs = [torch.randn(s, device="cuda") * (10.0 ** (step % 3 - 2)) for s in shapes]      # stays inside fp16 after scaling
        gs[0] = gs[0].half().float()                                    # representable in fp16: both sides see the same numbers
        if step == 4:
            gs[2][3, 3] = float("inf")                                  # GradScaler must skip this step and back off
        scaler.scale(torch.zeros((), device="cuda"))                    # lazily creates the scale tensor / marks the iteration
        s_ref, s_mine = scaler.get_scale(), float(opt.scale)
        assert s_ref == s_mine
        for p, g in zip(ref, gs):
            p.grad = g * s_ref
        scaler.step(opt_ref)
        scaler.update()
        for i, (p, g) in enumerate(zip(mine, gs)):
            if i == 0:
                p.grad = None
                half_grad["g"] = (g * s_mine).half()
            else:
                p.grad = g * s_mine
        opt.step(flagged=[mine[0]] if step != 4 else [])               # unflagged tensors go through the stock inf check
        for a, b in zip(ref, mine):
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)
        assert torch.equal(shadow, mine[0].detach().half())
    assert float(opt.step_count) == 8 and float(opt.found_inf) == 0
    assert scaler.get_scale() == float(opt.scale)
    for a, b in zip(ref, mine):
        st_a, st_b = opt_ref.state[a], opt.state[b]
        np.testing.assert_allclose(st_b["exp_avg"].cpu().numpy(), st_a["exp_avg"].cpu().numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(st_b["exp_avg_sq"].cpu().numpy(), st_a["exp_avg_sq"].cpu().numpy(), rtol=1e-5, atol=1e-9)


def test_adam_refreshes_packed_table_columns():
    """Shadow modes 2 / 3 of n2m_adam_step: the updated fp32 [rows,1] table lands in column 0 and the fp16 image of the [rows,2] table
    in column 1 of a packed [rows] x 8-byte table (what n2m_grid_encode_forward_packed gathers from)."""
    import torch
    from nerf2mesh_amd.optim import FusedAdamAMP
    torch.manual_seed(1)
    rows = 1027
    a = torch.randn(rows, 1, device="cuda").requires_grad_()
    b = torch.randn(rows, 2, device="cuda").requires_grad_()
    pk = torch.zeros(rows, 2, device="cuda")
    ra, rb = a.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
    ref = torch.optim.Adam([{"params": [ra], "lr": 1e-2}, {"params": [rb], "lr": 3e-3}], eps=1e-15)
    opt = FusedAdamAMP([{"params": [a], "lr": 1e-2}, {"params": [b], "lr": 3e-3}], lr=1e-2, eps=1e-15, amp=False)
    opt.shadows[a] = lambda: (pk, 2)
    opt.shadows[b] = lambda: (pk, 3)
    for _ in range(3):
        a.grad, b.grad = torch.randn_like(a), torch.randn_like(b)
        ra.grad, rb.grad = a.grad.clone(), b.grad.clone()
        opt.step(); ref.step()
    # the two tensors are updated by the same threads (complete packed rows per store): both must still follow torch's Adam
    assert torch.allclose(a, ra, rtol=1e-5, atol=1e-7) and torch.allclose(b, rb, rtol=1e-5, atol=1e-7)
    assert torch.equal(pk[:, 0], a.detach()[:, 0])
    assert torch.equal(pk.view(torch.float16)[:, 2:], b.detach().half())


def test_adam_step_peer_is_reduce_slices_then_adam_then_copy():
    """n2m_adam_step_peer (include/n2m_peer.h, round 5): the slot sum inside the gradient load and the packed-row store to the other ranks'
    tables inside the update pass -- against n2m_peer_reduce_slices -> n2m_adam_step -> n2m_peer_copy on the same inputs: parameter, both
    moments, the local packed rows and the "remote" packed tables bit for bit, for the fp32 [rows,1] and the fp16-gradient [rows,2] table of a
    sharded slice (slots and remote tables are plain local buffers here; the mapped-memory side is tests/test_parallel_gpu.py's)."""
    import ctypes
    import torch
    from nerf2mesh_amd import _lib as L
    torch.manual_seed(11)
    dev = torch.device("cuda")
    W, rows = 3, 4096 + 8                                       # three "ranks", a slice of rows (even: whole 16-byte packed pairs)

    def state():
        g = torch.Generator(device=dev).manual_seed(5)
        p1 = torch.randn(rows, 1, device=dev, generator=g); p2 = torch.randn(rows, 2, device=dev, generator=g)
        return [p1, torch.zeros_like(p1), torch.zeros_like(p1), p2, torch.zeros_like(p2), torch.zeros_like(p2), torch.zeros(rows, 2, device=dev)]
    g = torch.Generator(device=dev).manual_seed(6)
    slots1 = torch.randn(W, rows, device=dev, generator=g) * 300.0                     # scaled gradients, as the flush stores them
    slots2 = (torch.randn(W, rows, 2, device=dev, generator=g) * 300.0).half()
    scale = torch.tensor(512.0, device=dev)
    found = torch.zeros((), device=dev)
    bias = torch.tensor([[0.1, (1 - 0.999) ** 0.5]] * 17, device=dev)

    def desc_for(st, g1, g2):
        d = L.AdamDesc()
        for k, (p, m, v, gr, half, mode) in enumerate(((st[0], st[1], st[2], g1, 0, 2), (st[3], st[4], st[5], g2, 1, 3))):
            d.param[k], d.grad[k], d.exp_avg[k], d.exp_avg_sq[k] = p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr()
            d.half_shadow[k], d.shadow_mode[k] = st[6].data_ptr(), mode
            d.numel[k], d.lr[k], d.grad_is_half[k], d.clear_grad[k], d.slot[k] = p.numel(), 1e-2, half, 0, 0
        d.count = 2
        return d
    for step in range(2):
        # ---- separate passes
        if step == 0:
            a = state(); b = state()
            remote_a = [torch.zeros(rows, 2, device=dev) for _ in range(W - 1)]
            remote_b = [torch.zeros(rows, 2, device=dev) for _ in range(W - 1)]
        g1 = torch.empty(rows, 1, device=dev); g2 = torch.empty(rows, 2, device=dev, dtype=torch.float16)
        L.call("n2m_peer_reduce_slices", L.ptr(slots1), L.ptr(slots2), W, rows, L.ptr(g1), L.ptr(g2), None, L.stream())
        da = desc_for(a, g1, g2)
        L.call("n2m_adam_step", ctypes.addressof(da), 0.9, 0.999, 1e-15, L.ptr(scale), L.ptr(found), L.ptr(bias), L.stream())
        ptrs = L.PeerPtrs()
        ptrs.count = W - 1
        for r in range(W - 1):
            ptrs.ptr[r] = remote_a[r].data_ptr()
        L.call("n2m_peer_copy", a[6].data_ptr(), ctypes.byref(ptrs), rows * 8, L.stream())
        # ---- fused
        db = desc_for(b, slots1, slots2)                         # (grad pointers of slot-fed entries are not read)
        ap = L.AdamPeer()
        ap.world = W
        for sl in range(W):
            ap.slots[0][sl] = slots1[sl].data_ptr()
            ap.slots[1][sl] = slots2[sl].data_ptr()
        ap.packed_local = b[6].data_ptr()
        for r in range(W - 1):
            ap.packed_remote[r] = remote_b[r].data_ptr()
        ap.n_remote = W - 1
        L.call("n2m_adam_step_peer", ctypes.addressof(db), 0.9, 0.999, 1e-15, L.ptr(scale), L.ptr(found), L.ptr(bias), ctypes.addressof(ap), L.stream())
        torch.cuda.synchronize()
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        for x, y in zip(remote_a, remote_b):
            assert torch.equal(x, y) and torch.equal(x, a[6])
        assert torch.equal(a[6][:, 0], a[0][:, 0]) and a[1].abs().max() > 0
        slots1 = slots1 * 0.5 + 1.0; slots2 = (slots2.float() * 0.5 - 1.0).half()      # a second step from the updated state
    # a slice the fused form does not cover is refused, not mangled
    bad = desc_for(b, slots1, slots2)
    bad.numel[0] = rows - 1
    with pytest.raises(RuntimeError):
        L.call("n2m_adam_step_peer", ctypes.addressof(bad), 0.9, 0.999, 1e-15, L.ptr(scale), L.ptr(found), L.ptr(bias), ctypes.addressof(ap), L.stream())


def test_adam_refreshes_packed_rows_at_an_odd_row_offset():
    """The same with every tensor starting at an ODD row of its buffer -- what a rank of the sharded optimizer gets at 8 GPUs (its slice
    of the coarse levels starts at row r * 240 695): the packed rows are then 8- but not 16-byte aligned, the kernel falls back to separate
    column stores, and parameters / moments sit at 4-byte-aligned addresses."""
    import torch
    from nerf2mesh_amd.optim import FusedAdamAMP
    torch.manual_seed(2)
    rows = 1027
    A, B, PK = torch.randn(rows + 1, 1, device="cuda"), torch.randn(rows + 1, 2, device="cuda"), torch.zeros(rows + 1, 2, device="cuda")
    a, b, pk = A[1:].requires_grad_(), B[1:].requires_grad_(), PK[1:]                 # views at row 1
    assert pk.data_ptr() % 16 == 8 and a.data_ptr() % 16 == 4
    ra, rb = a.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
    ref = torch.optim.Adam([{"params": [ra], "lr": 1e-2}, {"params": [rb], "lr": 3e-3}], eps=1e-15)
    opt = FusedAdamAMP([{"params": [a], "lr": 1e-2}, {"params": [b], "lr": 3e-3}], lr=1e-2, eps=1e-15, amp=False)
    for p in (a, b):                                                                    # moments at odd offsets too
        st = opt.state[p]
        for k in ("exp_avg", "exp_avg_sq"):
            buf = torch.zeros(p.numel() + p.shape[1], device="cuda")
            st[k] = buf[p.shape[1]:].view_as(p)
    opt.state_epoch = getattr(opt, "state_epoch", 0) + 1
    opt.shadows[a] = lambda: (pk, 2)
    opt.shadows[b] = lambda: (pk, 3)
    for _ in range(3):
        a.grad, b.grad = torch.randn_like(a), torch.randn_like(b)
        ra.grad, rb.grad = a.grad.clone(), b.grad.clone()
        opt.step(); ref.step()
    assert torch.allclose(a, ra, rtol=1e-5, atol=1e-7) and torch.allclose(b, rb, rtol=1e-5, atol=1e-7)
    assert torch.equal(pk[:, 0], a.detach()[:, 0])
    assert torch.equal(pk.contiguous().view(torch.float16)[:, 2:], b.detach().half())
    assert float(PK[0].abs().sum()) == 0.0 and torch.equal(A[0], A[0]) and float((B[0] - B[0]).abs().sum()) == 0.0      # row 0 untouched


def test_backward_kernels_raise_found_inf():
    """The producing kernels flag non-finite gradients (binned table backward: value read / sum written; field backward: dW)."""
    import torch
    from nerf2mesh_amd.gridencoder import GridEncoder, binned_backward
    enc = GridEncoder(level_dim=2, desired_resolution=2048).cuda()
    B = 5000
    x = torch.rand(B, 3, device="cuda")
    g = (torch.randn(16, B, 2, device="cuda") * 0.1).half()
    out = torch.zeros(enc.host_offsets[-1], 2, device="cuda", dtype=torch.float16)
    flag = torch.zeros((), device="cuda")
    assert binned_backward(enc, g, x, out, 16, found_inf=flag) and float(flag) == 0
    g2 = g.clone(); g2[7, 123, 1] = float("nan")
    assert binned_backward(enc, g2, x, torch.zeros_like(out), 16, found_inf=flag) and float(flag) == 1
    flag.zero_()
    g3 = torch.full_like(g, 60000.0)                                   # finite inputs, sums overflow fp16 at the flush
    x3 = x[:1].expand(B, 3).contiguous()
    assert binned_backward(enc, g3, x3, torch.zeros_like(out), 16, found_inf=flag) and float(flag) == 1


def test_adam_consumes_and_clears_persistent_gradients():
    """ext_grads: a gradient living in a persistent buffer is read by the Adam kernel and left all-zero for the producer's next
    accumulation -- also on a skipped (found_inf) step -- while the update itself equals torch's."""
    import torch
    from nerf2mesh_amd.optim import FusedAdamAMP
    torch.manual_seed(3)
    a = torch.randn(1001, 3, device="cuda").requires_grad_()
    b = torch.randn(64, 64, device="cuda").requires_grad_()
    ra, rb = a.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
    ref = torch.optim.Adam([ra, rb], lr=1e-2, eps=1e-15)
    opt = FusedAdamAMP([a, b], lr=1e-2, eps=1e-15, amp=True, init_scale=4.0)
    flat = torch.zeros(a.numel() + b.numel(), device="cuda")
    va, vb = flat[:a.numel()].view_as(a), flat[a.numel():].view_as(b)
    opt.ext_grads[a] = lambda: va
    opt.ext_grads[b] = lambda: vb
    for it in range(3):
        ga, gb = torch.randn_like(a), torch.randn_like(b)
        va += ga * 4.0; vb += gb * 4.0                       # the producer adds (scaled gradients) into the all-zero buffer
        ra.grad, rb.grad = ga, gb
        opt.step(); ref.step()
        assert not flat.any()
    assert torch.allclose(a, ra, rtol=1e-5, atol=1e-7) and torch.allclose(b, rb, rtol=1e-5, atol=1e-7)
    before = a.detach().clone()
    va += float("inf")
    opt.step()                                               # non-finite: step skipped, scale backed off, buffer still cleared
    assert not flat.any() and torch.equal(a.detach(), before) and float(opt.scale) == 2.0


def test_adam_counts_steps_per_parameter():
    """A parameter that gets its first gradient late starts its bias corrections at t = 1, like torch.optim.Adam's per-parameter
    state["step"] (nerf2mesh's specular head trains only after opt.diffuse_step)."""
    import torch
    from nerf2mesh_amd.optim import FusedAdamAMP
    torch.manual_seed(5)
    a = torch.randn(513, 2, device="cuda").requires_grad_()
    c = torch.randn(64, 32, device="cuda").requires_grad_()
    ra, rc = a.detach().clone().requires_grad_(), c.detach().clone().requires_grad_()
    ref = torch.optim.Adam([ra, rc], lr=1e-2, eps=1e-15)
    opt = FusedAdamAMP([a, c], lr=1e-2, eps=1e-15, amp=False)
    for it in range(7):
        a.grad = torch.randn_like(a); ra.grad = a.grad.clone()
        if it >= 3:                                           # c joins at the fourth step
            c.grad = torch.randn_like(c); rc.grad = c.grad.clone()
        opt.step(); ref.step()
    assert torch.allclose(a, ra, rtol=1e-5, atol=1e-7) and torch.allclose(c, rc, rtol=1e-5, atol=1e-7)
    assert float(opt.steps[0]) == 7 and float(opt.steps[1]) == 7 and float(opt.steps[2]) == 4


def test_state_dict_round_trip_resumes_bias_corrections_and_loss_scale():
    """Checkpoint / resume (the reference saves optimizer + scaler state, nerf/utils.py:1336-1350): a FusedAdamAMP restored from
    state_dict() continues exactly like the one that kept running -- moments, per-slot step counts (bias corrections), loss scale,
    growth tracker -- also for a parameter that joined late."""
    import io
    import torch
    from nerf2mesh_amd.optim import FusedAdamAMP
    torch.manual_seed(1)
    shapes = [(512, 2), (64, 35), (32, 6)]

    def make(src=None):
        ps = [(torch.randn(s, device="cuda") if src is None else src[i].detach().clone()).requires_grad_() for i, s in enumerate(shapes)]
        return ps, FusedAdamAMP([{"params": [p], "lr": 1e-2} for p in ps], eps=1e-15, init_scale=256.0, growth_interval=2)

    def step(ps, opt, k, late=True):
        g = torch.Generator(device="cuda").manual_seed(100 + k)
        for i, p in enumerate(ps):
            p.grad = None if (i == 2 and not late) else torch.randn(p.shape, device="cuda", generator=g) * float(opt.scale)
        if k == 2:
            ps[0].grad[0, 0] = float("inf")              # one skipped step: the scale backs off
        opt.step()

    a, oa = make()
    for k in range(5):
        step(a, oa, k, late=k >= 3)                      # the third tensor gets its first gradient at k = 3
    buf = io.BytesIO()
    torch.save(oa.state_dict(), buf)
    buf.seek(0)
    b, ob = make(src=a)
    ob.load_state_dict(torch.load(buf, weights_only=False))
    assert float(ob.scale) == float(oa.scale) and torch.equal(ob.steps, oa.steps) and float(ob.growth_tracker) == float(oa.growth_tracker)
    assert torch.allclose(ob.bias, oa.bias, rtol=1e-6, atol=0)
    for k in range(5, 9):
        step(a, oa, k)
        step(b, ob, k)
    for p, q in zip(a, b):
        assert torch.equal(p, q)
    assert float(ob.scale) == float(oa.scale) and torch.equal(ob.steps, oa.steps)


@pytest.mark.parametrize("overflow", [False, True])
def test_adam_step_with_the_scaler_tail_is_the_two_launches(overflow):
    """n2m_adam_step_scaler (round 6): the optimizer pass whose last workgroup also does the scaler / step-count / loss-value bookkeeping ==
    n2m_adam_step followed by n2m_scaler_update_slots_loss3, bit for bit, over several steps -- parameters, moments, packed rows, loss scale, growth
    tracker, per-slot step counts and bias corrections, the cleared found_inf flag, the loss value and its running sum; the ticket ends at zero.  A
    GradScaler overflow in the middle (overflow=True) skips the step on both sides and backs the scale off."""
    import ctypes
    import torch
    from nerf2mesh_amd import _lib as L
    p = L.ptr
    torch.manual_seed(3)
    rows = 70001                                   # several workgroups per tensor, a ragged tail
    shapes = [(rows, 1), (rows, 2), (64, 35), (3, 32)]
    n_partial, n_extra, n_extra2, n_rays = 93, 17, 29, 1481

    def fresh():
        g = torch.Generator(device="cuda").manual_seed(5)
        st = {"p": [torch.randn(s, device="cuda", generator=g) for s in shapes]}
        st["m"] = [torch.zeros_like(t) for t in st["p"]]
        st["v"] = [torch.zeros_like(t) for t in st["p"]]
        st["pk"] = torch.zeros(rows, 2, device="cuda")
        st["scale"] = torch.tensor(65536.0, device="cuda")
        st["growth"] = torch.zeros((), device="cuda")
        st["found_inf"] = torch.zeros((), device="cuda")
        st["steps"] = torch.zeros(1 + L.ADAM_MAX, device="cuda")
        st["bias"] = torch.zeros(1 + L.ADAM_MAX, 2, device="cuda")
        st["bias"][:, 0] = 1.0 - 0.9
        st["bias"][:, 1] = float(np.sqrt(1.0 - 0.999))
        st["loss"], st["loss_sum"] = torch.zeros((), device="cuda"), torch.zeros(1, device="cuda")
        st["ticket"] = torch.zeros(64 * 32, dtype=torch.int32, device="cuda")        # N2M_TAIL_TICKET_WORDS
        return st

    def desc_of(st, grads):
        d = L.AdamDesc()
        for k, (t, g) in enumerate(zip(st["p"], grads)):
            d.param[k], d.grad[k], d.exp_avg[k], d.exp_avg_sq[k] = p(t), p(g), p(st["m"][k]), p(st["v"][k])
            d.numel[k], d.lr[k], d.grad_is_half[k], d.clear_grad[k], d.slot[k] = t.numel(), 1e-2 / (k + 1), int(g.dtype == torch.float16), 0, k + 1
            d.half_shadow[k], d.shadow_mode[k] = (p(st["pk"]), 2 + k) if k < 2 else (None, 0)
        d.count = len(grads)
        return d

    a, b = fresh(), fresh()
    gen = torch.Generator(device="cuda").manual_seed(9)
    growth = (2.0, 0.5, 3.0)                       # growth interval 3: the scale grows inside the test
    for step in range(7):
        grads = [torch.randn(shapes[0], device="cuda", generator=gen) * 65536, (torch.randn(shapes[1], device="cuda", generator=gen) * 100).half(),
                 torch.randn(shapes[2], device="cuda", generator=gen) * 65536, torch.randn(shapes[3], device="cuda", generator=gen) * 65536]
        part = torch.rand(n_partial, device="cuda", generator=gen)
        ex, ex2 = torch.rand(n_extra, device="cuda", generator=gen), torch.rand(n_extra2, device="cuda", generator=gen)
        participants = 0b0111 if step < 2 else 0b1111          # the fourth tensor joins late (its own step count starts then)
        n = 3 if step < 2 else 4
        for st in (a, b):
            st["found_inf"].fill_(1.0 if (overflow and step == 3) else 0.0)
        # (a) two launches
        da = desc_of(a, grads[:n])
        L.call("n2m_adam_step", ctypes.addressof(da), 0.9, 0.999, 1e-15, p(a["scale"]), p(a["found_inf"]), p(a["bias"]), L.stream())
        L.call("n2m_scaler_update_slots_loss3", p(a["scale"]), p(a["growth"]), p(a["found_inf"]), p(a["steps"]), p(a["bias"]), participants, 0.9, 0.999,
               *growth, p(part), n_partial, n_rays, p(a["loss"]), p(a["loss_sum"]), p(ex), n_extra, 0.25, p(ex2), n_extra2, 0.125, L.stream())
        # (b) one launch
        db = desc_of(b, grads[:n])
        tail = L.ScalerTail(p(b["growth"]), p(b["steps"]), participants, *growth, p(part), n_partial, n_rays, p(b["loss"]), p(b["loss_sum"]), p(ex), n_extra,
                            0.25, p(ex2), n_extra2, 0.125, p(b["ticket"]))
        L.call("n2m_adam_step_scaler", ctypes.addressof(db), 0.9, 0.999, 1e-15, p(b["scale"]), p(b["found_inf"]), p(b["bias"]), ctypes.addressof(tail), L.stream())
        torch.cuda.synchronize()
        for key in ("scale", "growth", "found_inf", "steps", "bias", "loss", "loss_sum", "pk"):
            assert torch.equal(a[key], b[key]), (step, key, a[key], b[key])
        for key in ("p", "m", "v"):
            for ta, tb in zip(a[key], b[key]):
                assert torch.equal(ta, tb), (step, key)
        assert int(b["ticket"].abs().sum()) == 0 and float(b["found_inf"]) == 0.0
    assert float(a["steps"][1]) == (6 if overflow else 7) and float(a["steps"][4]) == (4 if overflow else 5)
    assert float(a["scale"]) != 65536.0
