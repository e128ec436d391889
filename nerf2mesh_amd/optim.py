"""Adam + dynamic loss scaling for the whole parameter set in two launches (C ABI: n2m_adam_step / n2m_scaler_update).

Same arithmetic as `torch.optim.Adam(..., fused=True)` driven by `torch.amp.GradScaler` (main.py:221, nerf/utils.py:506,
1187-1190): gradients are divided by the loss scale inside the update, the whole step is skipped when a gradient is not finite,
the scale backs off / grows like GradScaler.update().  What differs is where the work happens:
* ONE kernel updates all tensors (torch: 4 multi-tensor launches + an unscale / inf-check pass over every gradient first);
* the non-finite check of the two big table gradients and of the MLP weight gradients is done by the kernels that produce them
  (they set `found_inf`); only gradients that did not come from those kernels are checked with the stock foreach op;
* the colour table's fp16 working copy is refreshed in the same pass, and its gradient is read as fp16.
"""
import ctypes

import torch

from . import _lib as L


class FusedAdamAMP(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, amp=True, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5,
                 growth_interval=2000):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        ps = [p for g in self.param_groups for p in g["params"]]
        if len(ps) > L.ADAM_MAX:
            raise ValueError(f"FusedAdamAMP handles at most {L.ADAM_MAX} tensors per step (got {len(ps)})")
        dev = ps[0].device
        if dev.type != "cuda" or any(p.dtype != torch.float32 or not p.is_contiguous() for p in ps):
            raise ValueError("FusedAdamAMP needs contiguous fp32 CUDA parameters")
        self.amp = bool(amp)
        self.scale = torch.full((), float(init_scale) if amp else 1.0, device=dev)
        self.growth_tracker = torch.zeros((), device=dev)
        self.found_inf = torch.zeros((), device=dev)
        # successful steps on the device (a skipped step does not count): [0] all steps, [1 + i] those parameter i took part in --
        # torch.optim.Adam counts per parameter, so a parameter whose first gradient comes late starts its bias corrections at t = 1
        self.steps = torch.zeros(1 + L.ADAM_MAX, device=dev)
        self.step_count = self.steps[0]
        self.growth = (float(growth_factor), float(backoff_factor), float(growth_interval))
        b1, b2 = betas
        self.bias = torch.tensor([[1.0 - b1, (1.0 - b2) ** 0.5]] * (1 + L.ADAM_MAX), dtype=torch.float32).to(dev)   # corrections of step t = 1, per slot
        self._slot = {p: 1 + i for i, p in enumerate(ps)}
        self._one = torch.ones((), device=dev)
        for p in ps:
            self.state[p] = dict(exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
        self.shadows = {}          # param -> callable returning the working copy to refresh: a tensor (plain fp16 copy), (tensor, mode)
                                   # with mode 2 / 3 = column of a packed table (N2mAdamDesc.shadow_mode), or None
        self.half_grads = {}       # param -> callable returning an fp16 gradient produced outside autograd (or None)
        self.ext_grads = {}        # param -> callable returning an fp32 gradient that lives in a PERSISTENT buffer its producer adds into
                                   # (or None): the Adam kernel zeroes it after reading, so the producer needs no zero-fill launch

    def state_dict(self):
        """torch's optimizer state (exp_avg / exp_avg_sq per parameter, param_groups) + what torch keeps in state[p]["step"] and in a
        separate GradScaler: per-slot step counts, loss scale, growth tracker.  A resumed run continues its bias corrections and its
        loss scale where the saved one stopped (the reference saves optimizer and scaler state side by side, nerf/utils.py:1336-1350)."""
        if getattr(self, "shard_sync", None) is not None:
            # engine.Stage0Engine with the optimizer sharded over the ranks: each rank has advanced the moments of its own 1/W of the table
            # rows only.  Gather them (COLLECTIVE: every rank calls state_dict() at the same step) so that the saved state is complete
            self.shard_sync()
        sd = super().state_dict()
        sd["n2m_amp"] = {"steps": self.steps.detach().cpu().clone(), "scale": float(self.scale), "growth_tracker": float(self.growth_tracker),
                         "amp": self.amp, "growth": self.growth}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        extra = state_dict.pop("n2m_amp", None)
        super().load_state_dict(state_dict)
        if extra is not None:
            self.steps.copy_(extra["steps"].to(self.steps.device))
            self.scale.fill_(extra["scale"])
            self.growth_tracker.fill_(extra["growth_tracker"])
            self.growth = tuple(extra.get("growth", self.growth))
            # bias corrections of the NEXT step of every slot, like n2m_scaler_update_slots leaves them (double arithmetic)
            b1, b2 = self.param_groups[0]["betas"]
            t = self.steps.double().cpu() + 1
            bias = torch.stack([1.0 - torch.tensor(b1, dtype=torch.float64) ** t, (1.0 - torch.tensor(b2, dtype=torch.float64) ** t).sqrt()], dim=1)
            self.bias.copy_(bias.float().to(self.bias.device))
        self.found_inf.zero_()
        self.state_epoch = getattr(self, "state_epoch", 0) + 1      # the moment tensors were replaced: cached N2mAdamDesc pointers are stale

    def scale_loss(self, loss, world=1):
        """loss * scale / world: with gradients SUMMED over `world` ranks the update sees their mean."""
        f = (self.scale / world if world > 1 else self.scale) if self.amp else (1.0 / world if world > 1 else None)
        return loss if f is None else loss * f

    def backward(self, loss, world=1):
        """scale_loss(loss, world).backward() without materialising the product: the factor goes in as the seed gradient (a device
        scalar), which the loss head's backward kernel reads by pointer -- two small launches fewer per step."""
        f = (self.scale / world if world > 1 else self.scale) if self.amp else (self._one / world if world > 1 else None)
        if f is None:
            loss.backward()
        else:
            loss.backward(gradient=f.to(loss.dtype).reshape(loss.shape))

    @torch.no_grad()
    def step(self, flagged=()):
        """flagged: parameters whose gradients were already checked for inf/nan by the kernels that produced them."""
        flagged = {id(p) for p in flagged}
        desc = L.AdamDesc()
        keep, unchecked, k, participants = [], [], 0, 0
        for group in self.param_groups:
            for p in group["params"]:
                hg = self.half_grads.get(p)
                g = hg() if hg is not None else None
                is_half = g is not None
                clear = False
                if g is None:
                    eg = self.ext_grads.get(p)
                    g = eg() if eg is not None else None
                    clear = g is not None
                if g is None:
                    g = p.grad
                if g is None:
                    continue
                if not g.is_contiguous():
                    assert not clear, "a persistent gradient buffer must be contiguous"
                    g = g.contiguous()
                if id(p) not in flagged:
                    unchecked.append(g)
                st = self.state[p]
                sh = self.shadows.get(p)
                sh = sh() if sh is not None else None
                mode = 1
                if isinstance(sh, tuple):
                    sh, mode = sh
                desc.param[k], desc.grad[k] = p.data_ptr(), g.data_ptr()
                desc.exp_avg[k], desc.exp_avg_sq[k] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                desc.half_shadow[k] = sh.data_ptr() if sh is not None else None
                desc.numel[k], desc.lr[k], desc.grad_is_half[k] = p.numel(), float(group["lr"]), int(is_half)
                desc.shadow_mode[k] = int(mode) if sh is not None else 0
                desc.clear_grad[k] = int(clear and g.is_contiguous())
                desc.slot[k] = self._slot[p]
                participants |= 1 << (self._slot[p] - 1)
                keep += [g, sh]
                k += 1
        if k == 0:
            return
        desc.count = k
        if unchecked and self.amp:
            for dt in {g.dtype for g in unchecked}:
                torch._amp_foreach_non_finite_check_and_unscale_([g for g in unchecked if g.dtype == dt], self.found_inf, self._one)
        b1, b2 = self.param_groups[0]["betas"]
        s = L.stream()
        L.call("n2m_adam_step", ctypes.addressof(desc), float(b1), float(b2), float(self.param_groups[0]["eps"]),
               L.ptr(self.scale) if self.amp else None, L.ptr(self.found_inf), L.ptr(self.bias), s)
        gf, bf, gi = self.growth
        L.call("n2m_scaler_update_slots", L.ptr(self.scale) if self.amp else None, L.ptr(self.growth_tracker) if self.amp else None,
               L.ptr(self.found_inf), L.ptr(self.steps), L.ptr(self.bias), participants, float(b1), float(b2), gf, bf, gi, s)
