"""Exponential moving average of the parameters, as the reference's Trainer keeps it.

The reference imports `torch_ema.ExponentialMovingAverage` (torch-ema, un-vendored: requirements.txt:11, nerf/utils.py:29) and
  * builds it over `model.parameters()` with `ema_decay = 0.95` for stage 0 (nerf/utils.py:544-545, main.py:241),
  * calls `update()` ONCE PER EPOCH, i.e. every `len(train_loader)` steps = number of training views (nerf/utils.py:1213-1214),
  * evaluates and saves the "best" checkpoint with the averaged weights: `store(); copy_to()` ... `restore()`
    (nerf/utils.py:1103-1112, 1250-1252, 1340-1341, 1389-1401) and keeps `state_dict()` in checkpoints (:1364-1365, 1435-1437).
"Final PSNR" of a reference run is therefore the PSNR of the averaged weights; this class restates the library's published semantics
(version 0.3: `use_num_updates=True` warms the decay up as min(decay, (1 + n) / (10 + n))) over the same method names, with the update
itself one launch of `n2m_ema_update` (csrc/step_helpers.hip) for all tensors -- bit-identical to the library's three torch ops per tensor
(tests/test_ema.py).  Device tensors only: like the rest of the package there is no CPU path.
"""
import contextlib
import ctypes

import torch

from . import _lib as L


def decay_at(decay, num_updates):
    """Decay used by the `num_updates`-th call of update() (1-based), torch_ema's warm-up: min(decay, (1 + n) / (10 + n))."""
    return min(float(decay), (1.0 + num_updates) / (10.0 + num_updates))


class ExponentialMovingAverage:
    def __init__(self, parameters, decay, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay = float(decay)
        self.num_updates = 0 if use_num_updates else None
        self._params = [p for p in parameters]
        # (the library clones every parameter and later zips the shadows with the requires_grad ones: same thing when all of them train,
        # which is the reference's case -- nerf/network.py has no frozen parameter)
        self._params = [p for p in self._params if p.requires_grad]
        self.shadow_params = [p.detach().clone() for p in self._params]
        self.collected_params = None
        self._desc = None

    # ------------------------------------------------------------------------------------------------ the update
    def _descs(self, params):
        key = tuple((p.data_ptr(), s.data_ptr(), p.numel()) for p, s in zip(params, self.shadow_params))
        if self._desc is None or self._desc[0] != key:
            descs = []
            for i in range(0, len(params), L.EMA_MAX):
                d = L.EmaDesc()
                chunk = list(zip(params, self.shadow_params))[i:i + L.EMA_MAX]
                for k, (p, s) in enumerate(chunk):
                    d.shadow[k], d.param[k], d.numel[k] = s.data_ptr(), p.data_ptr(), p.numel()
                d.count = len(chunk)
                descs.append(d)
            self._desc = (key, descs)
        return self._desc[1]

    def _check(self, params):
        if len(params) != len(self.shadow_params):
            raise ValueError("Number of parameters passed as argument is different from number of shadow parameters maintained by this "
                             "ExponentialMovingAverage")
        for p, s in zip(params, self.shadow_params):
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and s.device == p.device):
                raise RuntimeError("ExponentialMovingAverage: contiguous fp32 device tensors only (the update is a HIP kernel; no CPU / PyTorch fallback)")

    @torch.no_grad()
    def update(self, parameters=None):
        params = self._params if parameters is None else [p for p in parameters if p.requires_grad]
        self._check(params)
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = decay_at(decay, self.num_updates)
        for d in self._descs(params):
            L.call("n2m_ema_update", ctypes.addressof(d), float(1.0 - decay), L.stream())

    # --------------------------------------------------------------------------------- averaged weights in / out
    @torch.no_grad()
    def copy_to(self, parameters=None):
        params = self._params if parameters is None else [p for p in parameters]
        for s, p in zip(self.shadow_params, params):
            p.data.copy_(s.data)

    @torch.no_grad()
    def store(self, parameters=None):
        params = self._params if parameters is None else [p for p in parameters]
        self.collected_params = [p.detach().clone() for p in params]

    @torch.no_grad()
    def restore(self, parameters=None):
        if self.collected_params is None:
            raise RuntimeError("This ExponentialMovingAverage has no `store()`ed weights to `restore()`")
        params = self._params if parameters is None else [p for p in parameters]
        for c, p in zip(self.collected_params, params):
            p.data.copy_(c.data)
        self.collected_params = None

    @contextlib.contextmanager
    def average_parameters(self, parameters=None):
        self.store(parameters)
        self.copy_to(parameters)
        try:
            yield
        finally:
            self.restore(parameters)

    def to(self, device=None, dtype=None):
        self.shadow_params = [s.to(device=device, dtype=dtype) if s.is_floating_point() else s.to(device=device) for s in self.shadow_params]
        self._desc = None
        return self

    # ---------------------------------------------------------------------------------------------- checkpoints
    def state_dict(self):
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": self.shadow_params,
                "collected_params": self.collected_params}

    def load_state_dict(self, state_dict):
        self.decay = float(state_dict["decay"])
        if self.decay < 0.0 or self.decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.num_updates = state_dict["num_updates"]
        assert self.num_updates is None or isinstance(self.num_updates, int), "Invalid num_updates"
        shadows = state_dict["shadow_params"]
        assert isinstance(shadows, list) and len(shadows) == len(self.shadow_params), "shadow_params mismatch"
        for s, new in zip(self.shadow_params, shadows):
            s.copy_(new.to(s.device, s.dtype))
        col = state_dict.get("collected_params")
        self.collected_params = None if col is None else [c.to(p.device, p.dtype).clone() for c, p in zip(col, self._params)]
