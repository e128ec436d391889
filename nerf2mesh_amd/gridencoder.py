"""Multiresolution hash-grid encoder on libn2m_hip.so -- host-side mirror of the reference's gridencoder/grid.py.

`grid_encode` and `GridEncoder` keep the reference's names, arguments, parameter/buffer names and shapes
(`embeddings [rows, C]`, `offsets [L+1] int32`, so reference checkpoints load unchanged) and its autocast policy
(fp16 tables iff autocast is on and C is even, grid.py:45).  Differences, all internal:

* features are produced directly SAMPLE-major [B, L*C] by the kernel (n2m_grid_encode_forward_bm) and the
  backward consumes the incoming gradient in that layout, so the two permute/contiguous passes of
  grid.py:63,81 (128 B per sample each way) disappear; the level-major ABI entry points remain for the
  reference's own wrapper (nerf2mesh_amd/backends/_gridencoder.py);
* level offsets are kept on the host as well, no device read is needed to size anything.
"""
import math

import os

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L

_p = L.ptr

_gridtype_to_id = {"hash": 0, "tiled": 1}
_interp_to_id = {"linear": 0, "smoothstep": 1}


def _dtype_id(t):
    return L.F16 if t.dtype == torch.float16 else L.F32


class _grid_encode(Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, interpolation=0, max_level=None):
        inputs = inputs.float().contiguous()
        B, D = inputs.shape
        Lv = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = float(np.log2(per_level_scale))
        H = int(base_resolution)
        max_level = Lv if max_level is None else min(int(max_level), Lv)

        # autocast policy of the reference: half tables only when C is even (packed fp16 atomics in backward)
        if torch.is_autocast_enabled() and C % 2 == 0:
            embeddings = embeddings.to(torch.half)
        embeddings = embeddings.contiguous()
        dt = _dtype_id(embeddings)
        s = L.stream()

        if calc_grad_inputs:
            # inputs need a gradient (SDF normals through autograd, stage-1 vertex offsets): keep dy_dx
            outputs = torch.empty(Lv, B, C, device=inputs.device, dtype=embeddings.dtype)
            dy_dx = torch.empty(B, Lv * D * C, device=inputs.device, dtype=embeddings.dtype)
            if max_level < Lv:
                outputs.zero_()
                dy_dx.zero_()
            L.call("n2m_grid_encode_forward", _p(inputs), _p(embeddings), _p(offsets), _p(outputs), B, D, C, Lv, max_level, S, H,
                   _p(dy_dx), gridtype, int(bool(align_corners)), interpolation, dt, s)
            outputs = outputs.permute(1, 0, 2).reshape(B, Lv * C)
        else:
            dy_dx = None
            outputs = torch.empty(B, Lv * C, device=inputs.device, dtype=embeddings.dtype)
            L.call("n2m_grid_encode_forward_bm", _p(inputs), _p(embeddings), _p(offsets), _p(outputs), B, D, C, Lv, max_level, S, H,
                   gridtype, int(bool(align_corners)), interpolation, dt, s)

        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.cfg = (B, D, C, Lv, S, H, gridtype, interpolation, max_level, int(bool(align_corners)))
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, Lv, S, H, gridtype, interpolation, max_level, align = ctx.cfg
        dt = _dtype_id(embeddings)
        s = L.stream()
        grad = grad.to(embeddings.dtype)
        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = None
        if dy_dx is None and _binned_single(grad, inputs, offsets, grad_embeddings, B, D, C, Lv, max_level, S, H, gridtype, align, interpolation, dt):
            pass                                           # exact fixed-point sums instead of order-dependent atomics
        elif dy_dx is None:
            grad = grad.contiguous()                       # [B, L*C], consumed as is
            L.call("n2m_grid_encode_backward_bm", _p(grad), _p(inputs), _p(embeddings), _p(offsets), _p(grad_embeddings), B, D, C,
                   Lv, max_level, S, H, gridtype, align, interpolation, dt, s)
        else:
            grad = grad.view(B, Lv, C).permute(1, 0, 2).contiguous()
            grad_inputs = torch.zeros(B, D, device=inputs.device, dtype=embeddings.dtype)
            L.call("n2m_grid_encode_backward", _p(grad), _p(inputs), _p(embeddings), _p(offsets), _p(grad_embeddings), B, D, C, Lv,
                   max_level, S, H, _p(dy_dx), _p(grad_inputs), gridtype, align, interpolation, dt, s)
            grad_inputs = grad_inputs.to(inputs.dtype)
        return grad_inputs, grad_embeddings, None, None, None, None, None, None, None, None


def host_offsets_of(offsets, Lv):
    """The level offsets of an `offsets` tensor as a HOST int32 array (the binned entry points plan on the host): read back once per tensor
    OBJECT -- the copy rides on the object; an address-keyed cache would hand a new model the offsets of a freed one."""
    cached = getattr(offsets, "_n2m_host_offsets", None)
    if cached is None or cached[0] != (offsets._version, int(Lv)):
        ho = np.ascontiguousarray(offsets.detach().cpu().numpy()[:Lv + 1].astype(np.int32))
        try:
            offsets._n2m_host_offsets = cached = ((offsets._version, int(Lv)), ho)
        except AttributeError:      # (an object that takes no attributes: read back per call)
            return ho
    return cached[1]


_GENERIC_ONLY = os.environ.get("N2M_SHIM_GENERIC", "0") == "1"


def _binned_single(grad_bm, inputs, offsets, grad_embeddings, B, D, C, Lv, max_level, S, H, gridtype, align, interp, dt):
    """One table's backward through the shared-fill kernels (the other table NULL) when the call is covered: D = 3, fp32 C = 1 or fp16
    C = 2, grad sample-major [B, L*C]; adds onto grad_embeddings.  False: the caller runs the generic kernel."""
    if _GENERIC_ONLY or D != 3 or B == 0 or max_level <= 0 or not ((C == 1 and dt == L.F32) or (C == 2 and dt == L.F16)):
        return False
    ho = host_offsets_of(offsets, Lv)
    need = L.lib().n2m_grid_binned_pair_workspace_bytes(B, max_level, ho.ctypes.data)
    if need == 0:
        return False
    g = grad_bm.reshape(B, Lv, C).permute(1, 0, 2).contiguous()
    ws = L.workspace(inputs.device, need, 0)
    L.grid_backward_config(1, 1.0)
    g1, g2, t1, t2 = (g, None, grad_embeddings, None) if C == 1 else (None, g, None, grad_embeddings)
    L.call("n2m_grid_encode_backward_binned_pair", _p(g1), _p(g2), _p(inputs), ho.ctypes.data, _p(t1), _p(t2), B, Lv, max_level, float(S), int(H),
           gridtype, int(align), interp, None, 0.0, 0.0, 1.0, None, None, 1.0, 0.0, 0, _p(ws), ws.numel(), L.stream())
    return True


def _host_offsets(enc):
    """int32 numpy copy of the level offsets for the binned kernels' host-side plan (cached on the encoder)."""
    arr = getattr(enc, "_host_offsets_np", None)
    if arr is None:
        arr = np.ascontiguousarray(np.asarray(enc.host_offsets, dtype=np.int32))
        enc._host_offsets_np = arr
    return arr


_PAIR_FOR_COLOR = os.environ.get("N2M_COLOR_BWD_SINGLE", "0") != "1"      # A/B switch: the round-1 single-table kernels for a lone C=2 fp16 table


def binned_backward(enc, grad_lm, x01, grad_embeddings, max_level, tv=None, found_inf=None, ws_slot=0):
    """grad_embeddings += scatter of grad_lm [L,B,C] through the binned fixed-point kernels (include/n2m_hip.h);
    returns False when the configuration is not covered (caller uses n2m_grid_encode_backward).
    tv = (embeddings fp32, weight, weight_outer, inner01, scale tensor | None) folds the TV gradient over the same inputs in."""
    B, C = x01.shape[0], grad_embeddings.shape[1]
    if x01.shape[1] != 3 or not hasattr(enc, "host_offsets"):
        return False
    dt = _dtype_id(grad_embeddings)
    ho = _host_offsets(enc)
    if C == 2 and dt == L.F16 and tv is None and max_level > 0 and B > 0 and _PAIR_FOR_COLOR:
        # the colour table alone through the shared-fill kernels (grad1 = NULL): same-cell runs of consecutive samples merged, SoA logs,
        # XCD-aware level order, walk-ahead accumulate -- measured 435 -> ~170 us on a stage-1 frame against the round-1 single-table kernels
        need = L.lib().n2m_grid_binned_pair_workspace_bytes(B, max_level, ho.ctypes.data)
        if need != 0:
            ws = L.workspace(x01.device, need, ws_slot)
            L.grid_backward_config(1, 1.0)
            L.call("n2m_grid_encode_backward_binned_pair", None, _p(grad_lm), _p(x01), ho.ctypes.data, None, _p(grad_embeddings), B, enc.num_levels,
                   max_level, float(np.log2(enc.per_level_scale)), int(enc.base_resolution), enc.gridtype_id, int(bool(enc.align_corners)),
                   enc.interp_id, None, 0.0, 0.0, 1.0, None, _p(found_inf), 1.0, 0.0, 0, _p(ws), ws.numel(), L.stream())
            return True
    if C == 1 and dt == L.F32 and max_level > 0 and B > 0 and _PAIR_FOR_COLOR and (tv is None or max_level == enc.num_levels):
        # the density table alone through the same kernels (grad2 = NULL): the 6 M stacked finite-difference samples of the SDF head
        need = L.lib().n2m_grid_binned_pair_workspace_bytes(B, max_level, ho.ctypes.data)
        if need != 0:
            ws = L.workspace(x01.device, need, ws_slot)
            tv_emb, tv_w, tv_wo, tv_in, tv_scale = tv if tv is not None else (None, 0.0, 0.0, 1.0, None)
            L.grid_backward_config(1, 1.0)
            L.call("n2m_grid_encode_backward_binned_pair", _p(grad_lm), None, _p(x01), ho.ctypes.data, _p(grad_embeddings), None, B, enc.num_levels,
                   max_level, float(np.log2(enc.per_level_scale)), int(enc.base_resolution), enc.gridtype_id, int(bool(enc.align_corners)),
                   enc.interp_id, _p(tv_emb), float(tv_w), float(tv_wo), float(tv_in), _p(tv_scale), _p(found_inf), 1.0, 0.0, 0, _p(ws),
                   ws.numel(), L.stream())
            return True
    need = L.lib().n2m_grid_binned_workspace_bytes(B, 3, C, max_level, ho.ctypes.data, dt, 0)
    if need == 0:
        return False
    if tv is not None and not (dt == L.F32 and C == 1 and max_level == enc.num_levels):
        return False
    ws = L.workspace(x01.device, need, ws_slot)
    tv_emb, tv_w, tv_wo, tv_in, tv_scale = tv if tv is not None else (None, 0.0, 0.0, 1.0, None)
    L.grid_backward_config(1, 1.0)          # per-thread setting: stated per call (a sharded engine on this thread may have left its own)
    L.call("n2m_grid_encode_backward_binned", _p(grad_lm), _p(x01), ho.ctypes.data, _p(grad_embeddings), B, 3, C, enc.num_levels, max_level,
           float(np.log2(enc.per_level_scale)), int(enc.base_resolution), enc.gridtype_id, int(bool(enc.align_corners)), enc.interp_id, dt,
           _p(tv_emb), float(tv_w), float(tv_wo), float(tv_in), _p(tv_scale), _p(found_inf), _p(ws), ws.numel(), L.stream())
    return True


def same_geometry(a, b):
    """Two encoders index their tables identically (levels, resolutions, hash/tiled, offsets): their backward can share one fill."""
    return (list(a.host_offsets) == list(b.host_offsets) and a.per_level_scale == b.per_level_scale and a.base_resolution == b.base_resolution
            and a.gridtype_id == b.gridtype_id and bool(a.align_corners) == bool(b.align_corners) and a.interp_id == b.interp_id
            and a.input_dim == b.input_dim == 3)


def binned_backward_pair(enc1, enc2, grad1_lm, grad2_lm, x01, g1, g2, max_level, tv=None, found_inf=None, in_affine=(1.0, 0.0), overwrite=False):
    """Both table gradients (g1 fp32 C=1, g2 fp16 C=2) from one shared fill (n2m_grid_encode_backward_binned_pair); False when not
    applicable.  overwrite=False: added onto g1 / g2 (zero-filled or running sums); True: g1 / g2 may be uninitialised, the call
    defines every row."""
    if not (g1.dtype == torch.float32 and g1.shape[1] == 1 and g2.dtype == torch.float16 and g2.shape[1] == 2):
        return False
    if not (hasattr(enc1, "host_offsets") and hasattr(enc2, "host_offsets") and same_geometry(enc1, enc2)):
        return False
    if tv is not None and max_level != enc1.num_levels:
        return False
    B = x01.shape[0]
    if B == 0:
        if overwrite:
            g1.zero_(); g2.zero_()
        return True
    ho = _host_offsets(enc1)
    need = L.lib().n2m_grid_binned_pair_workspace_bytes(B, max_level, ho.ctypes.data)
    if need == 0:
        return False
    ws = L.workspace(x01.device, need)
    tv_emb, tv_w, tv_wo, tv_in, tv_scale = tv if tv is not None else (None, 0.0, 0.0, 1.0, None)
    L.grid_backward_config(1, 1.0)          # per-thread setting: stated per call (a sharded engine on this thread may have left its own)
    L.call("n2m_grid_encode_backward_binned_pair", _p(grad1_lm), _p(grad2_lm), _p(x01), ho.ctypes.data, _p(g1), _p(g2), B, enc1.num_levels,
           max_level, float(np.log2(enc1.per_level_scale)), int(enc1.base_resolution), enc1.gridtype_id, int(bool(enc1.align_corners)),
           enc1.interp_id, _p(tv_emb), float(tv_w), float(tv_wo), float(tv_in), _p(tv_scale), _p(found_inf), float(in_affine[0]),
           float(in_affine[1]), int(bool(overwrite)), _p(ws), ws.numel(), L.stream())
    return True


def binned_tv(enc, x01, emb, grad, weight, weight_outer=None, inner01=1.0, scale=None):
    B, C = x01.shape[0], emb.shape[1]
    ho = _host_offsets(enc)
    need = L.lib().n2m_grid_binned_workspace_bytes(B, 3, C, enc.num_levels, ho.ctypes.data, L.F32, 1)
    if need == 0:
        return False
    ws = L.workspace(x01.device, need)
    L.grid_backward_config(1, 1.0)          # per-thread setting: stated per call (a sharded engine on this thread may have left its own)
    L.call("n2m_grad_total_variation_binned", _p(x01), _p(emb), _p(grad), ho.ctypes.data, float(weight),
           float(weight if weight_outer is None else weight_outer), float(inner01), _p(scale), B, 3, C, enc.num_levels,
           float(np.log2(enc.per_level_scale)), int(enc.base_resolution), enc.gridtype_id, int(bool(enc.align_corners)), _p(ws), ws.numel(),
           L.stream())
    return True


def grid_encode(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, interpolation=0, max_level=None):
    """inputs [B,D] in [0,1], embeddings [rows,C], offsets [L+1] -> features [B, L*C] (gridencoder/grid.py:24-98)."""
    return _grid_encode.apply(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs, gridtype,
                              align_corners, interpolation, max_level)


def level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners=False):
    """Row offset of every level (gridencoder/grid.py:121-135): (res+1)^D entries capped at 2^log2_hashmap_size,
    rounded up to a multiple of 8."""
    cap = 2 ** log2_hashmap_size
    offs, off = [], 0
    for i in range(num_levels):
        res = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(cap, (res if align_corners else res + 1) ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offs.append(off)
        off += n
    offs.append(off)
    return offs


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype="hash", align_corners=False, interpolation="linear"):
        super().__init__()
        if desired_resolution is not None:   # finest resolution wins over per_level_scale (grid.py:106-108)
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.interpolation = interpolation
        self.interp_id = _interp_to_id[interpolation]
        self.align_corners = align_corners
        self.max_params = 2 ** log2_hashmap_size

        offs = level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners)
        self.host_offsets = offs
        self.register_buffer("offsets", torch.tensor(offs, dtype=torch.int32))
        self.n_params = offs[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(offs[-1], level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)   # grid.py:144-146

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} gridtype={self.gridtype} "
                f"align_corners={self.align_corners} interpolation={self.interpolation}")

    def forward(self, inputs, bound=1, max_level=None):
        """inputs [..., input_dim] in [-bound, bound] -> [..., num_levels*level_dim] (grid.py:151-168)."""
        inputs = (inputs + bound) / (2 * bound)
        prefix = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        out = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, inputs.requires_grad,
                          self.gridtype_id, self.align_corners, self.interp_id, max_level)
        return out.view(prefix + [self.output_dim])

    def half_table(self):
        """fp16 view of the table as the encoder consumes it under autocast (grid.py:45 casts every call).  The copy is cached and
        re-made only when the fp32 master changed through torch (version counter); optim.FusedAdamAMP refreshes it in its own
        update pass and marks it fresh."""
        emb = self.embeddings
        sh = getattr(self, "_half_shadow", None)
        if sh is None or sh.shape != emb.shape or sh.device != emb.device or self._half_version != emb._version:
            sh = emb.detach().half().contiguous()
            self._half_shadow = sh
            self._half_version = emb._version
        return sh

    @torch.no_grad()
    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000, scale=None):
        """Adds the TV gradient of the cells containing `inputs` into embeddings.grad, in place, in fp32
        (grid.py:170-192).  Call after backward (and after GradScaler.unscale_) and before optimizer.step.
        scale (not in the reference signature): a device scalar the term is multiplied by inside the kernel -- the loss scale, when the
        gradients are still scaled -- so that the caller needs no host read of it."""
        D, C = self.input_dim, self.embeddings.shape[1]
        Lv = self.offsets.shape[0] - 1
        S = float(np.log2(self.per_level_scale))
        if inputs is None or inputs.size(0) == 0:
            inputs = torch.rand(B, D, device=self.embeddings.device)
        else:
            inputs = ((inputs + bound) / (2 * bound)).view(-1, D)
            B = inputs.shape[0]
        if self.embeddings.grad is None:
            raise ValueError("grad is None, should be called after loss.backward() and before optimizer.step()!")
        inputs = inputs.float().contiguous()
        emb = self.embeddings.detach().float().contiguous()
        grad = self.embeddings.grad
        if D == 3 and grad.dtype == torch.float32 and grad.is_contiguous():
            if binned_tv(self, inputs, emb, grad, float(weight), scale=scale):
                return
        if scale is not None:
            weight = float(weight) * float(scale)            # scatter fallback (table formats the binned path does not cover): host read
        L.call("n2m_grad_total_variation", _p(inputs), _p(emb), _p(self.embeddings.grad), _p(self.offsets), float(weight), B, D, C,
               Lv, S, int(self.base_resolution), self.gridtype_id, int(bool(self.align_corners)), L.F32, L.stream())
