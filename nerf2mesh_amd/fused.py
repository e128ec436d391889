"""Fused field evaluation: both hash-grid encodes + sigma_net / color_net / specular_net in three kernels forward
(two encodes, one MFMA field kernel) and three backward (one MFMA field kernel, two LDS-privatised table scatters).

Opt-in replacement (`opt.fused_mlp`) for NeRFNetwork.forward / .density as nerf/network.py:81-108,159-189 composes
them from nn.Linear calls; the unfused path stays the parity baseline (tests/test_mlp_parity.py).  The C ABI is
include/n2m_mlp.h.  Semantics follow the autocast graph of the reference: density table fp32, colour table fp16,
fp16 MLP operands with fp32 accumulation, exp in fp32.
"""
import numpy as np
import torch
from torch.autograd import Function

from . import _lib as L

_p = L.ptr
PACKED_FORWARD = True           # gather both tables from one packed copy (8-byte rows) when the network provides it
PAIR_FORWARD = True             # one forward launch for the two tables when their geometry is identical
PAIR_BACKWARD = True            # one shared fill for the two tables when their geometry is identical
CONCURRENT_BACKWARD = False     # A/B switch: run the two table backward passes on two streams (measured: no gain, the kernels already fill the chip)
SHADING = {"diffuse": 0, "full": 1, "specular": 2}


def _bind():
    L.lib()


def _encode_lm(x01, emb, enc, max_level):
    """Level-major features [L, B, C] (the reference kernel's own layout): fastest to write, and what the fused MLP reads."""
    B = x01.shape[0]
    Lv, C = enc.num_levels, emb.shape[1]
    out = torch.empty(Lv, B, C, device=x01.device, dtype=emb.dtype) if max_level >= Lv else torch.zeros(Lv, B, C, device=x01.device, dtype=emb.dtype)
    L.call("n2m_grid_encode_forward", _p(x01), _p(emb), _p(enc.offsets), _p(out), B, 3, C, Lv, max_level,
           float(np.log2(enc.per_level_scale)), int(enc.base_resolution), None, enc.gridtype_id, int(bool(enc.align_corners)), enc.interp_id,
           L.F16 if emb.dtype == torch.float16 else L.F32, L.stream())
    return out


def _affine(bound):
    """(scale, offset) with x01 = x * scale + offset, when that is bit-identical to grid.py:156's (x + bound) / (2 * bound): bound a
    power of two (every bound nerf2mesh uses).  None otherwise: the caller then materialises x01 with torch."""
    import math
    m, _ = math.frexp(float(bound))
    return (1.0 / (2.0 * float(bound)), 0.5) if m == 0.5 else None


def _encode_lm_pair(x01, emb1, emb2h, net, max_level, in_affine=(1.0, 0.0)):
    """(h1 [L,B,1] f32, h2 [L,B,2] f16) from one launch when the two encoders share their geometry, else None.
    in_affine: the kernel reads x01 * scale + offset (pass the raw points and _affine(bound) to skip the torch normalisation)."""
    from .gridencoder import same_geometry
    e1, e2 = net.encoder, net.encoder_color
    if not (PAIR_FORWARD and emb1.dtype == torch.float32 and emb1.shape[1] == 1 and emb2h.dtype == torch.float16 and emb2h.shape[1] == 2
            and hasattr(e1, "host_offsets") and hasattr(e2, "host_offsets") and same_geometry(e1, e2)):
        return None
    B, Lv = x01.shape[0], e1.num_levels
    mk = torch.empty if max_level >= Lv else torch.zeros
    h1 = mk(Lv, B, 1, device=x01.device, dtype=torch.float32)
    h2 = mk(Lv, B, 2, device=x01.device, dtype=torch.float16)
    L.call("n2m_grid_encode_forward_pair", _p(x01), _p(emb1), _p(emb2h), _p(e1.offsets), _p(h1), _p(h2), B, Lv, max_level,
           float(np.log2(e1.per_level_scale)), int(e1.base_resolution), e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id,
           float(in_affine[0]), float(in_affine[1]), L.stream())
    return h1, h2


def _encode_lm_packed(x, packed, net, max_level, in_affine, density_only=False):
    """_encode_lm_pair from the packed copy of the two tables (network.packed_tables); density_only: h2 is None (the kernel reads the
    density column alone -- measured SLOWER than the plain table for the occupancy refresh's Morton-ordered points, 488 against 382 us for
    2 M points: their 4-byte gathers coalesce in the L1, the 16-byte row pairs do not; kept for the test that pins the two paths together)."""
    e1 = net.encoder
    B, Lv = x.shape[0], e1.num_levels
    mk = torch.empty if max_level >= Lv else torch.zeros
    h1 = mk(Lv, B, 1, device=x.device, dtype=torch.float32)
    h2 = None if density_only else mk(Lv, B, 2, device=x.device, dtype=torch.float16)
    L.call("n2m_grid_encode_forward_packed", _p(x), _p(packed), _p(e1.offsets), _p(h1), _p(h2), B, Lv, max_level,
           float(np.log2(e1.per_level_scale)), int(e1.base_resolution), e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id,
           float(in_affine[0]), float(in_affine[1]), L.stream())
    return h1, h2


def _encode_backward_lm(grad_lm, x01, emb, enc, max_level, ws_slot=0):
    B = x01.shape[0]
    Lv, C = enc.num_levels, emb.shape[1]
    g = torch.zeros_like(emb)
    from .gridencoder import binned_backward
    req = getattr(enc, "tv_request", None)          # set by the trainer: fold the TV gradient into this backward
    if req is not None and (req.get("done") or (req.get("rows") is not None and req["rows"] != B)):
        req = None                                  # already folded in, or a different batch of points (SDF: the finite-difference offsets)
    amp = getattr(enc, "amp_request", None)         # set by optim.FusedAdamAMP users: {"found_inf": tensor, "flagged": bool}
    finf = amp["found_inf"] if amp is not None else None
    if req is not None and binned_backward(enc, grad_lm, x01, g, max_level, tv=(emb, req["weight"], req["weight_outer"], req["inner01"], req["scale"]),
                                           found_inf=finf, ws_slot=ws_slot):
        req["done"] = True
        if amp is not None:
            amp["flagged"] = True
        return g
    if binned_backward(enc, grad_lm, x01, g, max_level, found_inf=finf, ws_slot=ws_slot):
        if amp is not None:
            amp["flagged"] = True
        return g
    L.call("n2m_grid_encode_backward", _p(grad_lm), _p(x01), _p(emb), _p(enc.offsets), _p(g), B, 3, C, Lv, max_level,
           float(np.log2(enc.per_level_scale)), int(enc.base_resolution), None, None, enc.gridtype_id, int(bool(enc.align_corners)),
           enc.interp_id, L.F16 if emb.dtype == torch.float16 else L.F32, L.stream())
    return g


def _encode_backward_pair(d_h1, d_h2, x01, emb1, emb2h, net, max_level, in_affine=(1.0, 0.0)):
    # emb2h may be None (packed forward): only its shape / dtype are needed here
    """(g1, g2) through the shared-fill kernel, or (None, None) when it does not apply."""
    from .gridencoder import binned_backward_pair
    enc1, enc2 = net.encoder, net.encoder_color
    g1 = torch.empty_like(emb1)                        # overwrite mode: the kernels define every row, no zero-fill
    g2 = torch.empty(emb1.shape[0], 2, dtype=torch.float16, device=emb1.device)
    req = getattr(enc1, "tv_request", None)
    if req is not None and (req.get("done") or (req.get("rows") is not None and req["rows"] != x01.shape[0])):
        req = None
    amp1, amp2 = getattr(enc1, "amp_request", None), getattr(enc2, "amp_request", None)
    finf = amp1["found_inf"] if amp1 is not None else (amp2["found_inf"] if amp2 is not None else None)
    tv = (emb1, req["weight"], req["weight_outer"], req["inner01"], req["scale"]) if req is not None else None
    if not binned_backward_pair(enc1, enc2, d_h1, d_h2, x01, g1, g2, max_level, tv=tv, found_inf=finf, in_affine=in_affine, overwrite=True):
        return None, None
    if req is not None:
        req["done"] = True
    for amp in (amp1, amp2):
        if amp is not None and finf is amp["found_inf"]:
            amp["flagged"] = True
    return g1, g2


class _fused_field(Function):
    @staticmethod
    def forward(ctx, xyz, dirs, emb1, emb2, w0, w1, w2, w3, w4, w5, w6, net, shading, want_color, want_density, normalize_dirs):
        _bind()
        xyz = xyz.float().contiguous()
        M = xyz.shape[0]
        bound, max_level = float(net.bound), int(min(net.max_level, net.encoder.num_levels))
        aff = _affine(bound)
        x01 = None                                        # materialised only for the code paths that need [0,1] points
        sigma = h1 = None
        rgb = spec = h2 = emb2h = None
        emb1 = emb1.float().contiguous() if want_density else None
        both = None
        packed = net.packed_tables() if (PACKED_FORWARD and want_density and want_color and aff is not None and hasattr(net, "packed_tables")) else None
        if packed is not None:
            both = _encode_lm_packed(xyz, packed, net, max_level, aff)
        elif want_color:
            emb2h = net.encoder_color.half_table() if hasattr(net.encoder_color, "half_table") else emb2.half().contiguous()   # grid.py:45
        if both is None and want_density and want_color:
            if aff is not None:
                both = _encode_lm_pair(xyz, emb1, emb2h, net, max_level, aff)      # normalisation folded into the kernel
            else:
                x01 = (xyz + bound) / (2 * bound)
                both = _encode_lm_pair(x01, emb1, emb2h, net, max_level)
        if both is not None:
            h1, h2 = both
        elif x01 is None:
            x01 = (xyz + bound) / (2 * bound)
        if want_density:
            if h1 is None:
                h1 = _encode_lm(x01, emb1, net.encoder, max_level)
            sigma = torch.empty(M, dtype=torch.float32, device=xyz.device)
        ws = [w.float().contiguous() for w in (w0, w1, w2, w3, w4, w5, w6)]
        if want_color:
            if h2 is None:
                h2 = _encode_lm(x01, emb2h, net.encoder_color, max_level)
            dirs = dirs.float().contiguous() if shading != 0 else None
            rgb = torch.empty(M, 3, dtype=torch.float32, device=xyz.device)
            spec = torch.empty(M, 3, dtype=torch.float32, device=xyz.device) if shading != 0 else None
        flags = int(bool(normalize_dirs)) | (2 if getattr(net.opt, "sdf", False) else 0)      # bit 1: SDF head, sigma = raw fp16 output
        L.call("n2m_field_forward", _p(xyz), _p(dirs), _p(h1), _p(h2), *[_p(w) for w in ws], M, shading, flags, _p(sigma), _p(rgb),
               _p(spec), L.stream())
        ctx.net, ctx.shading, ctx.want_color, ctx.max_level, ctx.want_density = net, shading, want_color, max_level, want_density
        ctx.normalize_dirs = flags
        ctx.bound = bound
        ctx.save_for_backward(xyz, dirs, x01, h1, h2, emb1, emb2h, *ws)
        if not want_color:
            return sigma
        outs = ([sigma] if want_density else []) + [rgb] + ([spec] if spec is not None else [])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        xyz, dirs, x01, h1, h2, emb1, emb2h, *ws = ctx.saved_tensors
        net, shading, want_color, max_level, want_density = ctx.net, ctx.shading, ctx.want_color, ctx.max_level, ctx.want_density
        grads = list(grads)
        d_sigma = grads.pop(0) if want_density else None
        d_rgb = grads.pop(0) if want_color else None
        d_spec = grads.pop(0) if grads else None
        M = xyz.shape[0]
        dev = xyz.device
        d_h1 = None
        if want_density:
            d_sigma = torch.zeros(M, device=dev) if d_sigma is None else d_sigma.float().contiguous()
            d_h1 = torch.zeros(16, M, dtype=torch.float32, device=dev) if max_level < 16 else torch.empty(16, M, dtype=torch.float32, device=dev)
        d_h2 = None
        if want_color:
            d_rgb = torch.zeros(M, 3, device=dev) if d_rgb is None else d_rgb.float().contiguous()
            d_spec = d_spec.float().contiguous() if (d_spec is not None and shading != 0) else None
            d_h2 = torch.empty(16, M, 2, dtype=torch.float16, device=dev)
        else:
            d_rgb = d_spec = None
        amp = getattr(net, "amp_request", None)          # optim.FusedAdamAMP: weight-gradient finiteness is checked by the kernel
        n_flat = sum(w.numel() for w in ws)
        persistent = amp is not None and amp.get("persistent_dw", False)
        if persistent:
            # the seven dW live in one buffer that is all-zero between steps: the kernel adds into it, the optimizer reads it through
            # ext_grads and its Adam kernel clears it again -- no fill launch, no AccumulateGrad nodes
            flat = getattr(net, "_dw_flat", None)
            if flat is None or flat.numel() != n_flat or flat.device != dev:
                flat = net._dw_flat = torch.zeros(n_flat, dtype=torch.float32, device=dev)
        else:
            flat = torch.zeros(n_flat, dtype=torch.float32, device=dev)      # one fill for the seven dW
        dws, o = [], 0
        for w in ws:
            dws.append(flat[o:o + w.numel()].view_as(w))
            o += w.numel()
        if persistent:
            # weights whose gradient this call does not produce stay out of the optimizer step (torch: grad is None), so that their
            # Adam step count starts when they first train -- the specular head after opt.diffuse_step
            live = [want_density] * 2 + [want_color] * 3 + [want_color and shading != 0] * 2
            prev = amp.get("dw_views") or [None] * 7        # an earlier backward call of the same step (SDF: field + normals) keeps its weights live
            amp["dw_flat"], amp["dw_views"] = flat, [g if (ok or pv is not None) else None for g, ok, pv in zip(dws, live, prev)]
        L.call("n2m_field_backward", _p(xyz), _p(dirs), _p(h1), _p(h2), *[_p(w) for w in ws], M, shading, ctx.normalize_dirs, _p(d_sigma), _p(d_rgb),
               _p(d_spec), _p(d_h1), _p(d_h2), *[_p(g) for g in dws], _p(amp["found_inf"]) if amp is not None else None, L.stream())
        if amp is not None:
            amp["flagged"] = True
        # both tables: one shared fill when their geometry is identical (it is for nerf2mesh), else one backward per table
        g1 = g2 = None
        aff = _affine(ctx.bound)
        if want_density and want_color and PAIR_BACKWARD:
            if x01 is None and aff is not None:
                g1, g2 = _encode_backward_pair(d_h1, d_h2, xyz, emb1, emb2h, net, max_level, aff)
            else:
                if x01 is None:
                    x01 = (xyz + ctx.bound) / (2 * ctx.bound)
                g1, g2 = _encode_backward_pair(d_h1, d_h2, x01, emb1, emb2h, net, max_level)
        if g1 is None:
            if x01 is None:
                x01 = (xyz + ctx.bound) / (2 * ctx.bound)
            if want_color and emb2h is None:
                emb2h = net.encoder_color.half_table()
            side = L.side_stream(dev) if (want_density and want_color and CONCURRENT_BACKWARD) else None
            if want_color and side is not None:
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    g2 = _encode_backward_lm(d_h2, x01, emb2h, net.encoder_color, max_level, ws_slot=1)
                for t_ in (g2, d_h2, x01, emb2h):
                    t_.record_stream(side)
            g1 = _encode_backward_lm(d_h1, x01, emb1, net.encoder, max_level) if want_density else None
            if want_color:
                if side is not None:
                    torch.cuda.current_stream().wait_stream(side)
                else:
                    g2 = _encode_backward_lm(d_h2, x01, emb2h, net.encoder_color, max_level)
        if want_color:
            amp2 = getattr(net.encoder_color, "amp_request", None)
            if amp2 is not None and amp2.get("keep_half"):
                amp2["grad_half"] = g2           # the optimizer reads the fp16 gradient directly; autograd gets no fp32 copy
                g2 = None
            else:
                g2 = g2.float()
        if persistent:
            dws = [None] * 7
        elif not want_color:
            dws[2:] = [None] * 5
        elif shading == 0:
            dws[5:] = [None] * 2
        if not want_density:
            dws[:2] = [None] * 2
        return (None, None, g1, g2, *dws, None, None, None, None, None)


def fused_field(net, xyz, dirs, shading="full", normalize_dirs=False):
    """sigma [M], rgb [M,3], specular [M,3] | None -- NeRFNetwork.forward without individual codes.
    normalize_dirs: `dirs` are raw ray directions and get safe_normalize'd inside the kernel."""
    sh = SHADING[shading]
    out = _fused_field.apply(xyz, dirs, net.encoder.embeddings, net.encoder_color.embeddings, net.sigma_net.net[0].weight,
                             net.sigma_net.net[1].weight, net.color_net.net[0].weight, net.color_net.net[1].weight,
                             net.color_net.net[2].weight, net.specular_net.net[0].weight, net.specular_net.net[1].weight, net, sh, True, True, normalize_dirs)
    if sh == 0:
        return out[0], out[1], None
    return out


def fused_density(net, xyz):
    """sigma [M] only (occupancy refresh, nerf/renderer.py:1112-1113)."""
    return _fused_field.apply(xyz, None, net.encoder.embeddings, net.encoder_color.embeddings, net.sigma_net.net[0].weight,
                              net.sigma_net.net[1].weight, net.color_net.net[0].weight, net.color_net.net[1].weight,
                              net.color_net.net[2].weight, net.specular_net.net[0].weight, net.specular_net.net[1].weight, net, 0, False, True, False)


def fused_color(net, xyz, dirs, shading="full"):
    """rgb [M,3], specular [M,3] | None -- NeRFNetwork.rgb without individual codes (stage 1: nerf/renderer.py:875-881)."""
    sh = SHADING[shading]
    out = _fused_field.apply(xyz, dirs, net.encoder.embeddings, net.encoder_color.embeddings, net.sigma_net.net[0].weight,
                             net.sigma_net.net[1].weight, net.color_net.net[0].weight, net.color_net.net[1].weight,
                             net.color_net.net[2].weight, net.specular_net.net[0].weight, net.specular_net.net[1].weight, net, sh, True, False, False)
    return (out[0], None) if sh == 0 else (out[0], out[1])
