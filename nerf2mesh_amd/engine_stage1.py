"""Stage-1 step executor: the iteration of trainer.Stage1Trainer's fused path (nerf/utils.py:708-721,745-789 over nerf/renderer.py:816-943)
as a FIXED SEQUENCE OF C-ABI CALLS on preallocated buffers -- no autograd graph, no per-step allocation, the backward written out by hand:

  verts = vertices + offsets -> n2m_to_clip -> n2m_rasterize_forward -> n2m_interpolate_forward x2 (positions, coverage)
  -> covered-pixel list (the ONE host read of the step: the shading kernels are sized by it) -> n2m_gather_rows x2
  -> n2m_grid_encode_forward (colour table, fp16) -> n2m_field_forward (colour + specular networks) -> n2m_scatter_rows (RGB + alpha image)
  -> n2m_antialias_forward -> n2m_stage1_head (clamp, alpha * rgb, ssaa reduction, background blend, loss, face errors AND the gradient
     w.r.t. the antialias output, loss scale applied)
  -> n2m_antialias_backward -> n2m_gather_rows (colour gradient of the covered pixels) -> n2m_field_backward
  -> n2m_grid_encode_backward_binned_pair (colour table alone, overwrite mode, fp16 gradient)
  -> n2m_scatter_rows (coverage gradient) -> n2m_interpolate_backward -> n2m_rasterize_backward (onto the antialias' vertex gradient)
  -> n2m_to_clip_backward -> n2m_laplacian_forward / _backward (smoothness + offset penalty) -> FusedAdamAMP.step

Every kernel is the one the autograd path launches, fed the same inputs; what differs is fp32 association in two sums (the two clip-space
gradients land in one buffer, the three vertex gradients are added in a fixed order) -- tests/test_stage1.py holds the executor to the
distance between two runs of the autograd trainer.  Reference call sites: `render_stage1` (nerf/renderer.py:816-921),
`update_triangles_errors` (:924-943), `train_step` stage-1 branch (nerf/utils.py:708-721), regularisers (:745-789)."""
import os
import time

import numpy as np
import torch

from . import _lib as L
from . import raster as dr
from .fused import SHADING

_p = L.ptr


class Stage1Engine:
    """Drives a trainer.Stage1Trainer's state (model, optimizer, schedule, views, Laplacian) with the fixed launch sequence."""

    @staticmethod
    def supported(tr):
        opt, model = tr.opt, tr.model
        return (tr.amp_adam and tr.fused_head and tr.packed_aa and bool(getattr(opt, "fused_mlp", False)) and not opt.contract
                and not opt.enable_offset_nerf_grad and getattr(model, "individual_dim", 0) == 0 and opt.lambda_lap > 0 and opt.lambda_offsets > 0
                and model.vertices.is_cuda and int(opt.ssaa) in (1, 2))

    def __init__(self, trainer):
        if not self.supported(trainer):
            raise ValueError("Stage1Engine covers the fused stage-1 recipe (fused_mlp, fp16, fused image head, packed antialias); "
                             "use trainer.Stage1Trainer.train_step for anything else")
        self.tr = tr = trainer
        model, opt, dev = tr.model, tr.opt, tr.device
        self.model, self.opt, self.device = model, opt, dev
        ssaa = int(opt.ssaa)
        self.h0, self.w0 = tr.H, tr.W
        self.h, self.w = tr.H * ssaa, tr.W * ssaa
        hw, N, V = self.h * self.w, tr.H * tr.W, model.vertices.shape[0]
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.verts, self.clip, self.d_clip, self.d_verts = f(V, 3), f(V, 4), f(V, 4), f(V, 3)
        self.verts1 = torch.ones(V, 4, dtype=torch.float32, device=dev)                       # [vertex | 1]: position and coverage interpolate together
        self.Lv, self.norm = f(V, 3), f(V)
        self.ones = torch.ones(V, 1, dtype=torch.float32, device=dev)
        self.zbuf = torch.empty(hw, dtype=torch.int64, device=dev)
        self.rast = f(self.h, self.w, 4)
        self.rgba, self.aa, self.d_aa, self.d_rgba = f(hw, 4), f(hw, 4), f(hw, 4), f(hw, 4)
        self.d_rast = f(self.h, self.w, 4)
        self.image, self.depth, self.ws, self.trig, self.loss_px = f(N, 3), f(N), f(N), f(N), f(N)
        nb_head, nb_reg = (N + 255) // 256, (V + 255) // 256
        self.partials = f(nb_head + nb_reg)                # [image head's | regularisers'] workgroup sums: ONE reduction gives the step's loss
        self.partial, self.reg_partial = self.partials[:nb_head], self.partials[nb_head:]
        self.half = torch.tensor(0.5, dtype=torch.float32, device=dev)
        self.zero = torch.zeros((), dtype=torch.float32, device=dev)
        self.cap = 0
        # covered pixels of the frame (nerf/renderer.py:862-863 `mask_flatten`): the library's own order-preserving selection (n2m_select_positive: three
        # launches, the count lands in pinned memory) instead of torch.nonzero's compare + count + select + blocking read (A/B: N2M_S1_NONZERO=torch)
        self.own_select = os.environ.get("N2M_S1_NONZERO", "n2m") != "torch"
        self.idx_buf = torch.empty(hw, dtype=torch.int64, device=dev)
        self.k_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.k_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.k_ready = torch.cuda.Event()
        enc = model.encoder_color
        self.enc = enc
        self.levels = int(enc.num_levels)
        self.merge_levels = int(os.environ.get("N2M_S1_MERGE_LEVELS", "16"))      # (A/B: 9 = the marched-sample default)
        self.geo = (float(np.log2(enc.per_level_scale)), int(enc.base_resolution), enc.gridtype_id, int(bool(enc.align_corners)), enc.interp_id)
        from .gridencoder import _host_offsets
        self.ho = _host_offsets(enc)
        self.g2 = torch.empty(enc.embeddings.shape[0], 2, dtype=torch.float16, device=dev)       # the colour table's gradient: every row defined per step
        self.mlp = [p for m in (model.sigma_net, model.color_net, model.specular_net) for p in m.parameters()]
        n_flat = sum(p.numel() for p in self.mlp)
        self.dw = torch.zeros(n_flat, dtype=torch.float32, device=dev)                          # all-zero between steps (the Adam kernel clears it)
        self.dw_views, o = [], 0
        for p in self.mlp:
            self.dw_views.append(self.dw[o:o + p.numel()].view_as(p))
            o += p.numel()
        self.table = dr._topology(model.triangles)
        self.bound = float(model.bound)
        self.pow2_bound = float(np.log2(self.bound)).is_integer()
        # the optimizer reads the executor's buffers
        opt_ = tr.optimizer
        opt_.half_grads[enc.embeddings] = lambda: self.g2
        for i, p in enumerate(self.mlp):
            opt_.ext_grads[p] = (lambda i=i: self.dw_views[i] if self._live[i] else None)
        self._live = [False] * 7
        # more than one rank (views shard): every gradient carries scale / world and is SUMMED over the ranks before the optimizer,
        # exactly as trainer.Stage1Trainer does it -- fp16 colour-table gradient and vertex gradient as they are, the weight gradients and the
        # non-finite flag in one small bucket
        self.world = int(tr.world)
        self.seed = tr.optimizer.scale if self.world == 1 else torch.empty_like(tr.optimizer.scale)
        # what was captured above belongs to THIS mesh and THIS optimizer: a refinement / broadcast_mesh() re-initialises the trainer (new
        # vertices, triangles, optimizer) and the executor must be rebuilt with it -- train_step checks
        self._captured = (int(model.vertices.shape[0]), int(model.triangles.shape[0]), model.triangles.data_ptr(), id(tr.optimizer))

    def _grow(self, K):
        if K > self.cap:
            self.cap = cap = int(K * 1.25) + 4096
            dev = self.device
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            self.pts, self.dsel, self.x01, self.rgb, self.spec, self.d_rgb = f(cap, 3), f(cap, 3), f(cap, 3), f(cap, 3), f(cap, 3), f(cap, 3)
            self.h2 = torch.empty(self.levels * cap * 2, dtype=torch.float16, device=dev)
            self.d_h2 = torch.empty(self.levels * cap * 2, dtype=torch.float16, device=dev)

    def train_step(self):
        tr, model, opt, dev = self.tr, self.model, self.opt, self.device
        now = (int(model.vertices.shape[0]), int(model.triangles.shape[0]), model.triangles.data_ptr(), id(tr.optimizer))
        if now != self._captured:
            raise RuntimeError("Stage1Engine: the mesh or the optimizer changed since the executor was built (refinement, broadcast_mesh): its buffers, "
                               "antialias topology and gradient hooks are stale -- build a new Stage1Engine(trainer)")
        if not model.training:
            model.train()
        v = tr.views[tr.global_step % len(tr.views)]
        tr.global_step += 1
        rays_o, rays_d, rgba_gt = tr._view(v)
        bg = torch.rand(self.h0 * self.w0, 3, device=dev, generator=tr.gen)
        dirs = tr._dirs.get(v)
        if dirs is None:
            dirs = tr._dirs[v] = model.stage1_dirs(rays_d, self.h0, self.w0).detach()
        o = tr.optimizer
        s = L.stream()
        h, w, h0, w0 = self.h, self.w, self.h0, self.w0
        hw, V, F = h * w, self.verts.shape[0], model.triangles.shape[0]
        tri, mvp = model.triangles, tr.mvps[v]
        shading = SHADING["diffuse" if opt.diffuse_only else "full"]
        with torch.no_grad():
            if self.world > 1:
                torch.div(o.scale, float(self.world), out=self.seed)
            # ---- front half (nerf/renderer.py:855-872)
            torch.add(model.vertices, model.vertices_offsets.detach(), out=self.verts)
            L.call("n2m_to_clip", _p(self.verts), _p(mvp), V, _p(self.clip), s)
            L.call("n2m_rasterize_forward", _p(self.clip), _p(tri), V, F, h, w, _p(self.zbuf), _p(self.rast), s)
            # positions AND coverage of every pixel in one pass, straight into the RGBA image: its alpha channel IS the coverage (0 where nothing
            # is covered, like the other three), and the RGB of the covered pixels overwrites their positions once those have been gathered
            self.verts1[:, :3] = self.verts
            L.call("n2m_interpolate_forward", _p(self.verts1), _p(self.rast), _p(tri), V, F, 4, h, w, _p(self.rgba), s)
            if self.own_select:
                L.call("n2m_select_positive", self.rgba.data_ptr() + 12, hw, 4, _p(self.idx_buf), _p(self.k_dev), s)
                self.k_host.copy_(self.k_dev, non_blocking=True)
                self.k_ready.record()
                t0 = time.perf_counter()
                while not self.k_ready.query():                                                  # the step's one host read (sizes the shading kernels):
                    if time.perf_counter() - t0 > 5e-3:                                          # polled -- a parked thread wakes tens of microseconds late
                        self.k_ready.synchronize()
                        break
                K = int(self.k_host[0])
                idx = self.idx_buf[:K]
            else:
                idx = torch.nonzero(self.rgba[:, 3] > 0, as_tuple=False).squeeze(1)
                K = int(idx.numel())
            model.last_covered = K
            tr.covered_seen += K
            if K > 0:
                self._grow(K)
                pts, dsel, x01, rgb = self.pts[:K], self.dsel[:K], self.x01[:K], self.rgb[:K]
                L.call("n2m_gather_rows_strided", _p(self.rgba), _p(idx), K, 3, 4, _p(pts), 3, s)
                L.call("n2m_gather_rows", _p(dirs), _p(idx), K, 3, _p(dsel), s)
                # ---- colour field of the covered pixels (nerf/renderer.py:875-881, nerf/network.py:159-189 under autocast)
                if self.pow2_bound:
                    torch.add(self.half, pts, alpha=0.5 / self.bound, out=x01)               # == (x + bound) / (2 bound) bit for bit (grid.py:156)
                else:
                    torch.add(pts, self.bound, out=x01)
                    x01.div_(2.0 * self.bound)
                emb2h = self.enc.half_table()
                L.call("n2m_grid_encode_forward", _p(x01), _p(emb2h), _p(self.enc.offsets), _p(self.h2), K, 3, 2, self.levels, self.levels, self.geo[0],
                       self.geo[1], None, self.geo[2], self.geo[3], self.geo[4], L.F16, s)
                ws_ = [p.detach() for p in self.mlp]
                L.call("n2m_field_forward", _p(pts), _p(dsel) if shading != 0 else None, None, _p(self.h2), *[_p(p) for p in ws_], K, shading, 0, None, _p(rgb),
                       _p(self.spec) if shading != 0 else None, s)
                L.call("n2m_scatter_rows_strided", _p(rgb), _p(idx), K, 3, 3, _p(self.rgba), 4, s)
            L.call("n2m_antialias_forward", _p(self.rgba), _p(self.rast), _p(self.clip), _p(tri), _p(self.table), self.table.shape[0], V, F, 4, h, w,
                   _p(self.aa), s)
            # ---- image head: loss, face errors, and d loss / d antialias output (scaled by the loss scale) in one launch
            te = (model.triangles_errors, model.triangles_errors_cnt) if opt.refine else (None, None)
            L.call("n2m_stage1_head", self.aa.data_ptr() + 12, _p(self.aa), _p(self.rast), h0, w0, int(opt.ssaa), _p(rgba_gt), _p(bg), 0.0, float(opt.lambda_rgb),
                   float(max(opt.lambda_mask, 0.0)), _p(self.image), _p(self.depth), _p(self.ws), _p(self.trig), _p(self.loss_px), self.d_aa.data_ptr() + 12,
                   _p(self.d_aa), _p(self.partial), _p(te[0]), _p(te[1]), 1, _p(self.seed), _p(self.d_rgba), s)
            # ---- backward
            self.d_clip.zero_()
            L.call("n2m_antialias_backward_seeded", _p(self.rgba), _p(self.rast), _p(self.clip), _p(tri), _p(self.table), self.table.shape[0], _p(self.d_aa), V, F, 4, h, w,
                   float(opt.pos_gradient_boost), _p(self.d_rgba), _p(self.d_clip), s)
            self._live = [False] * 7
            if K > 0:
                d_rgb = self.d_rgb[:K]
                L.call("n2m_gather_rows_strided", _p(self.d_rgba), _p(idx), K, 3, 4, _p(d_rgb), 3, s)
                L.call("n2m_field_backward", _p(pts), _p(dsel) if shading != 0 else None, None, _p(self.h2), *[_p(p) for p in ws_], K, shading, 0, None, _p(d_rgb),
                       None, None, _p(self.d_h2), *[_p(g) for g in self.dw_views], _p(o.found_inf), s)
                self._live = [False, False, True, True, True, shading != 0, shading != 0]
                need = L.lib().n2m_grid_binned_pair_workspace_bytes(K, self.levels, self.ho.ctypes.data)
                wsb = L.workspace(dev, need)
                L.grid_backward_config(1, 1.0)
                # consecutive covered pixels share cells on EVERY level (a frame samples the surface at ~1 / 1600 of its extent): merge runs on all
                # sixteen for this call (sticky thread-local: set in front of the call, restored behind it whatever happens)
                L.call("n2m_grid_backward_merge_levels", self.merge_levels)
                try:
                    L.call("n2m_grid_encode_backward_binned_pair", None, _p(self.d_h2), _p(x01), self.ho.ctypes.data, None, _p(self.g2), K, self.levels, self.levels,
                           self.geo[0], self.geo[1], self.geo[2], self.geo[3], self.geo[4], None, 0.0, 0.0, 1.0, None, _p(o.found_inf), 1.0, 0.0, 1, _p(wsb),
                           wsb.numel(), s)
                finally:
                    L.call("n2m_grid_backward_merge_levels", 0)
                if self.world > 1:        # the two large sums travel while the vertex path finishes
                    tok = tr.sync.all_reduce_sum_begin([self.g2], [self.dw])
                # coverage: its gradient reaches the vertex positions through the barycentrics (the interpolated attribute is the constant 1)
                L.call("n2m_interpolate_backward_strided", _p(self.ones), _p(self.rast), _p(tri), self.d_rgba.data_ptr() + 12, 4, V, F, 1, h, w, None,
                       _p(self.d_rast), s)
                L.call("n2m_rasterize_backward", _p(self.clip), _p(tri), _p(self.rast), _p(self.d_rast), V, F, h, w, _p(self.d_clip), s)
            else:
                self.g2.zero_()
                if self.world > 1:
                    tok = tr.sync.all_reduce_sum_begin([self.g2], [self.dw])
            L.call("n2m_to_clip_backward", _p(self.d_clip), _p(mvp), V, _p(self.d_verts), s)
            # ---- mesh regularisers (nerf/utils.py:761-789): value and gradient, the gradient scaled like everything else
            off = model.vertices_offsets.detach()
            n_in = int(model.v_cumsum[1]) if opt.bound > 1 else None
            if n_in is None or n_in >= V or n_in <= 0:
                n_in, w_in, w_out = V, float(opt.lambda_offsets) / V, 0.0
            else:
                w_in, w_out = float(opt.lambda_offsets) / n_in, 0.1 * float(opt.lambda_offsets) / (V - n_in)
            lap = tr.laplacian
            Npx = float(h0 * w0)
            # the VALUE's weights carry a factor h0 w0, so that (image-head sums + regulariser sums) / (h0 w0) is the loss in one reduction
            L.call("n2m_laplacian_forward", _p(self.verts), _p(lap.row_ptr), _p(lap.col), V, _p(off), float(opt.lambda_lap) * Npx, w_in * Npx, w_out * Npx, n_in,
                   _p(self.Lv), _p(self.norm), _p(self.reg_partial), s)
            # d offsets = rendering gradient of the vertices + smoothness + offset penalty, and its non-finite check, in one pass
            L.call("n2m_laplacian_backward_acc", _p(self.Lv), _p(self.norm), _p(lap.row_ptr), _p(lap.col), V, _p(self.seed), float(opt.lambda_lap), _p(off), w_in,
                   w_out, n_in, _p(self.d_verts), _p(o.found_inf), s)
            model.vertices_offsets.grad = self.d_verts
            loss = self.partials.sum() / Npx
            # ---- optimizer: colour table (fp16 gradient) and weight gradients were checked by the kernels that produced them
            flagged = [model.vertices_offsets] + (([self.enc.embeddings] + [p for p, lv in zip(self.mlp, self._live) if lv]) if K > 0 else [])
            if self.world > 1:
                # a rank whose view shows nothing still takes part (its gradients are zeros); a sum of finite values can overflow, so the
                # optimizer checks the reduced gradients again
                tr.sync.all_reduce_sum_end(tok)
                tr.sync.all_reduce_sum([self.d_verts, o.found_inf], [])
                self._live = [False, False, True, True, True, shading != 0, shading != 0]
                flagged = []
            o.step(flagged=flagged)
        tr.scheduler.step()
        return loss
