"""Minimal stage-0 training loop: the per-iteration behaviour of the reference's Trainer.train_one_epoch /
train_step / post_train_step (nerf/utils.py:628-823,1132-1211) and main.py:221-241, without its I/O
(checkpoints, logging, tensorboard, evaluation images stay the reference's Python).

One iteration = [every 16th: occupancy refresh] -> sample rays -> render (march, encode, MLPs, composite) -> MSE (+ mask,
+ specular reg) -> scaled backward -> [multi-GPU: gradient all-reduce] -> unscale -> in-place TV gradient -> Adam -> LR step.
Rays come from nerf2mesh_amd.synthetic (no dataset ships with the container); everything stays on the device.
"""
import os

import torch
import torch.nn.functional as F

from . import synthetic
from .losses import photo_loss
from .parallel import GradSync, shard_refresh_default as _shard_refresh_default


class LambdaLR:
    """torch.optim.lr_scheduler.LambdaLR for the one use main.py:239 makes of it (lr = initial_lr * f(step), stepped once per iteration),
    without the 0.18 ms of Python its step() spends per call -- a tenth of a host-bound stage-1 step."""

    def __init__(self, optimizer, lr_lambda):
        self.optimizer, self.lr_lambda, self.last_epoch = optimizer, lr_lambda, 0
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self._apply()

    def _apply(self):
        f = self.lr_lambda(self.last_epoch)
        for g in self.optimizer.param_groups:
            g["lr"] = g["initial_lr"] * f

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = int(sd["last_epoch"])
        self._apply()


class Stage0Trainer:
    def __init__(self, model, opt, poses, device, rank=0, world_size=1, seed=0, ema_decay=0.95):
        self.model, self.opt, self.device = model.to(device), opt, device
        self.poses = poses.to(device)
        # EMA of the parameters: main.py:241 (0.95 for stage 0), nerf/utils.py:544-545; one update per epoch = len(loader) steps (:1213-1214).
        # Device tensors only (the update is a HIP kernel); the CPU drivers of tests/test_parallel.py run without it.
        self.ema = None
        self.epoch_len = max(1, int(poses.shape[0]))
        if ema_decay is not None and torch.device(device).type == "cuda":
            from .ema import ExponentialMovingAverage
            self.ema = ExponentialMovingAverage(model.parameters(), decay=ema_decay)
        self.rank, self.world = rank, world_size
        self.global_step = 0
        self.num_rays = opt.num_rays
        # every rank draws its own rays.  Everything random about a batch (pixels, march jitter, background) comes
        # from ONE draw per batch of this generator, so a driver that prepares batches further ahead (engine.Stage0Engine) consumes
        # identical numbers per batch
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed + rank)
        # main.py:221 Adam(eps=1e-15) + nerf/utils.py:506 GradScaler.  Single GPU with the fused field: optim.FusedAdamAMP does both
        # in two launches and takes the inf/nan verdict from the kernels that produce the gradients.
        self.amp_adam = device.type == "cuda" and bool(getattr(opt, "fused_mlp", False)) and bool(opt.fp16) and getattr(opt, "ind_dim", 0) == 0
        if self.amp_adam:
            from .optim import FusedAdamAMP
            self.optimizer = FusedAdamAMP(model.get_params(opt.lr), eps=1e-15, amp=bool(opt.fp16))
            enc, encc = model.encoder, model.encoder_color
            self._amp = {}

            def shadow_density():          # column 0 of the packed table the forward gathers from (None: no packed table)
                pk = model.packed_tables()
                return (pk, 2) if pk is not None else None

            def shadow_color():            # column 1 of the packed table, else the plain fp16 copy
                pk = model.packed_tables()
                if pk is None:
                    return encc.half_table()
                encc._half_version = -1    # the plain fp16 copy is not refreshed any more: rebuild it on next use
                return (pk, 3)

            self.optimizer.shadows[enc.embeddings] = shadow_density
            self.optimizer.shadows[encc.embeddings] = shadow_color
            self.optimizer.half_grads[encc.embeddings] = lambda: self._amp.get("color", {}).get("grad_half")
            self._mlp_params = [p for m in (model.sigma_net, model.color_net, model.specular_net) for p in m.parameters()]
            for i, p in enumerate(self._mlp_params):   # the fused field backward adds the seven dW into one persistent buffer (fused._fused_field)
                self.optimizer.ext_grads[p] = lambda i=i: (self._amp.get("mlp", {}).get("dw_views") or [None] * 7)[i]
        else:
            self.optimizer = torch.optim.Adam(model.get_params(opt.lr), eps=1e-15, fused=(device.type == "cuda"))
        iters = opt.iters
        self.scheduler = LambdaLR(
            self.optimizer, lambda it: 0.01 + 0.99 * (it / 500) if it <= 500 else 0.1 ** ((it - 500) / (iters - 500)))   # main.py:239
        self.scaler = torch.amp.GradScaler("cuda", enabled=bool(opt.fp16) and device.type == "cuda")
        self.sync = GradSync(model, world_size) if world_size > 1 else None
        if world_size > 1:      # the occupancy refresh's density query sharded over the ranks by Morton range (renderer.update_extra_state)
            self.model.refresh_shard = (rank, world_size) if _shard_refresh_default() else None
        self.scene = getattr(opt, "scene", "lego")
        self.boxes = synthetic.boxes(device, self.scene)
        # --enable_cam_near_far (main.py:40): every ray is clamped to its camera's sparse-point depth range (nerf/renderer.py:689-691)
        self.cam_near_far = synthetic.cam_near_far(self.poses, self.scene) if getattr(opt, "enable_cam_near_far", False) else None
        self._loss_sum = torch.zeros((), device=device)
        self._loss_pending = []
        self.samples_seen = 0
        self.rays_seen = 0
        self.last_num_points = 0
        self.preload = True           # ground-truth images resident on the device, batches gathered from them
        self.images = None
        self.fused_tv = True          # TV gradient folded into the density encoder's binned backward
        self._one = torch.ones((), device=device)
        self.fused_loss = True        # losses.photo_loss instead of the torch graph of nerf/utils.py:658-683
        self.pipeline = True          # issue march pass 1 of the next batch one step ahead (results are identical)
        self.overlap_march = True     # ... on a second stream, next to this step's backward + optimizer kernels (single-rank path)
        self.early_march = True       # ... and already at the start of the step (A/B switch; measured equal to issuing it before the backward)
        self._next = None

    @property
    def loss_acc(self):
        """Sum of the training losses so far (device scalar)."""
        if self._loss_pending:
            self._loss_sum = self._loss_sum + torch.stack(self._loss_pending).sum()
            self._loss_pending = []
        return self._loss_sum

    def mark_untrained(self):
        if self.opt.mark_untrained:
            f = synthetic.LEGO_FOCAL
            self.model.mark_untrained_grid(self.poses, (f, f, synthetic.LEGO_HW / 2, synthetic.LEGO_HW / 2), cam_near_far=self.cam_near_far)

    def batch(self):
        """(rays_o, rays_d, rgba, noises, bg) of the next batch: ONE draw of [num_rays, 6] uniforms from the ray generator, turned into
        pixels, rays, ground truth, march jitter and random background by synthetic.batch_from_uniforms."""
        if self.images is None:
            self.images = synthetic.preload_images(self.poses, self.boxes)     # nerf/provider.py:224-233 (`preload`)
        u = torch.rand(self.num_rays, 6, device=self.device, generator=self.gen)
        rays_o, rays_d, rgba, nears, fars, noises, bg = synthetic.batch_from_uniforms(self.poses, self.images, u, self.model.aabb_train,
                                                                                      self.model.min_near, cam_near_far=self.cam_near_far)
        self._nears_fars = (nears, fars) if self.cam_near_far is not None else None
        return rays_o, rays_d, rgba, noises, bg

    def _prepare(self):
        """Occupancy refresh (every 16th step, nerf/utils.py:1155-1156) + next batch + march pass 1 for it."""
        opt, model = self.opt, self.model
        if self.global_step % opt.update_extra_interval == 0:
            if self.sync is not None:
                self.sync.sync_rng_for_grid_update(self.global_step)
            model.update_extra_state()
        rays_o, rays_d, images, noises, bg = self.batch()
        # sample buffers for the speculative write pass: a quarter above the last batch (adaptive_num_rays steers M towards
        # opt.num_points, nerf/utils.py:796-797); a batch that still does not fit is re-marched exactly by finish()
        expect = 0 if self.last_num_points <= 0 else ((int(1.25 * max(self.last_num_points, 1024)) + 1023) // 1024) * 1024
        ticket = model.march_ahead(rays_o, rays_d, dt_gamma=opt.dt_gamma, perturb=True, max_steps=opt.max_steps,
                                   expect_points=expect, noises=noises, nears_fars=self._nears_fars) if self.pipeline else None
        return rays_o, rays_d, images, ticket, (bg if opt.background != "white" else 1), noises

    def _prepare_overlapped(self):
        """_prepare() on the side stream.  The next batch's ray generation and march pass 1 read only the camera set and the occupancy
        bit field, so they may run NEXT TO this step's backward and Adam kernels: the count pass is latency-bound (a wave per ray,
        ~6 GB/s) and leaves the memory system to them.  Ordering: the side stream starts behind everything already queued on the main
        stream (covers an occupancy refresh), the main stream picks the results up through the ticket's event
        (march_rays_train_finish).  The random streams are consumed in the same order as in the serial schedule, so results are identical."""
        from . import _lib as L
        main = torch.cuda.current_stream(self.device)
        side = L.side_stream(self.device, slot=2)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            nxt = self._prepare()
        rays_o, rays_d, images, ticket, bg, noises = nxt
        for t in (rays_o, rays_d, images, bg, noises, ticket.rays, ticket.counter, ticket.noises) + tuple(ticket.keep) + tuple(ticket.spec or ()):
            if torch.is_tensor(t):
                t.record_stream(main)          # allocated on the side stream, consumed on the main one
        return nxt

    def train_step(self):
        opt, model = self.opt, self.model
        if not model.training:
            model.train()
        if self._next is None:
            self._next = self._prepare()
        rays_o, rays_d, images, ticket, bg_color, noises = self._next
        nears_fars = self._nears_fars                      # of THIS batch (batch() sets it; the overlapped preparation below replaces it)
        self._next = None
        self.global_step += 1
        self.optimizer.zero_grad(set_to_none=True)
        N = rays_o.shape[0]
        if opt.sdf:
            opt.cos_anneal_ratio = min(1, self.global_step / (0.5 * opt.iters))
            opt.normal_anneal_epsilon = 1e-1 * (1 - min(0.999, self.global_step / (0.5 * opt.iters)))
        if opt.progressive_level:
            model.max_level = 4 + int(12 * min(1, self.global_step / (0.5 * opt.iters)))
        shading = "diffuse" if (self.global_step < opt.diffuse_step or opt.diffuse_only) else "full"

        adapted = False
        if self.overlap_march and self.early_march and self.pipeline and self.amp_adam and self.sync is None and ticket is not None \
                and torch.device(self.device).type == "cuda" and self.global_step % opt.update_extra_interval != 0:
            # The sample count of THIS batch is all the next batch waits for (adaptive_num_rays): read it now and put the next
            # batch's ray generation + march on the side stream before this step's own kernels are queued.  They then run beside
            # the forward pass instead of queueing up behind the field backward (one wave per SIMD, it shares the chip with nobody),
            # the count is back long before the host needs it, and the host stays a step tail ahead of the GPU.  Same draws from
            # the same generators in the same order as the serial schedule (bg_color above, then the batch): identical results.
            from . import raymarching
            ticket = raymarching.march_rays_train_finish(ticket)
            M0 = ticket[0].shape[0]
            self.last_num_points = M0
            if opt.adaptive_num_rays and M0 > 0:                             # nerf/utils.py:796-797
                self.num_rays = max(1, int(round((opt.num_points / M0) * self.num_rays)))
            adapted = True
            self._next = self._prepare_overlapped()

        # (without the pipelined march the per-view near / far clamp of --enable_cam_near_far has to reach render() itself)
        out = model.render(rays_o, rays_d, bg_color=bg_color, perturb=True, shading=shading, dt_gamma=opt.dt_gamma,
                           max_steps=opt.max_steps, ticket=ticket, blend_bg=not self.fused_loss,
                           nears_fars=nears_fars if ticket is None else None)
        if self.fused_loss:
            # background blend + ground-truth compositing + rgb/mask MSE + mean in one kernel (nerf/utils.py:658-683)
            loss = photo_loss(out["image"], out["weights_sum"], images, bg_color, opt.lambda_rgb, max(opt.lambda_mask, 0.0))
        else:
            gt_mask = images[..., 3:]
            gt_rgb = images[..., :3] * gt_mask + bg_color * (1 - gt_mask)
            loss = opt.lambda_rgb * F.mse_loss(out["image"], gt_rgb, reduction="none").mean(-1)
            if opt.lambda_mask > 0:
                loss = loss + opt.lambda_mask * F.mse_loss(out["weights_sum"], gt_mask.squeeze(1), reduction="none")
            loss = loss.mean()
        if opt.lambda_entropy > 0:
            w = out["weights"].clamp(1e-5, 1 - 1e-5)
            w2 = out["weights_sum"].clamp(1e-5, 1 - 1e-5)
            ent = lambda p: (-p * torch.log2(p) - (1 - p) * torch.log2(1 - p)).mean()
            loss = loss + opt.lambda_entropy * (ent(w) + ent(w2))
        if opt.lambda_specular > 0 and out["speculars"] is not None:
            loss = loss + opt.lambda_specular * (out["speculars"] ** 2).sum(-1).mean()
        if opt.sdf and opt.lambda_eikonal > 0:
            loss = loss + opt.lambda_eikonal * ((torch.linalg.norm(out["normal"], ord=2, dim=-1) - 1) ** 2).mean()

        M = out["num_points"]
        self.last_num_points = M
        self.samples_seen += M
        self.rays_seen += N
        if opt.adaptive_num_rays and M > 0 and not adapted:              # nerf/utils.py:796-797
            self.num_rays = max(1, int(round((opt.num_points / M) * self.num_rays)))

        # TV regulariser (nerf/utils.py:812-821 adds it to the unscaled gradients after backward).  Fast path: hand it to the
        # density encoder's backward, which folds it into its own table scatter pre-multiplied by the loss scale.
        tv_req = None
        if opt.lambda_tv > 0 and M > 0 and self.fused_tv and getattr(model, "_can_fuse", lambda: False)() \
                and model.max_level >= model.encoder.num_levels:
            if self.amp_adam:
                # gradients are summed over ranks and carry scale / world (see scale_loss): the TV term must carry the same factor
                scale_t = (self.optimizer.scale if self.world == 1 else self.optimizer.scale / self.world) if self.optimizer.amp else \
                    (None if self.world == 1 else self._one / self.world)
            else:
                scale_t = self.scaler.scale(self._one) if self.scaler.is_enabled() else None   # device scalar, no host sync
            # rows: only the backward call over THIS batch's M samples folds the term in (SDF evaluates the density encoder a second time
            # on the 6 M finite-difference offsets)
            tv_req = dict(weight=opt.lambda_tv, weight_outer=opt.lambda_tv * (10 if opt.bound > 1 else 1), inner01=0.5 / model.bound,
                          scale=scale_t, done=False, rows=M)
            model.encoder.tv_request = tv_req
        xyzs = out["xyzs"]
        if self.amp_adam:
            o = self.optimizer
            self._amp = {"density": dict(found_inf=o.found_inf, flagged=False), "color": dict(found_inf=o.found_inf, flagged=False, keep_half=True),
                         "mlp": dict(found_inf=o.found_inf, flagged=False, persistent_dw=len(self._mlp_params) == 7)}
            model.encoder.amp_request, model.encoder_color.amp_request, model.amp_request = self._amp["density"], self._amp["color"], self._amp["mlp"]
            if self.overlap_march and self.pipeline and self.sync is None and torch.device(self.device).type == "cuda" \
                    and self.global_step % opt.update_extra_interval != 0 and self._next is None:
                self._next = self._prepare_overlapped()
            o.backward(loss, self.world)
            model.encoder.amp_request = model.encoder_color.amp_request = model.amp_request = None
            model.encoder.tv_request = None
            if tv_req is None or not tv_req["done"]:
                self._tv(xyzs, 1.0 / self.world, scale_tensor=o.scale if o.amp else None)        # gradients are still scaled here
            flagged = []
            if self.sync is not None:
                # one SUM all-reduce per table gradient (the colour one stays fp16: half the bytes) + one small bucket; the reduced
                # gradients are then checked for inf/nan like any others (a sum of finite fp16 values can overflow)
                mlp = [self._amp["mlp"]["dw_flat"]] if "dw_flat" in self._amp["mlp"] else [p.grad for p in self._mlp_params]
                token = self.sync.all_reduce_sum_begin([model.encoder.embeddings.grad, self._amp["color"].get("grad_half"),
                                                        model.encoder_color.embeddings.grad], mlp + [o.found_inf])
                # the next batch's ray generation and march pass 1 read neither gradients nor parameters: enqueue them now so that they
                # run while the collectives are in flight (not on occupancy-refresh steps, which need the updated parameters first)
                if self.pipeline and self.global_step % opt.update_extra_interval != 0:
                    self._next = self._prepare()
                self.sync.all_reduce_sum_end(token)
            else:
                if self._amp["density"]["flagged"]:
                    flagged.append(model.encoder.embeddings)
                if self._amp["color"]["flagged"]:
                    flagged.append(model.encoder_color.embeddings)
                if self._amp["mlp"]["flagged"]:
                    flagged += self._mlp_params
            o.step(flagged=flagged)
        else:
            self.scaler.scale(loss).backward()
            model.encoder.tv_request = None
            tv_pending = tv_req is None or not tv_req["done"]
            if self.sync is None:
                self.scaler.unscale_(self.optimizer)                         # nerf/utils.py:812
                if tv_pending:
                    self._tv(xyzs, 1.0)
            else:
                # multi-GPU: every rank adds its own TV term (pre-multiplied by the loss scale), then the summed
                # gradients are averaged, then unscaled -- so all ranks see identical gradients and inf flags
                if tv_pending:
                    self._tv(xyzs, self.scaler.get_scale() if self.scaler.is_enabled() else 1.0)
                self.sync.all_reduce()
                self.scaler.unscale_(self.optimizer)
            self.scaler.step(self.optimizer)
            self.scaler.update()
        self.scheduler.step()
        if self.ema is not None and self.global_step % self.epoch_len == 0:      # end of an epoch (nerf/utils.py:1213-1214)
            self.ema.update()
        self._loss_pending.append(loss.detach())       # summed lazily (loss_acc): no per-step add kernel
        if len(self._loss_pending) >= 1024:
            _ = self.loss_acc
        if self.pipeline and self._next is None:
            # everything the next step needs before its sample count is known goes into the queue now, behind this step's
            # optimizer update (same order as the reference: refresh -> batch -> march), so the GPU never drains at the read-back
            self._next = self._prepare()
        return loss

    def _tv(self, xyzs, scale, scale_tensor=None):
        """Stand-alone TV pass (TV not folded into the backward: progressive levels, bound > 1 on the unfused path).  scale_tensor: the
        loss scale as a device scalar, multiplied in by the kernel (no host read-back)."""
        opt, model = self.opt, self.model
        if opt.lambda_tv <= 0 or xyzs is None or xyzs.shape[0] == 0:
            return
        lam = opt.lambda_tv * scale
        if opt.bound > 1:                                                # nerf/utils.py:815-821
            inner = xyzs.abs().amax(dim=-1) <= 1
            model.encoder.grad_total_variation(lam, xyzs[inner].contiguous(), model.bound, scale=scale_tensor)
            model.encoder.grad_total_variation(lam * 10, xyzs[~inner].contiguous(), model.bound, scale=scale_tensor)
        else:
            model.encoder.grad_total_variation(lam, xyzs, model.bound, scale=scale_tensor)

    @torch.no_grad()
    def averaged_parameters(self):
        """Context: the model carries the EMA weights (store / copy_to ... restore, nerf/utils.py:1250-1252,1340-1341); no-op without EMA."""
        import contextlib
        return self.ema.average_parameters() if self.ema is not None else contextlib.nullcontext()

    def eval_psnr(self, cam=0, downscale=4, use_ema=False):
        """PSNR of one rendered view against the analytic ground truth (white background).  use_ema: with the averaged weights, as the
        reference's evaluate_one_epoch renders (nerf/utils.py:1250-1252)."""
        if use_ema and getattr(self, "ema", None) is not None:
            with self.averaged_parameters():
                return type(self).eval_psnr(self, cam, downscale)
        self.model.eval()
        H = W = synthetic.LEGO_HW // downscale
        dev = self.device
        jj, ii = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
        pix = (jj * downscale * synthetic.LEGO_HW + ii * downscale).reshape(-1)
        rays_o, rays_d = synthetic.rays_from_pixels(self.poses, torch.full_like(pix, cam), pix)
        rgba = synthetic.render_gt(rays_o, rays_d, self.boxes)
        gt = rgba[:, :3] * rgba[:, 3:] + (1 - rgba[:, 3:])
        cnf = getattr(self, "cam_near_far", None)
        out = self.model.render(rays_o, rays_d, bg_color=1, perturb=False, shading="full", dt_gamma=self.opt.dt_gamma,
                                max_steps=self.opt.max_steps, T_thresh=1e-4, cam_near_far=None if cnf is None else cnf[cam:cam + 1])
        mse = F.mse_loss(out["image"], gt)
        return float(-10 * torch.log10(mse))


class UniformLaplacian:
    """Uniform-Laplacian smoothness of the mesh, `laplacian_smooth_loss(verts, faces)` of nerf/utils.py:176-221: with L = D - A over the
    UNIQUE directed edges (diagonal = number of distinct neighbours, -1 per neighbour), loss = mean_i || (L v)_i ||_2 =
    mean_i || deg_i v_i - sum_{j in N(i)} v_j || -- a norm, not a squared norm, and not divided by the degree (checked against the unchanged
    reference function in tests/test_stage1_reference.py; rounds 1-2 had the squared, degree-normalised form here).  The edge list
    depends on the topology only, so it is built once per mesh (the reference rebuilds its sparse matrix every step) and the loss is
    two index_add passes."""

    def __init__(self, faces, n_verts):
        f = faces.long()
        ii = torch.cat([f[:, 0], f[:, 1], f[:, 1], f[:, 2], f[:, 2], f[:, 0]])
        jj = torch.cat([f[:, 1], f[:, 0], f[:, 2], f[:, 1], f[:, 0], f[:, 2]])
        key = torch.unique(ii * n_verts + jj)
        self.ii, self.jj = key // n_verts, key % n_verts
        self.deg = torch.zeros(n_verts, device=faces.device).index_add_(0, self.ii, torch.ones_like(self.ii, dtype=torch.float32)).unsqueeze(1)

        # CSR of the (sorted) edge list for the HIP form of the loss (n2m_laplacian_*): one launch each way instead of two index_add
        # passes of 52 us and a dozen elementwise / reduce launches
        self.n_verts = int(n_verts)
        counts = torch.bincount(self.ii, minlength=n_verts)
        self.row_ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=faces.device), counts.cumsum(0)]).to(torch.int32).contiguous()
        self.col = self.jj.to(torch.int32).contiguous()

    def __call__(self, verts):
        if verts.is_cuda and verts.dtype == torch.float32:
            return _MeshRegularisers.apply(verts, None, self.row_ptr, self.col, 1.0, 0.0, 0.0, 0)
        nb = _NeighbourSum.apply(verts, self.ii, self.jj)
        return (verts * self.deg - nb).norm(dim=1).mean()

    def regularisers(self, verts, offsets, lam_lap, lam_off, n_in=None):
        """lam_lap * laplacian_smooth_loss(verts) + lam_off * the offset penalty of nerf/utils.py:772-789 (n_in: vertices of the inner mesh when
        bound > 1 -- the outer meshes' mean counts a tenth) as ONE value with one launch each way (n2m_laplacian_*)."""
        V = verts.shape[0]
        if n_in is None or n_in >= V or n_in <= 0:
            n_in, w_in, w_out = V, lam_off / V, 0.0
        else:
            w_in, w_out = lam_off / n_in, 0.1 * lam_off / (V - n_in)
        return _MeshRegularisers.apply(verts, offsets if lam_off > 0 else None, self.row_ptr, self.col, float(lam_lap), float(w_in), float(w_out), int(n_in))


class _MeshRegularisers(torch.autograd.Function):
    """lam_lap * mean_i || deg_i v_i - sum_{j in N(i)} v_j || [+ sum_i w_i |off_i|^2] through n2m_laplacian_forward / _backward (neighbour sums in
    the CSR's fixed order)."""

    @staticmethod
    def forward(ctx, verts, offsets, row_ptr, col, lam_lap, w_in, w_out, n_in):
        from . import _lib as L
        verts = verts.contiguous()
        offsets = offsets.contiguous() if offsets is not None else None
        V = verts.shape[0]
        Lv, norm = torch.empty_like(verts), torch.empty(V, dtype=torch.float32, device=verts.device)
        partial = torch.empty((V + 255) // 256, dtype=torch.float32, device=verts.device)
        L.call("n2m_laplacian_forward", L.ptr(verts), L.ptr(row_ptr), L.ptr(col), V, L.ptr(offsets), lam_lap, w_in, w_out, n_in, L.ptr(Lv), L.ptr(norm),
               L.ptr(partial), L.stream())
        ctx.save_for_backward(Lv, norm, row_ptr, col, offsets)
        ctx.args = (lam_lap, w_in, w_out, n_in)
        return partial.sum()

    @staticmethod
    def backward(ctx, g):
        from . import _lib as L
        Lv, norm, row_ptr, col, offsets = ctx.saved_tensors
        lam_lap, w_in, w_out, n_in = ctx.args
        g = g.float().contiguous()
        d = torch.empty_like(Lv)
        d_off = torch.empty_like(Lv) if offsets is not None else None
        L.call("n2m_laplacian_backward", L.ptr(Lv), L.ptr(norm), L.ptr(row_ptr), L.ptr(col), Lv.shape[0], L.ptr(g), lam_lap, L.ptr(offsets), w_in, w_out,
               n_in, L.ptr(d), L.ptr(d_off), L.stream())
        return d, d_off, None, None, None, None, None, None


class _NeighbourSum(torch.autograd.Function):
    """nb[i] = sum_{j in N(i)} v[j] over a SYMMETRIC directed edge list (i,j) and (j,i) both present: the adjoint is the same
    operator, so backward is another gather + index_add instead of autograd's sort-based index_put of the gather's backward
    (measured: 270 us + 170 us of sort per step on the 150 k-vertex mesh)."""

    @staticmethod
    def forward(ctx, verts, ii, jj):
        ctx.save_for_backward(ii, jj)
        return torch.zeros_like(verts).index_add_(0, ii, verts[jj])

    @staticmethod
    def backward(ctx, g):
        ii, jj = ctx.saved_tensors
        g = g.contiguous()
        return torch.zeros_like(g).index_add_(0, ii, g[jj]), None, None


def laplacian_smooth_loss(verts, faces):
    return UniformLaplacian(faces, verts.shape[0])(verts)


class Stage1Trainer:
    """Stage-1 iteration of the reference (nerf/utils.py:708-721,745-789; one full view per step, nerf/provider.py:298-306):
    rasterise at ssaa x resolution, shade covered pixels with the colour networks, antialias, downscale, MSE (+ mask,
    + Laplacian smoothness, + offset L2), Adam on colour networks + vertex offsets."""

    def __init__(self, model, opt, poses, vertices, triangles, device, H=synthetic.LEGO_HW, W=synthetic.LEGO_HW, rank=0, world_size=1, seed=0):
        self.model, self.opt, self.device = model.to(device), opt, device
        self.H, self.W = H, W
        self.poses = poses.to(device)
        self.views = list(range(rank, poses.shape[0], world_size))            # views shard across ranks
        self.mvps = torch.stack([synthetic.mvp_matrix(p, H, W) for p in self.poses])
        model.init_stage1(vertices, triangles)
        params = model.get_params(opt.lr) + [{"params": model.vertices_offsets, "lr": opt.lr_vert, "weight_decay": 0}]
        # main.py:221 Adam(eps=1e-15) + nerf/utils.py:506 GradScaler, as in stage 0: optim.FusedAdamAMP does both in two launches (torch:
        # unscale + inf check + four multi-tensor Adam launches, 0.35 ms of host time per step of a step that is host-bound) and refreshes
        # the colour table's fp16 working copy in the same pass
        # (multi-GPU, SURVEY 8e: views shard over the ranks, the gradients are SUMMED with 1 / world folded into the loss scale like stage 0 --
        #  round 3 dropped back to torch's Adam + GradScaler there)
        self.world, self.rank = world_size, rank
        self.amp_adam = torch.device(device).type == "cuda" and bool(opt.fp16)
        if self.amp_adam:
            from .optim import FusedAdamAMP
            self.optimizer = FusedAdamAMP(params, eps=1e-15, amp=True)
            encc = model.encoder_color
            if hasattr(encc, "half_table"):
                self.optimizer.shadows[encc.embeddings] = lambda: encc.half_table()
            # as in stage 0: the colour table's gradient stays fp16 (the binned backward writes it, Adam reads it: no fp32 copy), the weight
            # gradients live in one persistent buffer Adam clears, and the kernels that produce them raise found_inf themselves -- the
            # foreach inf check then only covers the vertex offsets (it was a 35 us pass over the 49 MB fp32 table gradient)
            self._amp = {}
            self.optimizer.half_grads[encc.embeddings] = lambda: self._amp.get("color", {}).get("grad_half")
            self._mlp_params = [p for m in (model.sigma_net, model.color_net, model.specular_net) for p in m.parameters()]
            self._amp_mlp = bool(getattr(opt, "fused_mlp", False)) and len(self._mlp_params) == 7
            if self._amp_mlp:
                for i, p in enumerate(self._mlp_params):
                    self.optimizer.ext_grads[p] = lambda i=i: (self._amp.get("mlp", {}).get("dw_views") or [None] * 7)[i]
        else:
            self.optimizer = torch.optim.Adam(params, eps=1e-15, fused=(torch.device(device).type == "cuda"))
        iters = opt.iters
        # main.py:239 applies the same schedule to both stages: 0.01 -> 1 over 500 iterations, then 0.1 ** ((it - 500) / (iters - 500))
        self.scheduler = LambdaLR(
            self.optimizer, lambda it: 0.01 + 0.99 * (it / 500) if it <= 500 else 0.1 ** ((it - 500) / (iters - 500)))
        self.scaler = torch.amp.GradScaler("cuda", enabled=bool(opt.fp16))
        self.sync = GradSync(model, world_size) if world_size > 1 else None
        self.boxes = synthetic.boxes(device)
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed + rank)
        self.global_step = 0
        jj, ii = torch.meshgrid(torch.arange(H, device=device), torch.arange(W, device=device), indexing="ij")
        self.pix = (jj * W + ii).reshape(-1)
        self.laplacian = UniformLaplacian(model.triangles, model.vertices.shape[0])
        self.view_cache = {}          # per view: rays + ground-truth RGBA, resident in HBM like the reference's --preload
        self._dirs = {}               # per view: unit directions at the ssaa resolution (30 MB per 800 x 800 view at ssaa 2)
        self.covered_seen = 0         # shaded (covered) full-resolution pixels so far: the unit of the stage-1 byte model
        self.fused_head = torch.device(device).type == "cuda" and int(opt.ssaa) in (1, 2)      # losses.stage1_head (False: the torch graph)
        self.packed_aa = os.environ.get("N2M_S1_PACKED_AA", "1") != "0"      # one antialias call on RGB + alpha instead of two

    @torch.no_grad()
    def sync_refine_state(self):
        """Views shard over the ranks, so every rank has accumulated the per-face errors of ITS views only (update_triangles_errors,
        nerf/renderer.py:924-943): sum both accumulators over the ranks before rank 0 refines the mesh (nerf/utils.py:1204-1207).  Collective."""
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.model.triangles_errors, op=dist.ReduceOp.SUM)
            dist.all_reduce(self.model.triangles_errors_cnt, op=dist.ReduceOp.SUM)

    @torch.no_grad()
    def broadcast_mesh(self, src=0):
        """After rank `src` has refined / decimated the mesh (refine_and_decimate, nerf/renderer.py:168-296: pymeshlab, outside this library) the
        other ranks take its vertices and faces, re-initialise stage 1 on them and rebuild what depends on the mesh (the optimizer over
        the new vertex offsets -- nerf/utils.py:1209-1211 does the same -- and the Laplacian).  Collective."""
        if self.world <= 1:
            return
        import torch.distributed as dist
        model, dev = self.model, self.device
        n = torch.tensor([model.vertices.shape[0], model.triangles.shape[0]], dtype=torch.int64, device=dev)
        dist.broadcast(n, src=src)
        nv, nf = int(n[0]), int(n[1])
        mine = self.rank == src
        v = model.vertices.detach().float().contiguous() if mine else torch.empty(nv, 3, dtype=torch.float32, device=dev)
        f = model.triangles.detach().to(torch.int32).contiguous() if mine else torch.empty(nf, 3, dtype=torch.int32, device=dev)
        dist.broadcast(v, src=src)
        dist.broadcast(f, src=src)
        step, covered, views = self.global_step, self.covered_seen, (self.view_cache, self._dirs)
        self.__init__(model, self.opt, self.poses, v, f, dev, self.H, self.W, self.rank, self.world)      # fresh offsets, accumulators, optimizer, schedule, Laplacian
        self.global_step, self.covered_seen = step, covered
        self.view_cache, self._dirs = views                  # rays / ground truth / directions do not depend on the mesh

    def _view(self, v):
        if v not in self.view_cache:
            rays_o, rays_d = synthetic.rays_from_pixels(self.poses, torch.full_like(self.pix, v), self.pix, self.H, self.W)
            self.view_cache[v] = (rays_o, rays_d, synthetic.render_gt(rays_o, rays_d, self.boxes))
        return self.view_cache[v]

    def preload(self):
        """Rays + ground truth of every view of this rank resident on the device before training (the reference's --preload)."""
        for v in self.views:
            rays_d = self._view(v)[1]
            if self.fused_head and v not in self._dirs:
                self._dirs[v] = self.model.stage1_dirs(rays_d, self.H, self.W).detach()

    def train_step(self):
        opt, model = self.opt, self.model
        if not model.training:
            model.train()
        v = self.views[self.global_step % len(self.views)]
        self.global_step += 1
        rays_o, rays_d, rgba = self._view(v)
        bg = torch.rand(self.H * self.W, 3, device=self.device, generator=self.gen)
        self.optimizer.zero_grad(set_to_none=True)
        shading = "diffuse" if opt.diffuse_only else "full"                  # nerf/utils.py:669-672 (diffuse_step only gates stage 0)
        verts = None
        if self.fused_head:
            # everything behind the two antialias calls (clamp, alpha * rgb, depth, T, ssaa reduction, background blend, per-pixel loss,
            # mean) and its backward in ONE launch (losses.stage1_head) instead of ~40 full-image elementwise / resize launches
            from .losses import stage1_head
            dirs = self._dirs.get(v)
            if dirs is None:              # unit directions at the rendered resolution: per view, resident like the rays they come from
                dirs = self._dirs[v] = model.stage1_dirs(rays_d, self.H, self.W).detach()
            verts = model.vertices + model.vertices_offsets            # once: the front half and the smoothness loss share it
            rast, aa_alpha, aa_rgb = model._stage1_front(rays_d, self.mvps[v], self.H, self.W, shading, dirs=dirs, packed=self.packed_aa, vertices=verts)
            te = (model.triangles_errors, model.triangles_errors_cnt) if opt.refine else (None, None)      # update_triangles_errors rides along
            # (seed: with FusedAdamAMP the total loss is differentiated with gradient = loss scale and this term enters it with weight 1)
            # (multi-GPU: the gradient that flows into the loss is scale / world -- FusedAdamAMP.backward(loss, world) -- and the head must be
            #  handed exactly that)
            seed = None
            if self.amp_adam:
                seed = self.optimizer.scale if self.world == 1 else self.optimizer.scale / self.world
            loss, _, _, _, trig, loss_px = stage1_head(aa_alpha, aa_rgb, rast, rgba, bg, self.H, self.W, int(opt.ssaa), opt.lambda_rgb,
                                                       max(opt.lambda_mask, 0.0), *te, seed=seed)
        else:
            gt_mask = rgba[:, 3:]
            gt_rgb = rgba[:, :3] * gt_mask + bg * (1 - gt_mask)
            out = model.render_stage1(rays_o, rays_d, self.mvps[v], self.H, self.W, bg_color=bg, shading=shading)
            loss = opt.lambda_rgb * F.mse_loss(out["image"], gt_rgb, reduction="none").mean(-1)
            if opt.lambda_mask > 0:
                loss = loss + opt.lambda_mask * F.mse_loss(out["weights_sum"].view(-1), gt_mask.view(-1), reduction="none")
            if opt.refine:
                model.update_triangles_errors(loss.detach())
            loss = loss.mean()
        self.covered_seen += getattr(model, "last_covered", 0)
        if verts is not None and verts.is_cuda and opt.lambda_lap > 0 and opt.lambda_offsets > 0:
            # both mesh regularisers (nerf/utils.py:761-789) as one value, one launch each way
            loss = loss + self.laplacian.regularisers(verts, model.vertices_offsets, opt.lambda_lap, opt.lambda_offsets,
                                                      int(model.v_cumsum[1]) if opt.bound > 1 else None)
            reg_done = True
        else:
            reg_done = False
        if opt.lambda_lap > 0 and not reg_done:
            loss = loss + opt.lambda_lap * self.laplacian(verts if verts is not None else model.vertices + model.vertices_offsets)
        if opt.lambda_offsets > 0 and not reg_done:                          # nerf/utils.py:772-789
            off = model.vertices_offsets
            if opt.bound > 1:       # inner mesh (cascade 0) + 0.1 x the outer cascades' meshes
                n_in = int(model.v_cumsum[1])
                loss_offsets = (off[:n_in] ** 2).sum(-1).mean() + 0.1 * (off[n_in:] ** 2).sum(-1).mean()
            else:
                loss_offsets = (off ** 2).sum(-1).mean()
            loss = loss + opt.lambda_offsets * loss_offsets
        if self.amp_adam:
            o = self.optimizer
            fused_field = bool(getattr(opt, "fused_mlp", False))
            self._amp = {"color": dict(found_inf=o.found_inf, flagged=False, keep_half=fused_field)}
            model.encoder_color.amp_request = self._amp["color"]
            if self._amp_mlp:
                self._amp["mlp"] = dict(found_inf=o.found_inf, flagged=False, persistent_dw=True)
                model.amp_request = self._amp["mlp"]
            # (the colour table's backward merges same-cell runs on all sixteen levels, like engine_stage1: consecutive covered pixels share cells)
            from . import _lib as L
            L.call("n2m_grid_backward_merge_levels", int(os.environ.get("N2M_S1_MERGE_LEVELS", "16")))
            try:
                o.backward(loss, self.world)
            finally:
                L.call("n2m_grid_backward_merge_levels", 0)
            model.encoder_color.amp_request = model.amp_request = None
            flagged = [model.encoder_color.embeddings] if self._amp["color"]["flagged"] else []
            if "mlp" in self._amp and self._amp["mlp"]["flagged"]:
                flagged += self._mlp_params
            if self.sync is not None:
                # one SUM all-reduce per large gradient in its own dtype (the colour table's stays fp16) + one small bucket with the MLP
                # weights' gradients and the inf flag; a sum of finite values can overflow, so the reduced gradients are checked again
                ce = model.encoder_color.embeddings
                mlp = [self._amp["mlp"]["dw_flat"]] if "dw_flat" in self._amp.get("mlp", {}) else [p.grad for p in self._mlp_params]
                self.sync.all_reduce_sum([self._amp["color"].get("grad_half"), ce.grad, model.vertices_offsets.grad], mlp + [o.found_inf])
                flagged = []
            o.step(flagged=flagged)
        else:
            self.scaler.scale(loss).backward()
            if self.sync is not None:
                self.sync.all_reduce()
            self.scaler.step(self.optimizer)
            self.scaler.update()
        self.scheduler.step()
        return loss
