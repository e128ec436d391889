"""Stage-0 step executor: the iteration of trainer.Stage0Trainer's fast path (lego recipe: fused field, packed tables, fused loss, TV
folded into the table backward, FusedAdamAMP) issued as a FIXED sequence of C-ABI launches on preallocated buffers.

Why it exists: with the kernels of this package a training step is ~0.7 ms of GPU work, and the same step driven through
torch.autograd (custom Functions, ~50 tensor allocations, AccumulateGrad, LambdaLR, optimizer hooks) costs the host ~0.85 ms -- the
step was HOST-bound (tools/cpu_bound.py).  The executor keeps PyTorch for what the task statement keeps it for -- device memory,
streams, the random generators and torch.distributed -- and does the rest itself:

  prepare (side stream, for batch i+2):  torch.rand(N, 6) -> n2m_batch_rays (pixels, rays, ground truth, near/far, jitter, background)
        -> n2m_march_rays_train_fused (one march: counts + recorded chunks, replay) -> count to pinned memory + event
  step (main stream, batch i):  n2m_grid_encode_forward_packed[_tvterms: one GPU, the lookup also leaves the backward's TV terms] ->
        n2m_field_forward -> n2m_composite_loss_train (compositing, loss head,
        both backward passes) -> n2m_field_backward -> n2m_grid_encode_backward_binned_pair (+TV) / ..._pair_tvt -> [world > 1: SUM all-reduce of the
        fixed gradient buffers, fine levels first] -> n2m_adam_step -> n2m_scaler_update_slots_loss

Same kernels, same arguments, same random draws in the same order as Stage0Trainer: the two produce the same parameters
(tests/test_engine.py).  What the reference does per iteration is cited there (nerf/utils.py:628-823,1152-1190, main.py:221-241).
Configurations outside the fast path (SDF, individual codes, unfused MLPs, bound > 1 without the packed tables) stay on Stage0Trainer.
"""
import ctypes
import os
import time

import numpy as np
import torch

from . import _lib as L
from . import raymarching, synthetic
from .fused import SHADING, _affine
from .gridencoder import _host_offsets, same_geometry
from .optim import FusedAdamAMP

_p = L.ptr


def _shard_refresh_default():
    from .parallel import shard_refresh_default
    return shard_refresh_default()


def lr_lambda(it, iters):
    """main.py:239: 0.01 -> 1 over the first 500 iterations, then 0.1 ** ((it - 500) / (iters - 500))."""
    return 0.01 + 0.99 * (it / 500) if it <= 500 else 0.1 ** ((it - 500) / (iters - 500))


def peer_chunk_rows(split, world, pad=0):
    """Rows of one rank's chunk of the coarse half [0, split) in peer-store mode: split / world rounded UP to a multiple of four (+ pad), so that
    every chunk starts 16-byte aligned in the packed table (8-byte rows, stored as 16-byte row pairs) and holds a multiple of four rows; the last
    rank's chunk is what is left.  None when the layout does not exist (split not a multiple of four, or the padding leaves the last rank empty)."""
    if split % 4 != 0 or pad % 4 != 0:
        return None
    cs = (split + 4 * world - 1) // (4 * world) * 4 + pad
    return cs if (world - 1) * cs < split else None


class _RayBufs:
    """Per-batch ray-side tensors.  Three sets rotate: while step i consumes batch i, batch i+1 waits for its turn and batch i+2 is
    produced on the side stream (behind the marker in front of Adam(i), i.e. after the last kernel that read batch i-1's set)."""

    def __init__(self, cap, dev):
        self.cap = cap
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.o, self.d, self.rgba = f(cap, 3), f(cap, 3), f(cap, 4)
        self.nears, self.fars, self.noises, self.bg_buf = f(cap), f(cap), f(cap), f(cap, 3)
        self.u = self.bg = None
        self.rays = torch.empty(cap, 2, dtype=torch.int32, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.ws = torch.empty(int(L.lib().n2m_march_fused_workspace_bytes(cap)), dtype=torch.uint8, device=dev)   # chunk records, group totals
        self.one_pass = False        # samples written by the single-pass marcher (complete once the count is)
        self.host_count = torch.empty(1, dtype=torch.int32, pin_memory=True)
        self.count_ready = torch.cuda.Event()
        self.written = torch.cuda.Event()
        self.spec = False
        self.index = 0
        self.M = None                # sample count once the host has read it
        self.loss = torch.empty(1, dtype=torch.float32, device=dev)
        self.samples = None          # [cap_m, 8] fp32: xyzs | dirs | ts (speculative write pass)
        self.cap_m = 0
        self.N = 0
        self.args = None
        self.refreshed = False


class Stage0Engine:
    def __init__(self, model, opt, poses, device, rank=0, world_size=1, seed=0, ema_decay=0.95):
        self.model, self.opt, self.device = model.to(device), opt, torch.device(device)
        dev = self.device
        assert dev.type == "cuda", "the step executor drives HIP kernels: no CPU path"
        if bool(opt.sdf) and world_size > 1:
            raise ValueError("Stage0Engine runs the SDF recipe on one rank; use trainer.Stage0Trainer for multi-rank SDF training")
        if not self.supported(model, opt):
            raise ValueError("Stage0Engine covers the fused recipes (fused_mlp, fp16, no individual codes, power-of-two "
                             "bound, shared encoder geometry); use trainer.Stage0Trainer for other configurations")
        L.lib()
        self.poses = poses.to(dev).float().contiguous()
        self.rank, self.world = rank, world_size
        self.global_step = 0
        self.num_rays = opt.num_rays
        # ONE draw per batch for everything random about it (synthetic.batch_from_uniforms): which batch a number goes to then does not
        # depend on how far ahead batches are prepared, so this executor and Stage0Trainer consume identical draws (rank r: its own rays)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed + rank)
        self.optimizer = FusedAdamAMP(model.get_params(opt.lr), eps=1e-15, amp=True)
        # EMA of the parameters (main.py:241: decay 0.95 for stage 0; nerf/utils.py:544-545), updated once per epoch = every len(loader) =
        # number-of-training-views steps (nerf/utils.py:1213-1214); evaluation runs on the averaged weights (averaged_parameters())
        self.ema = None
        self.epoch_len = max(1, int(poses.shape[0]))
        if ema_decay is not None:
            from .ema import ExponentialMovingAverage
            self.ema = ExponentialMovingAverage(model.parameters(), decay=ema_decay)
        self.images = None
        self.scene = getattr(opt, "scene", "lego")
        self.boxes = synthetic.boxes(dev, self.scene)
        # --enable_cam_near_far (main.py:40, config 4): per-view (near, far) from the sparse points, applied by the batch kernel
        self.cam_near_far = synthetic.cam_near_far(self.poses, self.scene) if getattr(opt, "enable_cam_near_far", False) else None
        self.samples_seen = self.rays_seen = 0
        self.last_num_points = 0
        self._loss_pending, self._loss_sum = [], torch.zeros(1, device=dev)
        self.sync = None
        if world_size > 1:
            from .parallel import GradSync
            self.sync = GradSync(model, world_size)
            # occupancy refresh: every rank queries 1 / W of the cells (Morton ranges) and the densities are all-gathered -- the bit field
            # stays the replicated path's, bit for bit (renderer.update_extra_state; A/B: N2M_SHARD_REFRESH=0)
            self.model.refresh_shard = (rank, world_size) if _shard_refresh_default() else None
        self.side = L.side_stream(dev, slot=2)
        # TV terms of the batch as their own kernel on a third stream beside the field kernels (n2m_grid_tv_terms + ..._pair_tvt): takes the
        # stencil's gathers out of the fill (backward 286 -> 250 us) -- but whatever kernel the terms' launch overlaps slows down by about its
        # own 55-95 us (field forward 31 -> 52, compositing 18 -> 50, field backward 59 -> 102: tools/sweep_tv.sh, profiles/r03_tv_split_sweep.txt),
        # so the step LOSES 14-27 us.  Kept as a measured alternative (N2M_TV_SPLIT=1, N2M_TV_AT, N2M_TV_PRIO); off by default.  Same bits.
        self.tv_stream = L.side_stream(dev, slot=3, priority=int(os.environ.get("N2M_TV_PRIO", "0")))
        self.tv_split = os.environ.get("N2M_TV_SPLIT", "0") == "1"
        self.tv_at = int(os.environ.get("N2M_TV_AT", "0"))      # where the terms' kernel may start: 0 behind the lookup, 1 behind the field forward, 2 behind compositing
        self.split_backward = True            # multi-rank: table backward in two level halves, the first half's all-reduce under the second
        self.overlap = True                   # next batch on the side stream (False: everything on the main stream, same results)
        self.single_pass = os.environ.get("N2M_MARCH_PASSES", "1") != "2"      # one-launch marcher (A/B: N2M_MARCH_PASSES=2)
        # side-stream go-ahead: 0 before Adam (default), 1 before the table backward, 2 before the field backward, 3 between the table backward's
        # fill and its accumulate (n2m_grid_backward_mid_event).  Measured (round 4, 200 steps): 3 takes the marcher off the lookup (71.4 ->
        # 67.8 us) and off Adam (99 -> 91) but doubles the accumulate beside it (backward 235 -> 301 us): 0.569 -> 0.613 ms/step.
        self.marker_at = int(os.environ.get("N2M_MARKER_AT", "0"))
        # [round 6, MEASURED AND REJECTED: off by default, N2M_TV_CORNERS=1 turns it on]  The forward lookup leaves, per (hashed level, sample), the
        # density values of corners 000 / 100 / 010 / 001 of the sample's cell (n2m_grid_encode_forward_packed_tv): centre and +x / +y / +z
        # neighbours of the TV stencil the table backward's fill evaluates for the same sample a few kernels later on the same table -- which
        # then gathers three neighbours instead of six.  Why it was built: the fill's fine levels are bound by their XCD's L2 request rate, that
        # stencil is six of their ~8 scattered requests per (sample, level), and an ABLATION build (three gathers, no records: wrong results)
        # ran the backward 20.8 us faster (208.8 -> 188.0 us, step 0.541 -> 0.516 ms).  What the real thing does (profiles/r06_tv_corners.txt,
        # ABBA): backward 210.9 -> 206.8 us, lookup 71.3 -> 77.1 us (46 MB more stores), step 0.5394 -> 0.5431 ms -- the 16-byte record comes
        # from HBM where the three gathers it replaces hit the L2, and requested a tile ahead like the other inputs it costs the fill's 124-VGPR
        # kernel its last registers (122 + 32 bytes of scratch).  Same bits either way (tests/test_tv_corners.py).
        self.tv_corners = (os.environ.get("N2M_TV_CORNERS", "0") == "1" and world_size == 1 and not opt.sdf and opt.lambda_tv > 0
                           and not self.tv_split)
        # [round 6; the default on one GPU, A/B: N2M_TV_FWD=0]  TV terms from the forward lookup: n2m_grid_encode_forward_packed_tvterms leaves the FINISHED term of every
        # (sample, level) -- the stencil's centre and +x / +y / +z values are corners the lookup holds in registers, at most three more rows are
        # gathered -- and the table backward consumes them through n2m_grid_encode_backward_binned_pair_tvt: its fill reads 4 coalesced bytes per
        # (sample, level) instead of gathering six rows inside its tile's dependent chain (what the corner records above could not deliver: they
        # still left three gathers and a 16-byte record in that chain).  Same bits (tests/test_tv_fwd.py).  Measured (profiles/r06_tv_fwd.txt):
        # backward 208 -> 180 us, lookup 72 -> 94 us; lego step -0.7 % over 40-step windows, -1.4 % over 192 steps, -2.0 % in the diffuse phase; garden +-0.
        self.tv_fwd = (os.environ.get("N2M_TV_FWD", "1") != "0" and world_size == 1 and not opt.sdf and opt.lambda_tv > 0
                       and not self.tv_split and not self.tv_corners)
        # [round 6, MEASURED AND NOT ADOPTED: off by default, N2M_ADAM_TAIL=1 turns it on]  The scaler / step-count / loss-value bookkeeping behind the
        # optimizer pass as the TAIL of that pass (n2m_adam_step_scaler: one wave of its last workgroup runs the code of n2m_scaler_update_slots_loss3
        # once every other wave has left an arrival mark) instead of a one-workgroup launch of its own on the step's critical path (~10 us + the queue gap
        # in front of the next lookup).  Identical bits (tests/test_optim.py, tests/test_adam_tail.py) -- but the optimizer pass takes 136 us instead of 93
        # in the kernel that also holds the bookkeeping code, whatever the arrival scheme: step 0.540 -> 0.563 ms.
        self.adam_tail = os.environ.get("N2M_ADAM_TAIL", "0") == "1" and world_size == 1
        self._tail_ticket = torch.zeros(64 * 32, dtype=torch.int32, device=dev)       # N2M_TAIL_TICKET_WORDS
        self._mid_events = None
        # Live-first sample order for the table backward (round 5; MEASURED AND REJECTED, off by default -- N2M_LIVE_FIRST=1 turns it on).
        # The compositing kernel leaves per ray how many samples precede its early stop -- the others, 48 % of a trained lego batch
        # (profiles/r05_fill_stats.txt), receive exactly zero gradients (raymarching.cu:553,640) and deliver at most a TV term;
        # n2m_sample_order_live_first turns that into a permutation and the fill visits the samples in that order, so that the dead tails fill
        # whole waves, which take a TV-only path (one entry, one value through the run merge, ~30 % of the VALU work).  Same sums (fixed point).
        # Measured (profiles/r05_live_first_ab.txt, table backward per step, kernel with the additions): off 211.1 us | identity order 212.7 | live-first 212.7 | live-first
        # with the TV-only path switched off 218.7: the path saves 6 us, the order costs 7.6 (a ray's live prefix is 5.5 samples: 22-66 byte
        # pieces per gathered input instead of whole lines) + 4 us of order kernel and longer compositing: the fill is bound by its waits
        # (SQ: waiting 0.51 of wave cycles), not by the instructions the dead half of the batch issues.
        self.live_first = os.environ.get("N2M_LIVE_FIRST", "0") not in ("0", "")
        self._identity_order = os.environ.get("N2M_LIVE_FIRST", "0") == "2"      # (measurement: the order's indirection alone)
        # [experiment, N2M_LOOKUP_OVERLAP=1, single GPU] Adam in two calls by level half -- fine rows first -- and the NEXT step's lookup of the fine
        # levels on a stream of its own behind the first call: the lookup (L2-request bound) beside the optimizer pass of the coarse rows + the MLP
        # weights (HBM-stream bound).  Same bits (Adam is element-wise, the lookup's levels write disjoint rows).  Measured: see DESIGN section 7.
        self.lookup_overlap = os.environ.get("N2M_LOOKUP_OVERLAP", "0") == "1" and world_size == 1 and not opt.sdf
        self._fine_ready = None
        if self.lookup_overlap:
            self.s_lookup = L.side_stream(dev, slot=4)
        if os.environ.get("N2M_FILL_DBG"):                                        # (measurement switches of the fill, wrong results for most)
            L.call("n2m_debug_fill_times", int(os.environ["N2M_FILL_DBG"]), None)

        e1, e2 = model.encoder, model.encoder_color
        self.rows = e1.embeddings.shape[0]
        self.Lv = e1.num_levels
        self.S = float(np.log2(e1.per_level_scale))
        self.H0 = int(e1.base_resolution)
        self.ho = _host_offsets(e1)
        self.aff = _affine(float(model.bound))
        self.mlp_params = [p for m in (model.sigma_net, model.color_net, model.specular_net) for p in m.parameters()]
        assert len(self.mlp_params) == 7
        # persistent gradient buffers: both table gradients are fully rewritten by every backward (overwrite mode); the seven dW live
        # in one flat buffer that is all-zero between steps (the field backward adds into it, the Adam kernel clears it)
        self.g1 = torch.empty(self.rows, 1, dtype=torch.float32, device=dev)
        self.g2 = torch.empty(self.rows, 2, dtype=torch.float16, device=dev)
        self.dw = torch.zeros(sum(p.numel() for p in self.mlp_params), dtype=torch.float32, device=dev)
        self.dw_views, o = [], 0
        for p in self.mlp_params:
            self.dw_views.append(self.dw[o:o + p.numel()])
            o += p.numel()
        self._desc = {}
        self._bufs = [None, None, None]
        self._marker = None
        self._last = None                     # the newest prepared batch
        self._queue = []                      # prepared batches, oldest first (steady state: the next one and the one after)
        self._prepared = 0                    # index (1-based) of the newest prepared batch
        self.depth = 2                        # batches prepared ahead of the running step
        self._cur = 0

        self._work_cap = (0, 0)
        self._n_spec = int(L.lib().n2m_field_spec_partials())
        self._ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        self._seed = None
        self._aabb = None
        # ---- multi-GPU: optimizer sharded over the ranks (ZeRO-1 for the two tables).  Gradient rows are reduce-scattered (fine levels
        # first, under the coarse half of the backward), each rank runs Adam on 1/W of the rows and the packed 8-byte rows the forward
        # reads are all-gathered: fewer wire bytes than the all-reduce (73.5 + 49 MB against 2 x 73.5) and 1/W of the 540 MB Adam pass.
        # Only the packed table stays complete on every rank: the TV stencil reads its density column (n2m_grid_backward_config);
        # the fp32 parameter tensors are current inside the own shard only until sync_parameters() gathers them (checkpoint, export).
        self.shard = False
        if world_size > 1:
            import torch.distributed as dist
            W, split = world_size, int(self.ho[8]) if self.Lv == 16 else 0
            fine = self.rows - split
            # (a slice that starts at an odd row -- the coarse half at 8 ranks -- leaves the packed rows 8-byte aligned only: n2m_adam_step then
            # writes the two columns separately instead of whole 16-byte row pairs; tests/test_optim.py covers that form)
            # peer-store mode: coarse chunks padded to a multiple of FOUR rows (the last rank's is shorter), so that every chunk starts 16-byte aligned
            # in the packed table and n2m_adam_step_peer's fused form covers W = 4 and W = 8 too (split / W = 481 390 / 240 695 rows there: not
            # multiples of four, the second one odd); the collective path needs equal chunks (reduce_scatter_tensor) and keeps split / W.
            # (N2M_PEER_PAD_ROWS: extra padding, a multiple of four -- exercises the uneven layout with two ranks, tests/test_parallel_gpu.py)
            peer_mode = os.environ.get("N2M_PEER_STORE", "0") == "1" and not opt.sdf
            cs = split // W
            self._uneven = False
            padded = peer_chunk_rows(split, W, int(os.environ.get("N2M_PEER_PAD_ROWS", "0"))) if peer_mode else None
            if padded is not None:
                cs, even_ok = padded, True
                self._uneven = cs * W != split
            else:
                even_ok = split % W == 0
            ok = self.Lv == 16 and even_ok and fine % W == 0 and os.environ.get("N2M_SHARD_ADAM", "1") != "0"
            if ok:
                self.shard = True
                self._split, self._Cs, self._Fs = split, cs, fine // W
                f32 = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
                f16 = lambda *sh: torch.empty(*sh, dtype=torch.float16, device=dev)
                self.g1s = {"c": f32(self._Cs, 1), "f": f32(self._Fs, 1)}
                self.g2s = {"c": f16(self._Cs, 2), "f": f16(self._Fs, 2)}
                self._inplace_gather = dist.get_backend() == "nccl"      # RCCL gathers in place; gloo (tests) gets a copy of the shard
                self.chunked_gather = os.environ.get("N2M_CHUNKED_GATHER", "1") != "0"      # A/B: wait for both chunks right behind Adam
        # per-thread settings of the binned backward, stated before every backward of this engine (train_step): sharded -> the TV stencil
        # reads the density column of the packed table (row stride 2); W ranks -> gradients are SUMMED, so a local fp16 row above max / W
        # raises found_inf already
        self._bwd_cfg = (2 if self.shard else 1, float(world_size))
        # [opt-in, N2M_PEER_STORE=1] the sharded exchange without a collective in the data path (parallel.PeerExchange, include/n2m_peer.h): the
        # table backward's flush stores gradient rows into their owner's staging slots, the owner sums the W slots in rank order, Adam's
        # refreshed packed rows are stored into every rank's packed table; epoch flags instead of reduce-scatter / all-gather.  Tested between
        # two processes on one GPU, never run over xGMI: off by default.
        self.peer = None
        if self.shard and os.environ.get("N2M_PEER_STORE", "0") == "1":
            if opt.sdf:
                raise ValueError("N2M_PEER_STORE covers the NeRF recipes (the SDF head's folded backward is not routed)")
            from .parallel import PeerExchange
            self.peer = PeerExchange(rank, world_size, self.rows, self._split, self._Cs, self._Fs, dev, small_n=self.dw.numel() + 1)
            self._peer_small = torch.empty(self.dw.numel() + 1, dtype=torch.float32, device=dev)
            pk = model.packed_tables()
            self.peer.packed.copy_(pk)
            model._packed = self.peer.packed          # same values, exported memory; _packed_key stays valid
            model._packed_buffer = self.peer.packed   # ... and every later rebuild of the copy (load_state_dict, an edited table) lands in it too
            self._peer_route = self.peer.route()
            # n2m_adam_step_peer: the slot sum inside Adam's gradient load, the row push inside its packed-row store (two passes over the rows and
            # four launches fewer per step); covers even splits (rows per rank a multiple of 2, 16-byte aligned slices), else the separate passes
            self._peer_fused = (os.environ.get("N2M_PEER_FUSED", "1") != "0" and self._Cs % 4 == 0 and self._Fs % 4 == 0 and self._split % 2 == 0)
            self._adam_peer = None
        if self.shard:
            self.optimizer.shard_sync = lambda: self.sync_parameters(moments=True)      # state_dict() of a sharded run: gather first
        # ---- single GPU, measured alternative (N2M_FUSE_ADAM=1, off by default): the optimizer pass of the hashed levels (94 % of the rows)
        # inside the table backward's accumulate kernels (n2m_grid_encode_backward_binned_pair_adam): parameter and moments of both tables
        # exist twice, the flush reads one set and writes the other, and the two sets swap roles after every step -- model.encoder.embeddings
        # .data / optimizer.state always name the current one, so nothing outside the step sees the double buffering.  The TV stencil reads
        # the density column of the packed table.  Same bits as the separate pass (tests/test_engine.py), but the step LOSES 50-85 us: the
        # accumulates are latency-bound work items (2 per CU, barriers between their phases) and move the optimizer's 0.3 GB at a third of the
        # rate the streaming n2m_adam_step reaches (backward 287 -> 399-434 us against Adam 93 -> 15 us; lookup 74 -> 88 us), DESIGN section 7.
        self.sdf_fold = os.environ.get("N2M_SDF_FOLD", "1") != "0"      # SDF recipe: finite-difference copies folded into the batch's table backward
        self.sdf_tv_all = os.environ.get("N2M_SDF_TV_ALL", "1") != "0"  # SDF recipe, progressive phase: TV of all levels inside the batch's backward
        self.fuse_adam = None
        if world_size == 1 and not opt.sdf and self.Lv == 16 and os.environ.get("N2M_FUSE_ADAM", "0") == "1":
            fl, fr = ctypes.c_uint32(0), ctypes.c_uint32(0)
            self._fuse_cap = 1 << 19               # samples per batch the plan below holds for (larger batches take the unfused pass)
            L.call("n2m_grid_pair_fuse_plan", self._fuse_cap, self.Lv, self.ho.ctypes.data, ctypes.byref(fl), ctypes.byref(fr))
            if fl.value < self.Lv:
                self.fuse_adam = {"first_level": int(fl.value), "first_row": int(fr.value), "alt": None, "desc": {}}
                self._bwd_cfg = (2, 1.0)

    # ------------------------------------------------------------------------------------------------ configuration
    @staticmethod
    def supported(model, opt):
        e1, e2 = model.encoder, model.encoder_color
        sdf = bool(opt.sdf)      # SDF recipe (config 5): NeuS alpha, finite-difference normals, eikonal loss, progressive levels -- _step_sdf
        return (bool(getattr(opt, "fused_mlp", False)) and bool(opt.fp16) and getattr(opt, "ind_dim", 0) == 0
                and _affine(float(model.bound)) is not None and same_geometry(e1, e2) and e1.embeddings.shape[1] == 1
                and e2.embeddings.shape[1] == 2 and (sdf or not getattr(opt, "progressive_level", False))
                and opt.patch_size == 1 and (sdf or model.max_level >= e1.num_levels) and not (sdf and opt.lambda_entropy > 0))

    @property
    def loss_acc(self):
        return self._loss_sum

    def mark_untrained(self):
        if self.opt.mark_untrained:
            f = synthetic.LEGO_FOCAL
            self.model.mark_untrained_grid(self.poses, (f, f, synthetic.LEGO_HW / 2, synthetic.LEGO_HW / 2), cam_near_far=self.cam_near_far)

    # ------------------------------------------------------------------------------------------------------ buffers
    def _ray_bufs(self, N):
        self._cur = (self._cur + 1) % len(self._bufs)
        b = self._bufs[self._cur]
        if b is None or b.cap < N:
            b = self._bufs[self._cur] = _RayBufs(max(int(N * 1.5), 8192), self.device)
        return b

    def _sample_bufs(self, b, cap_m):
        if b.samples is None or b.cap_m < cap_m:
            b.cap_m = int(cap_m)
            b.samples = torch.empty(b.cap_m * 8, dtype=torch.float32, device=self.device)
        c = b.cap_m
        s = b.samples
        return s[:3 * c], s[3 * c:6 * c], s[6 * c:]

    def _work(self, M, N):
        """Step-local buffers, grow-only (level-major feature layouts are [16, M] with the step's own M as the stride)."""
        cm, cn = self._work_cap
        if M > cm or N > cn:
            cm, cn = max(cm, int(M * 1.25) + 1024), max(cn, int(N * 1.5) + 1024)
            dev = self.device
            f = lambda n: torch.empty(n, dtype=torch.float32, device=dev)
            w = self._w = {}
            w["h1"], w["d_h1"] = f(16 * cm), f(16 * cm)
            w["h2"] = torch.empty(32 * cm, dtype=torch.float16, device=dev)
            w["d_h2"] = torch.empty(32 * cm, dtype=torch.float16, device=dev)
            w["sigma"], w["rgb"], w["weights"], w["d_sr"] = f(cm), f(3 * cm), f(cm), f(4 * cm)
            w["tv"] = f(16 * cm)
            # the forward lookup's corner records for the table backward's TV stencil ([16, M, 4] fp32; hashed levels written): see tv_corners below
            w["tv4"] = f(64 * cm) if self.tv_corners else None
            w["spec_partial"] = torch.zeros(self._n_spec, dtype=torch.float32, device=dev)
            w["ws"], w["depth"], w["image"], w["d_image"], w["d_ws"], w["bg"] = f(cn), f(cn), f(3 * cn), f(3 * cn), f(cn), f(3 * cn)
            w["partial"] = f((cn + 3) // 4 + 1)
            w["live"] = torch.zeros(cn, dtype=torch.int32, device=dev)
            w["block_live"] = torch.zeros((cn + 15) // 16 + 1, dtype=torch.int32, device=dev)
            w["perm"] = torch.empty(cm, dtype=torch.int32, device=dev)
            w["zeros"] = torch.zeros(max(cm, 3 * cn), dtype=torch.float32, device=dev)
            self._work_cap = (cm, cn)
        return self._w

    # ------------------------------------------------------------------------------------------------- next batch
    def _refresh(self):
        """Occupancy refresh (every 16th step, nerf/utils.py:1155-1156): reads the parameters, so it runs on the main stream behind the
        optimizer update of the step before."""
        if self.sync is not None:
            self.sync.sync_rng_for_grid_update(self.global_step)
        if self.peer is not None:
            self.peer.check()                            # (the refresh drains the queue anyway: a blocking look at the error word)
        if self.shard:
            self.sync_parameters(density_only=True)      # the refresh evaluates the density from the fp32 table: gather the other ranks' rows
        self.model.update_extra_state()

    def _prepare(self, N):
        """Batch of N rays: pixel choice, rays + ground truth, near/far, march pass 1 (count + offset scan), count on its way to the host.
        Reads the cameras, the images and the occupancy bit field only."""
        opt, model, dev = self.opt, self.model, self.device
        if self.images is None:
            self.images = synthetic.preload_images(self.poses, self.boxes)
        b = self._ray_bufs(N)
        b.N, b.M = N, None
        s = L.stream()
        # one draw for everything random about the batch + one kernel for pixels, rays, ground truth, near/far, jitter, background
        b.u = torch.rand(N, 6, device=dev, generator=self.gen)
        self._aabb = model.aabb_train
        b.bg = b.bg_buf if opt.background != "white" else None
        synthetic.batch_from_uniforms(self.poses, self.images, b.u, self._aabb, model.min_near,
                                      out=(b.o, b.d, b.rgba, b.nears, b.fars, b.noises, b.bg), counter=b.counter, cam_near_far=self.cam_near_far)
        bits = model.density_bitfield
        b.args = (_p(b.o), _p(b.d), _p(bits), float(model.real_bound), int(bool(opt.contract)), float(opt.dt_gamma), int(opt.max_steps), N,
                  int(model.cascade), int(model.grid_size), _p(b.nears), _p(b.fars))
        b.bits = bits
        b.spec = b.one_pass = False
        expect = 0 if self.last_num_points <= 0 else ((int(1.25 * max(self.last_num_points, 1024)) + 1023) // 1024) * 1024
        if expect > 0 and self.single_pass:
            # march ONCE: count + recorded chunks, then replayed into buffers a quarter above the last batch; a ray that does
            # not fit is skipped like raymarching.cu:417 and _finish() then writes the batch exactly from the (offset, count) it left
            x, d, t = self._sample_bufs(b, expect)
            L.call("n2m_march_rays_train_fused", *b.args, _p(x), _p(d), _p(t), _p(b.rays), _p(b.counter), _p(b.noises), b.cap_m,
                   _p(b.ws), b.ws.numel(), s)
            b.host_count.copy_(b.counter, non_blocking=True)
            b.count_ready.record()
            b.spec = b.one_pass = True
            return b
        L.call("n2m_march_rays_train", *b.args, None, None, None, _p(b.rays), _p(b.counter), _p(b.noises), s)
        b.host_count.copy_(b.counter, non_blocking=True)
        b.count_ready.record()
        # speculative pass 2 right behind it, into buffers a quarter above the last batch (adaptive num_rays steers M towards
        # opt.num_points): a ray that does not fit is skipped like raymarching.cu:417 and _finish() re-marches the batch exactly.  With
        # batches prepared two ahead the pass has finished a whole step before its samples are read, so the consumer normally needs no
        # cross-stream wait at all (an event that has already fired is not waited for)
        if expect > 0:
            x, d, t = self._sample_bufs(b, expect)
            L.call("n2m_march_rays_train_write", *b.args, _p(x), _p(d), _p(t), _p(b.rays), _p(b.noises), b.cap_m, s)
            b.spec = True
            b.written.record()
        return b

    def _count(self, b):
        """Sample count of batch b on the host (waits for the event behind its offset scan; normally long complete)."""
        if b.M is None:
            # poll before blocking: hipEventSynchronize parks the thread, and waking it costs tens of microseconds the GPU then idles
            # (after an occupancy refresh the step cannot be enqueued before this count is known)
            t0 = time.perf_counter()
            while not b.count_ready.query():
                if time.perf_counter() - t0 > 5e-3:
                    b.count_ready.synchronize()
                    break
            b.M = int(b.host_count[0])
        return b.M

    def _fill_pipeline(self, max_new=None):
        """Prepare batches until `depth` of them wait in the queue (at most max_new of them in this call).  Batch j takes its ray count from batch j-1's sample count
        (adaptive num_rays, nerf/utils.py:796-797) and, when (j-1) % 16 == 0, follows an occupancy refresh that needs step j-1's
        parameter update: such a batch cannot be prepared early -- it is issued on the main stream behind that step.  Every other batch
        is issued on the side stream behind the marker in front of the running step's Adam kernel: its count pass then runs beside
        the optimizer update (a streaming kernel that leaves the ALUs idle) instead of beside the forward lookup, and its count is on
        the host a whole step before it is needed."""
        opt = self.opt
        new = 0
        while len(self._queue) < self.depth and (max_new is None or new < max_new):
            new += 1
            j = self._prepared + 1
            need_refresh = (j - 1) % opt.update_extra_interval == 0
            if need_refresh and j - 1 > self.global_step:
                break                                   # its refresh waits for a step that is not queued yet
            prev = self._last                           # batch j-1 (still queued, or the one the running step consumes)
            if prev is not None:                        # N(j) from M(j-1)
                M = self._count(prev)
                if opt.adaptive_num_rays and M > 0:
                    self.num_rays = max(1, int(round((opt.num_points / M) * prev.N)))
            N = int(self.num_rays)
            defer = False
            if need_refresh or not self.overlap or self._marker is None:
                if need_refresh:
                    self._refresh()
                b = self._prepare(N)
                if need_refresh and self.overlap and self._marker is not None:
                    # the batch behind this one needs THIS batch's count, which the host can only wait for -- and the running step cannot be
                    # enqueued before it has that count either.  So the next batch is prepared right behind the step's first launch (the
                    # lookup: _mid_step_fill), on the side stream behind this event instead of the step's marker: it marches beside the
                    # lookup and the field kernels as before, but the GPU no longer idles ~100 us while the host prepares it
                    self._post_refresh = torch.cuda.Event()
                    self._post_refresh.record()
                    self._mid_fill = True
                    defer = True
            else:
                after_refresh = (j - 2) % opt.update_extra_interval == 0 and getattr(self, "_post_refresh", None) is not None
                self.side.wait_event(self._post_refresh if after_refresh else self._marker)
                with torch.cuda.stream(self.side):
                    b = self._prepare(N)
                main = torch.cuda.current_stream(self.device)
                b.u.record_stream(main)
            b.index = j
            self._prepared = j
            self._last = b
            self._queue.append(b)
            if defer:
                break

    def _mid_step_fill(self):
        """Behind the step's first launch: the one batch whose preparation the refresh in front of this step had to put off."""
        if getattr(self, "_mid_fill", False):
            self._mid_fill = False
            self._fill_pipeline(max_new=1)

    def _finish(self, b):
        """Samples of batch b: the speculative pass 2 of _prepare() when the count fits its buffers (the normal case), else an exact
        pass 2 on the main stream (the host has waited for the event behind the offset scan, so no cross-stream dependency is needed)."""
        M = self._count(b)
        if M > 0 and b.spec and M <= b.cap_m:
            if not b.one_pass and not b.written.query():      # (single pass: the host has seen the count, which the same kernel wrote last)
                torch.cuda.current_stream(self.device).wait_event(b.written)
        elif M > 0:
            x, d, t = self._sample_bufs(b, ((int(1.25 * M) + 1023) // 1024) * 1024 if b.cap_m < M else b.cap_m)
            L.call("n2m_march_rays_train_write", *b.args, _p(x), _p(d), _p(t), _p(b.rays), _p(b.noises), b.cap_m, L.stream())
        return M

    # ------------------------------------------------------------------------------------------------------- Adam
    def _adam_desc(self, full, dense_only=False):
        """The N2mAdamDesc of this model (pointers are fixed for the life of the engine); `full`: the specular head takes part;
        `dense_only`: the two tables take part with their rows below fuse_adam['first_row'] only (the others got their update inside the
        table backward)."""
        o, model = self.optimizer, self.model
        # packed_tables() hands out a NEW tensor whenever a table was changed through torch (load_state_dict, an in-place edit): the
        # descriptor carries its address (Adam refreshes the copy the forward gathers from), so the address is part of the key
        pk = model.packed_tables()
        assert pk is not None
        if self.peer is not None and pk.data_ptr() != self.peer.packed.data_ptr():
            raise RuntimeError("peer-store exchange: the packed table left the exported buffer (peers would keep writing into the old one)")
        key = (full, getattr(o, "state_epoch", 0), pk.data_ptr(), model.encoder.embeddings.data_ptr(), model.encoder_color.embeddings.data_ptr(),
               dense_only)
        d = self._desc.get(key)
        if d is not None:
            return d
        live = lambda k: k[1:3] == key[1:3] and (k[3:5] == key[3:5] or (self.fuse_adam is not None and self._is_alt(k[3], k[4])))
        self._desc = {k: v for k, v in self._desc.items() if live(k)}      # drop descriptors of an older table / state generation
        self._packed = pk
        desc = L.AdamDesc()
        params = [p for g in o.param_groups for p in g["params"]]
        if self.shard:
            return self._adam_desc_sharded(full, key, pk, params)
        grads = {model.encoder.embeddings: (self.g1, 0, 0, (pk, 2)), model.encoder_color.embeddings: (self.g2, 1, 0, (pk, 3))}
        for i, p in enumerate(self.mlp_params):
            grads[p] = (self.dw_views[i], 0, 1, None)
        live = set(params[:2]) | set(self.mlp_params[:5]) | (set(self.mlp_params[5:]) if full else set())
        if self.opt.sdf:          # the NeuS variance (lr x 0.1, nerf/network.py:186): its gradient is a scalar the SDF head's backward leaves in d_var
            grads[model.variance] = (self._sdf_buf()["d_var"], 0, 0, None)
            live.add(model.variance)
        k, participants, groups = 0, 0, []
        group_of = {p: gi for gi, g in enumerate(o.param_groups) for p in g["params"]}
        for p in params:
            if p not in live:
                continue
            groups.append(group_of[p])
            g, is_half, clear, sh = grads[p]
            st = o.state[p]
            desc.param[k], desc.grad[k] = p.data_ptr(), g.data_ptr()
            desc.exp_avg[k], desc.exp_avg_sq[k] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            desc.half_shadow[k] = sh[0].data_ptr() if sh is not None else None
            desc.shadow_mode[k] = sh[1] if sh is not None else 0
            desc.numel[k], desc.grad_is_half[k], desc.clear_grad[k] = p.numel(), is_half, clear
            if dense_only and sh is not None:
                desc.numel[k] = self.fuse_adam["first_row"] * p.shape[1]
            desc.slot[k] = o._slot[p]
            participants |= 1 << (o._slot[p] - 1)
            k += 1
        desc.count = k
        d = self._desc[key] = (desc, participants, groups)
        return d

    def _adam_desc_halves(self, full, lr_factor):
        """(fine, coarse) descriptors of the single-GPU optimizer pass in two calls: rows [ho[8], rows) of both tables | rows [0, ho[8]) of both
        tables + the MLP weights.  Pointers are offsets into the same tensors n2m_adam_step takes in one call (_adam_desc)."""
        o, model = self.optimizer, self.model
        pk = model.packed_tables()
        key = ("halves", full, getattr(o, "state_epoch", 0), pk.data_ptr(), model.encoder.embeddings.data_ptr(), model.encoder_color.embeddings.data_ptr())
        cached = self._desc.get(key)
        if cached is None:
            self._packed = pk
            split = int(self.ho[8])
            group_of = {p: gi for gi, g in enumerate(o.param_groups) for p in g["params"]}
            e1p, e2p = model.encoder.embeddings, model.encoder_color.embeddings
            out = []
            for row0, n, with_mlp in ((split, self.rows - split, False), (0, split, True)):
                d, k, groups = L.AdamDesc(), 0, []
                for p, C, g, is_half, mode, gb in ((e1p, 1, self.g1, 0, 2, 4), (e2p, 2, self.g2, 1, 3, 2)):
                    st = o.state[p]
                    off = row0 * C * 4
                    d.param[k], d.grad[k] = p.data_ptr() + off, g.data_ptr() + row0 * C * gb
                    d.exp_avg[k], d.exp_avg_sq[k] = st["exp_avg"].data_ptr() + off, st["exp_avg_sq"].data_ptr() + off
                    d.half_shadow[k], d.shadow_mode[k] = pk.data_ptr() + row0 * 8, mode
                    d.numel[k], d.grad_is_half[k], d.clear_grad[k], d.slot[k] = n * C, is_half, 0, o._slot[p]
                    groups.append(group_of[p]); k += 1
                if with_mlp:
                    live = set(self.mlp_params[:5]) | (set(self.mlp_params[5:]) if full else set())
                    views = dict(zip(self.mlp_params, self.dw_views))
                    for p in [q for g in o.param_groups for q in g["params"]]:
                        if p not in live:
                            continue
                        st = o.state[p]
                        d.param[k], d.grad[k] = p.data_ptr(), views[p].data_ptr()
                        d.exp_avg[k], d.exp_avg_sq[k] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                        d.half_shadow[k], d.shadow_mode[k] = None, 0
                        d.numel[k], d.grad_is_half[k], d.clear_grad[k], d.slot[k] = p.numel(), 0, 1, o._slot[p]
                        groups.append(group_of[p]); k += 1
                d.count = k
                out.append((d, groups))
            cached = self._desc[key] = tuple(out)
        for d, groups in cached:
            for k, gi in enumerate(groups):
                d.lr[k] = float(o.param_groups[gi]["initial_lr"]) * lr_factor
        return cached[0][0], cached[1][0]

    # ---- double-buffered table state of the fused optimizer pass
    def _is_alt(self, p1_ptr, p2_ptr):
        alt = self.fuse_adam["alt"]
        return alt is not None and alt["p"][0].data_ptr() == p1_ptr and alt["p"][1].data_ptr() == p2_ptr

    def _fuse_desc(self, lr_factor):
        """N2mAdamFuse for this step: live set = what the model / optimizer name now, other set = fuse_adam['alt']."""
        f, o, model = self.fuse_adam, self.optimizer, self.model
        ps = (model.encoder.embeddings, model.encoder_color.embeddings)
        if f["alt"] is None or getattr(o, "state_epoch", 0) != f.get("epoch"):
            f["alt"] = {"p": [torch.empty_like(p.data) for p in ps], "m": [torch.empty_like(p.data) for p in ps],
                        "v": [torch.empty_like(p.data) for p in ps]}
            f["epoch"], f["desc"] = getattr(o, "state_epoch", 0), {}
        alt = f["alt"]
        key = (ps[0].data_ptr(), ps[1].data_ptr(), self._packed.data_ptr())
        d = f["desc"].get(key)
        if d is None:
            d = L.AdamFuse()
            for t, p in enumerate(ps):
                st = o.state[p]
                d.p_in[t], d.m_in[t], d.v_in[t] = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                d.p_out[t], d.m_out[t], d.v_out[t] = alt["p"][t].data_ptr(), alt["m"][t].data_ptr(), alt["v"][t].data_ptr()
                d.slot[t] = o._slot[p]
            d.packed, d.first_level = self._packed.data_ptr(), f["first_level"]
            d.beta1, d.beta2 = (float(x) for x in o.param_groups[0]["betas"])
            d.eps, d.scale, d.bias = float(o.param_groups[0]["eps"]), o.scale.data_ptr(), o.bias.data_ptr()
            f["desc"] = {k: v for k, v in f["desc"].items() if k[2] == key[2]}
            f["desc"][key] = d
        group_of = {p: g for g in o.param_groups for p in g["params"]}
        for t, p in enumerate(ps):
            d.lr[t] = float(group_of[p]["initial_lr"]) * lr_factor
        return d

    def _fuse_swap(self):
        """The other buffer set holds the state of the finished step (n2m_adam_fuse_restore has completed it): make it the one the model
        and the optimizer name."""
        alt, o, model = self.fuse_adam["alt"], self.optimizer, self.model
        for t, p in enumerate((model.encoder.embeddings, model.encoder_color.embeddings)):
            st = o.state[p]
            p.data, alt["p"][t] = alt["p"][t], p.data
            st["exp_avg"], alt["m"][t] = alt["m"][t], st["exp_avg"]
            st["exp_avg_sq"], alt["v"][t] = alt["v"][t], st["exp_avg_sq"]
        a, b = model.encoder.embeddings, model.encoder_color.embeddings
        model._packed_key = (a._version, b._version, a.data_ptr(), b.data_ptr())       # same values, new address: the packed copy stays valid

    def _shard_ranges(self):
        """(first row, rows) of this rank's slice of the coarse half (levels 0..7) and of the fine half (levels 8..15)."""
        return self._shard_ranges_of(self.rank)

    def _shard_ranges_of(self, r):
        c0 = min(self._split, r * self._Cs)
        return {"c": (c0, min(self._Cs, self._split - c0)), "f": (self._split + r * self._Fs, self._Fs)}

    def _adam_desc_sharded(self, full, key, pk, params):
        """Descriptor of the rank's OWN rows: per level half one [rows,1] + one [rows,2] entry (pointers offset into the full state
        tensors, gradients = the reduce-scattered slices, working copy = the same rows of the packed table), then the MLP weights."""
        o, model = self.optimizer, self.model
        desc = L.AdamDesc()
        e1p, e2p = model.encoder.embeddings, model.encoder_color.embeddings
        group_of = {p: gi for gi, g in enumerate(o.param_groups) for p in g["params"]}
        k, participants, groups = 0, 0, []
        for h, (row0, n) in self._shard_ranges().items():
            for p, C, g, is_half, mode in ((e1p, 1, self.g1s[h], 0, 2), (e2p, 2, self.g2s[h], 1, 3)):
                st = o.state[p]
                off = row0 * C * 4
                desc.param[k], desc.grad[k] = p.data_ptr() + off, g.data_ptr()
                desc.exp_avg[k], desc.exp_avg_sq[k] = st["exp_avg"].data_ptr() + off, st["exp_avg_sq"].data_ptr() + off
                desc.half_shadow[k], desc.shadow_mode[k] = pk.data_ptr() + row0 * 8, mode
                desc.numel[k], desc.grad_is_half[k], desc.clear_grad[k] = n * C, is_half, 0
                desc.slot[k] = o._slot[p]
                participants |= 1 << (o._slot[p] - 1)
                groups.append(group_of[p])
                k += 1
        live = set(self.mlp_params[:5]) | (set(self.mlp_params[5:]) if full else set())
        views = dict(zip(self.mlp_params, self.dw_views))
        for p in params:
            if p not in live:
                continue
            st = o.state[p]
            desc.param[k], desc.grad[k] = p.data_ptr(), views[p].data_ptr()
            desc.exp_avg[k], desc.exp_avg_sq[k] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            desc.half_shadow[k], desc.shadow_mode[k] = None, 0
            desc.numel[k], desc.grad_is_half[k], desc.clear_grad[k] = p.numel(), 0, 1
            desc.slot[k] = o._slot[p]
            participants |= 1 << (o._slot[p] - 1)
            groups.append(group_of[p])
            k += 1
        desc.count = k
        d = self._desc[key] = (desc, participants, groups)
        return d

    def _gather_packed(self):
        """All-gather of the packed rows after the sharded Adam (each rank has refreshed its own rows), as two chunks: the coarse half of the
        levels (32 % of the bytes) first, then the fine half.  Nothing is waited for here: the next step's lookup waits for the coarse chunk,
        looks its eight levels up (n2m_grid_encode_forward_packed_levels) while the fine chunk is still on the wire, then waits for that one
        (_wait_gather); everything else that reads the packed table or writes this rank's rows waits for both first."""
        import torch.distributed as dist
        flat = self._packed.view(-1)                 # [rows * 2] fp32 words, 8 bytes per row
        self._gathers = {}
        for h in ("c", "f"):
            row0, n = self._shard_ranges()[h]
            lo = 0 if h == "c" else self._split
            out = flat[lo * 2:(lo + self.world * n) * 2]
            mine = flat[row0 * 2:(row0 + n) * 2]
            self._gathers[h] = dist.all_gather_into_tensor(out, mine if self._inplace_gather else mine.clone(), async_op=True)
        if not self.chunked_gather:
            self._wait_gather()

    def _wait_gather(self, which=("c", "f")):
        g = getattr(self, "_gathers", None)
        if not g:
            return
        for h in which:
            w = g.pop(h, None)
            if w is not None:
                w.wait()

    @torch.no_grad()
    def sync_parameters(self, density_only=False, moments=False):
        """Sharded optimizer: bring the fp32 parameter tensors of both tables up to date on every rank (each rank owns 1/W of the
        rows; the training step itself only needs the packed copy).  Called before the occupancy refresh (density table, every 16
        steps: 24.5 MB) and to be called before a checkpoint, an export, an evaluation or a comparison."""
        if not self.shard:
            return
        self._wait_gather()
        if self.peer is not None:
            self.peer.check()
        # COLLECTIVE: every rank must call it at the same step.  A repeat at the same step is a no-op (so a rank may evaluate or save on
        # its own after all ranks have synchronised once -- bench.py's rank 0 does).  moments=True also gathers Adam's exp_avg / exp_avg_sq
        # of both tables (each rank has only advanced its own rows): what a checkpoint needs -- FusedAdamAMP.state_dict() of a sharded
        # run calls it, so optimizer.state_dict() is a collective too (the reference saves complete optimizer state, nerf/utils.py:1336-1350)
        done = getattr(self, "_synced", (-1, False, False))
        level = (not density_only, moments)
        if done[0] == self.global_step and (done[1] or density_only) and (done[2] or not moments):
            return
        self._synced = (self.global_step, level[0] or (done[0] == self.global_step and done[1]), level[1] or (done[0] == self.global_step and done[2]))
        import torch.distributed as dist
        e1p, e2p = self.model.encoder.embeddings, self.model.encoder_color.embeddings
        tables = [(e1p.data, 1)] if density_only else [(e1p.data, 1), (e2p.data, 2)]
        if moments:
            for p, C in ((e1p, 1), (e2p, 2)):
                st = self.optimizer.state[p]
                tables += [(st["exp_avg"], C), (st["exp_avg_sq"], C)]
        for t, C in tables:
            flat = t.view(-1)
            for h, (row0, n) in self._shard_ranges().items():
                lo = 0 if h == "c" else self._split
                if h == "c" and getattr(self, "_uneven", False):      # padded chunks (peer-store mode): the last one is shorter -- one broadcast per owner
                    for r in range(self.world):
                        r0, rn = self._shard_ranges_of(r)["c"]
                        if rn > 0:
                            dist.broadcast(flat[r0 * C:(r0 + rn) * C], src=r)
                    continue
                dist.all_gather_into_tensor(flat[lo * C:(lo + self.world * n) * C], flat[row0 * C:(row0 + n) * C].clone())

    def _optimizer_step(self, full, lr_factor, loss_out=None, fused=None):
        o = self.optimizer
        self._wait_gather()          # (a step without samples ran no lookup: the rows the last gather sends must not be rewritten under it)
        desc, participants, groups = self._adam_desc(full, dense_only=fused is not None)
        for k, gi in enumerate(groups):
            desc.lr[k] = float(o.param_groups[gi]["initial_lr"]) * lr_factor
        b1, b2 = o.param_groups[0]["betas"]
        s = L.stream()
        peer_fused = self.peer is not None and self._peer_fused
        scaler_done = False
        if self.lookup_overlap and fused is None and not self.shard and self.Lv == 16:
            # fine rows (levels 8..15) first, an event behind them, then the coarse rows + every other tensor
            d_f, d_c = self._adam_desc_halves(full, lr_factor)
            for d in (d_f, d_c):
                L.call("n2m_adam_step", ctypes.addressof(d), float(b1), float(b2), float(o.param_groups[0]["eps"]), _p(o.scale), _p(o.found_inf),
                       _p(o.bias), s)
                if d is d_f:
                    self._fine_ready = torch.cuda.Event()
                    self._fine_ready.record()
        elif peer_fused:
            if self._adam_peer is None:     # entries 0..3 of the sharded descriptor: (density, colour) of the coarse half, then of the fine half
                self._adam_peer = self.peer.adam_peer([("s1", "c"), ("s2", "c"), ("s1", "f"), ("s2", "f")] + [None] * (desc.count - 4))
            L.call("n2m_adam_step_peer", ctypes.addressof(desc), float(b1), float(b2), float(o.param_groups[0]["eps"]), _p(o.scale), _p(o.found_inf),
                   _p(o.bias), ctypes.addressof(self._adam_peer), s)
        elif (self.adam_tail and loss_out is not None and fused is None and self.peer is None and not self.shard):
            # optimizer pass + the scaler / step counts / loss value in one launch (the tail of the pass's last workgroup)
            gf, bf, gi = o.growth
            n_rays, buf, extra = loss_out[:3]
            extra2 = loss_out[3] if len(loss_out) > 3 else None
            ex_buf, ex_scale = extra if extra is not None else (None, 0.0)
            e2_buf, e2_n, e2_scale = extra2 if extra2 is not None else (None, 0, 0.0)
            tail = L.ScalerTail(_p(o.growth_tracker), _p(o.steps), participants, gf, bf, gi, _p(self._w["partial"]), (n_rays + 15) // 16, n_rays,
                                _p(buf), _p(self._loss_sum), _p(ex_buf), self._n_spec, float(ex_scale), _p(e2_buf), int(e2_n), float(e2_scale),
                                _p(self._tail_ticket))
            L.call("n2m_adam_step_scaler", ctypes.addressof(desc), float(b1), float(b2), float(o.param_groups[0]["eps"]), _p(o.scale), _p(o.found_inf),
                   _p(o.bias), ctypes.addressof(tail), s)
            scaler_done = True
        else:
            L.call("n2m_adam_step", ctypes.addressof(desc), float(b1), float(b2), float(o.param_groups[0]["eps"]), _p(o.scale), _p(o.found_inf),
                   _p(o.bias), s)
        if fused is not None:      # behind both optimizer passes, in front of the scaler update that clears found_inf
            L.call("n2m_adam_fuse_restore", ctypes.addressof(fused), self.ho.ctypes.data, self.Lv, _p(o.found_inf), s)
            self._fuse_swap()
        if self.peer is not None:      # this rank's refreshed rows into every rank's packed table, coarse chunk first; the next lookup waits per half
            for h, (row0, n) in self._shard_ranges().items():
                if peer_fused:
                    self.peer.signal_rows(h)               # (Adam has stored them everywhere itself)
                else:
                    self.peer.push_rows(h, row0, n)
            self._gathers = self.peer.rows_tokens()
            if not self.chunked_gather:
                self._wait_gather()
        elif self.shard:
            self._gather_packed()
        gf, bf, gi = o.growth
        if scaler_done:
            pass
        elif loss_out is None:
            L.call("n2m_scaler_update_slots", _p(o.scale), _p(o.growth_tracker), _p(o.found_inf), _p(o.steps), _p(o.bias), participants,
                   float(b1), float(b2), gf, bf, gi, s)
        else:        # + the step's loss value from the compositing kernel's per-workgroup partials
            n_rays, buf, extra = loss_out[:3]
            extra2 = loss_out[3] if len(loss_out) > 3 else None
            ex_buf, ex_scale = extra if extra is not None else (None, 0.0)
            e2_buf, e2_n, e2_scale = extra2 if extra2 is not None else (None, 0, 0.0)
            L.call("n2m_scaler_update_slots_loss3", _p(o.scale), _p(o.growth_tracker), _p(o.found_inf), _p(o.steps), _p(o.bias), participants,
                   float(b1), float(b2), gf, bf, gi, _p(self._w["partial"]), (n_rays + 15) // 16, n_rays, _p(buf), _p(self._loss_sum),
                   _p(ex_buf), self._n_spec, float(ex_scale), _p(e2_buf), int(e2_n), float(e2_scale), s)
        nxt = lr_lambda(self.global_step, self.opt.iters)          # like LambdaLR.step(): param_groups carry the NEXT step's rate
        for group in o.param_groups:
            group["lr"] = float(group["initial_lr"]) * nxt

    # ------------------------------------------------------------------------------------------------------- step
    @torch.no_grad()
    def train_step(self):
        opt, model, dev = self.opt, self.model, self.device
        if not model.training:
            model.train()
        for g in self.optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        if not self._queue:
            self._fill_pipeline()
        b = self._queue.pop(0)
        self.global_step += 1
        assert b.index == self.global_step
        N = b.N
        random_bg = b.bg is not None
        bg = b.bg
        shading = SHADING["diffuse" if (self.global_step < opt.diffuse_step or opt.diffuse_only) else "full"]
        M = self._finish(b)
        self.last_num_points = M
        self.samples_seen += M
        self.rays_seen += N

        if opt.sdf:               # schedules of the SDF recipe (nerf/utils.py:651-655), exactly as trainer.Stage0Trainer sets them
            opt.cos_anneal_ratio = min(1, self.global_step / (0.5 * opt.iters))
            opt.normal_anneal_epsilon = 1e-1 * (1 - min(0.999, self.global_step / (0.5 * opt.iters)))
            if opt.progressive_level:
                model.max_level = 4 + int(12 * min(1, self.global_step / (0.5 * opt.iters)))
        w = self._work(max(M, 1), N)
        s = L.stream()
        c = b.cap_m
        xyzs = dirs = ts = None
        if M > 0:
            xyzs, dirs, ts = b.samples[:3 * c], b.samples[3 * c:6 * c], b.samples[6 * c:]
        o = self.optimizer
        pk = model.packed_tables()
        sw = self.mlp_params
        e1 = model.encoder
        # seed gradient = loss scale [/ world]: gradients are SUMMED over ranks
        seed = o.scale if self.world == 1 else o.scale / self.world
        if opt.sdf:
            return self._step_sdf(b, M, N, w, xyzs, dirs, ts, shading, seed, pk)
        # ---- forward
        fwd_terms = False
        if M > 0:
            fwd = (_p(xyzs), _p(pk), _p(e1.offsets), _p(w["h1"]), _p(w["h2"]), M, self.Lv, self.Lv, self.S, self.H0, e1.gridtype_id,
                   int(bool(e1.align_corners)), e1.interp_id, float(self.aff[0]), float(self.aff[1]))
            if self.shard and getattr(self, "_gathers", None) and self.Lv == 16:
                # the packed rows of the other ranks are still arriving: coarse half first, its lookup runs under the fine half's transfer
                self._wait_gather(("c",))
                L.call("n2m_grid_encode_forward_packed_levels", *fwd, 0, 8, s)
                self._wait_gather(("f",))
                L.call("n2m_grid_encode_forward_packed_levels", *fwd, 8, 8, s)
            elif self.lookup_overlap and self._fine_ready is not None and self.Lv == 16:
                # fine levels on their own stream behind the fine half of the last optimizer pass, coarse levels on the main stream behind all of it
                self.s_lookup.wait_event(self._fine_ready)
                with torch.cuda.stream(self.s_lookup):
                    L.call("n2m_grid_encode_forward_packed_levels", *fwd, 8, 8, L.stream())
                    done = torch.cuda.Event()
                    done.record()
                L.call("n2m_grid_encode_forward_packed_levels", *fwd, 0, 8, s)
                torch.cuda.current_stream(dev).wait_event(done)
            elif self.tv_corners and self.Lv == 16 and self.fuse_adam is None:
                L.call("n2m_grid_encode_forward_packed_tv", *fwd, _p(w["tv4"]), s)
                self._corners_of = (self.global_step, M)          # the records in w["tv4"] belong to THIS step's samples
            elif self.tv_fwd and self.Lv == 16 and self.fuse_adam is None:
                # (weights as in the backward's tv_args below; `seed` = the loss scale this step's backward runs under: the scaler's update of the
                #  last step is ahead of this launch on the stream)
                L.call("n2m_grid_encode_forward_packed_tvterms", *fwd, float(opt.lambda_tv), float(opt.lambda_tv * (10 if opt.bound > 1 else 1)),
                       float(0.5 / model.bound), _p(seed), _p(w["tv"]), s)
                fwd_terms = True
            else:
                self._wait_gather()
                L.call("n2m_grid_encode_forward_packed", *fwd, s)
            self._mid_step_fill()
            tv_terms = None

            def start_tv():
                # the TV terms need the samples and the density table only: evaluated on their own stream while the field kernels run
                go = torch.cuda.Event()
                go.record()
                self.tv_stream.wait_event(go)
                L.grid_backward_config(*self._bwd_cfg)
                with torch.cuda.stream(self.tv_stream):
                    L.call("n2m_grid_tv_terms", _p(xyzs), _p(pk) if self._bwd_cfg[0] == 2 else _p(e1.embeddings), self.ho.ctypes.data, M, self.Lv, self.S, self.H0,
                           e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id, float(opt.lambda_tv),
                           float(opt.lambda_tv * (10 if opt.bound > 1 else 1)), float(0.5 / model.bound),
                           _p(seed), float(self.aff[0]), float(self.aff[1]), _p(w["tv"]), L.stream())
                    self._tv_done = torch.cuda.Event()
                    self._tv_done.record()
            want_tv = opt.lambda_tv > 0 and self.tv_split and self.Lv == 16
            if want_tv:
                tv_terms = w["tv"]
                if self.tv_at == 0:
                    start_tv()
            elif fwd_terms:
                tv_terms = w["tv"]                               # written by this step's lookup, on this stream
                self._tv_done = None
            # full shading: the specular regulariser (nerf/utils.py:733-737) rides in the field kernels -- the forward leaves per-workgroup
            # sums of specular^2, the backward adds 2 lambda / M * specular * seed to the recomputed activation's gradient; the [M,3]
            # specular tensor is neither written nor read
            spec_reg = shading != 0 and opt.lambda_specular > 0
            L.call("n2m_field_forward_train", _p(xyzs), _p(dirs) if shading != 0 else None, _p(w["h1"]), _p(w["h2"]), *[_p(p) for p in sw], M, shading, 1,
                   _p(w["sigma"]), _p(w["rgb"]), None, _p(w["spec_partial"]) if spec_reg else None, s)
            if want_tv and self.tv_at == 1:
                start_tv()
        bg_t, bg_s = (bg, 0.0) if random_bg else (None, 1.0)
        lam_rgb, lam_mask = float(opt.lambda_rgb), float(max(opt.lambda_mask, 0.0))
        # ---- compositing + loss head + both backward passes: one launch (seed gradient = loss scale [/ world]: gradients are SUMMED over ranks)
        early = None
        d_sigma, d_rgb = w["d_sr"][:max(M, 1)], w["d_sr"][max(M, 1):4 * max(M, 1)]
        # (+ the entropy regulariser of config 4, nerf/utils.py:728-733: its per-sample gradient is the backward's grad_weights)
        order = self.live_first and M > 0 and M <= (1 << 20) and self.peer is None
        if order:
            L.call("n2m_composite_live_counts", _p(w["live"]), _p(w["block_live"]))
        try:
            L.call("n2m_composite_loss_train_ent", _p(w["sigma"]), _p(w["rgb"]), _p(ts), _p(b.rays), M, N, 1e-4, _p(b.rgba), _p(bg_t), bg_s, lam_rgb, lam_mask,
                   _p(seed), None, None, _p(d_sigma), _p(d_rgb), _p(w["partial"]), None, None, None, float(max(opt.lambda_entropy, 0.0)), s)      # loss value: summed by the scaler kernel
        finally:
            if order:
                L.call("n2m_composite_live_counts", None, None)
        if order:
            L.call("n2m_sample_order_live_first", _p(b.rays), _p(w["live"]), _p(w["block_live"]), N, M, _p(w["perm"]), s)
            if self._identity_order:
                torch.arange(M, dtype=torch.int32, device=dev, out=w["perm"][:M])
        if M > 0:
            if want_tv and self.tv_at == 2:
                start_tv()
            if self.marker_at == 2:
                self._marker = torch.cuda.Event(); self._marker.record()
            L.call("n2m_field_backward_train", _p(xyzs), _p(dirs) if shading != 0 else None, _p(w["h1"]), _p(w["h2"]), *[_p(p) for p in sw], M, shading, 1,
                   _p(d_sigma), _p(d_rgb), None, _p(w["d_h1"]), _p(w["d_h2"]), *[_p(g) for g in self.dw_views], _p(o.found_inf),
                   float(2.0 * opt.lambda_specular / M) if spec_reg else 0.0, _p(seed) if spec_reg else None, s)
            L.grid_backward_config(*self._bwd_cfg)
            need = L.lib().n2m_grid_binned_pair_workspace_bytes(M, self.Lv, self.ho.ctypes.data)
            ws = L.workspace(dev, need)
            tv = opt.lambda_tv > 0
            common = (_p(w["d_h1"]), _p(w["d_h2"]), _p(xyzs), self.ho.ctypes.data, _p(self.g1), _p(self.g2), M,
                      self.Lv, self.Lv, self.S, self.H0, e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id)
            tail = (_p(o.found_inf), float(self.aff[0]), float(self.aff[1]), 1, _p(ws), ws.numel(), s)
            fused = None
            if self.fuse_adam is not None and M <= self._fuse_cap and tv_terms is None:
                self._adam_desc(shading != 0, dense_only=True)             # (names self._packed)
                fused = self._fuse_desc(lr_lambda(self.global_step - 1, opt.iters))
            if fused is not None:
                tv_args = (_p(pk) if tv else None, float(opt.lambda_tv), float(opt.lambda_tv * (10 if opt.bound > 1 else 1)),
                           float(0.5 / model.bound), _p(seed) if tv else None)
                backward = lambda half: L.call("n2m_grid_encode_backward_binned_pair_adam", *common, *tv_args, *tail[:3], _p(ws), ws.numel(),
                                               ctypes.addressof(fused), s)
            elif tv_terms is not None:
                if self._tv_done is not None:
                    torch.cuda.current_stream(dev).wait_event(self._tv_done)
                backward = lambda half: L.call("n2m_grid_encode_backward_binned_pair_tvt", *common, _p(tv_terms), *tail, half)
            else:
                tv_args = ((_p(pk) if self._bwd_cfg[0] == 2 else _p(e1.embeddings)) if tv else None, float(opt.lambda_tv),
                           float(opt.lambda_tv * (10 if opt.bound > 1 else 1)), float(0.5 / model.bound), _p(seed) if tv else None)
                backward = lambda half: (L.call("n2m_grid_encode_backward_binned_pair", *common, *tv_args, *tail) if half == 0 else
                                         L.call("n2m_grid_encode_backward_binned_pair_half", *common, *tv_args, *tail, half))
            if self.marker_at == 1:
                self._marker = torch.cuda.Event(); self._marker.record()
            if order and fused is None:
                plain_backward = backward

                def backward(half, _b=plain_backward, _perm=w["perm"]):
                    # the order is a sticky thread-local of the library: set right in front of the call it is meant for, cleared behind it
                    # whatever happens in between (an exception must not leave a stale pointer for the next backward of this thread)
                    L.call("n2m_grid_backward_sample_order", _p(_perm))
                    try:
                        return _b(half)
                    finally:
                        L.call("n2m_grid_backward_sample_order", None)
            if self.peer is not None:
                # the flush of each level half stores its rows into their owners' slots; the signal behind it is the whole exchange
                self.peer.begin_step()
                L.call("n2m_grid_backward_peer_route", ctypes.byref(self._peer_route))
                try:
                    backward(1)
                    self.peer.signal_grad("f")
                    backward(2)
                    self.peer.signal_grad("c")
                finally:
                    L.call("n2m_grid_backward_peer_route", None)
                early = []
            elif self.shard:
                import torch.distributed as dist
                sp = self._split
                rs = lambda out, src: dist.reduce_scatter_tensor(out.view(-1), src.view(-1), op=dist.ReduceOp.SUM, async_op=True)
                backward(1)
                early = [rs(self.g1s["f"], self.g1[sp:]), rs(self.g2s["f"], self.g2[sp:])]          # fine rows: exchanged under the coarse half
                backward(2)
                early += [rs(self.g1s["c"], self.g1[:sp]), rs(self.g2s["c"], self.g2[:sp])]
            elif self.sync is not None and self.split_backward and self.Lv == 16:
                # multi-GPU: the table backward in two halves of the levels.  The rows of the fine half (levels 8..15: 68 % of the
                # bytes) are final after the first call; their SUM all-reduce runs on the collective stream while the coarse half is
                # still being computed, so most of the exchange hides behind the backward's own tail (at 0.75 ms per step
                # a 49 MB all-reduce no longer is the "< 4 %" the survey estimated at 26 ms per step)
                split = int(self.ho[8])
                backward(1)
                early = self.sync.all_reduce_sum_begin([self.g1[split:], self.g2[split:]], [])
                backward(2)
            else:
                mid = self.marker_at == 3 and fused is None and tv_terms is None
                if mid:
                    if self._mid_events is None:          # a few events, created (= recorded once) up front, re-recorded by the library
                        self._mid_events = [torch.cuda.Event() for _ in range(4)]
                        for e in self._mid_events:
                            e.record()
                    ev = self._mid_events[self.global_step % len(self._mid_events)]
                    L.call("n2m_grid_backward_mid_event", ctypes.c_void_p(ev.cuda_event))
                use_corners = (self.tv_corners and fused is None and tv_terms is None and tv
                               and getattr(self, "_corners_of", None) == (self.global_step, M))      # ... and only if THIS step's forward wrote them
                if use_corners:      # (sticky thread-local of the library: set in front of the call it is meant for, cleared behind it whatever happens)
                    L.call("n2m_grid_backward_tv_corners", _p(w["tv4"]))
                try:
                    backward(0)
                finally:
                    if use_corners:
                        L.call("n2m_grid_backward_tv_corners", None)
                if mid:
                    self._marker = ev
        else:
            # no sample in the batch: every gradient is zero (the reduction below still takes part on every rank)
            self.g1.zero_()
            self.g2.zero_()
            if self.peer is not None:     # zeros into this rank's slots -- once every owner's rows of the last step have arrived (see PeerExchange)
                self._wait_gather()
                self.peer.begin_step()
                self.peer.zero_slots()
                early = []
            elif self.shard:      # same collectives in the same ORDER as on the ranks that have samples (fine halves, coarse halves, then the bucket)
                import torch.distributed as dist
                sp = self._split
                rs = lambda out, src: dist.reduce_scatter_tensor(out.view(-1), src.view(-1), op=dist.ReduceOp.SUM, async_op=True)
                early = [rs(self.g1s["f"], self.g1[sp:]), rs(self.g2s["f"], self.g2[sp:]), rs(self.g1s["c"], self.g1[:sp]), rs(self.g2s["c"], self.g2[:sp])]
            elif self.sync is not None and self.split_backward and self.Lv == 16:
                early = self.sync.all_reduce_sum_begin([self.g1[int(self.ho[8]):], self.g2[int(self.ho[8]):]], [])
        # ---- [multi-GPU] one SUM all-reduce per fixed gradient buffer (the colour table's stays fp16) + the small bucket
        if self.peer is not None:
            # no collective anywhere in the step: weight gradients + non-finite flag go to every rank's slot and are summed in rank order
            # everywhere; then, owner side, the W gradient slots of each level half -> their sum where the reduce-scatter would have put it
            n_dw = self.dw.numel()
            torch.cat([self.dw, o.found_inf.reshape(-1)], out=self._peer_small)
            self.peer.all_sum_small(self._peer_small)
            self.dw.copy_(self._peer_small[:n_dw])
            o.found_inf.copy_(self._peer_small[n_dw:].view_as(o.found_inf))
            for h in ("f", "c"):
                if self._peer_fused:
                    self.peer.wait_grad(h)                    # (the sum itself happens in n2m_adam_step_peer's gradient load)
                else:
                    self.peer.reduce(h, self.g1s[h], self.g2s[h])
            # a wait that ran into its timeout has summed stale slots: the step is skipped on the device (found_inf) and the host raises as
            # soon as it sees the error word (one step later at most: the word travels to pinned memory behind the step's last wait)
            self.peer.fold_error_into(o.found_inf)
            self.peer.raise_if_failed()
        elif self.shard:
            token = self.sync.all_reduce_sum_begin([], [self.dw, o.found_inf])
            for w_ in early:
                w_.wait()
            self.sync.all_reduce_sum_end(token)
        elif self.sync is not None:
            if early is not None:
                split = int(self.ho[8])
                token = self.sync.all_reduce_sum_begin([self.g1[:split], self.g2[:split]], [self.dw, o.found_inf])
                self.sync.all_reduce_sum_end(early)
            else:
                token = self.sync.all_reduce_sum_begin([self.g1, self.g2], [self.dw, o.found_inf])
            self.sync.all_reduce_sum_end(token)
        # ONE event per step on the main stream (an event record is a marker packet the queue idles ~6 us behind, measured): behind the
        # last kernel that reads this batch's buffers and in front of the optimizer update -- the side stream's go-ahead
        if self.marker_at == 0 or M == 0 or (self.marker_at == 3 and not (M > 0 and self.sync is None and fused is None and tv_terms is None)):
            self._marker = torch.cuda.Event()
            self._marker.record()
        # ---- Adam + loss-scale bookkeeping, LR schedule (main.py:239)
        extra = (w["spec_partial"], float(opt.lambda_specular / M)) if (M > 0 and shading != 0 and opt.lambda_specular > 0) else None
        self._lr_step(shading != 0, loss_out=(N, b.loss, extra), fused=fused if M > 0 else None)
        loss = b.loss.view(())                 # written by the scaler kernel (photometric + specular terms); lives in the batch's buffer set (valid until the set comes round again)
        self._fill_pipeline()
        return loss

    # ------------------------------------------------------------------------------------------------ SDF recipe (config 5)
    def _sdf_buf(self, M=0):
        """Buffers of the SDF head, grow-only: the six finite-difference copies of the batch (points, [0,1] points, density features, sdf
        values and their gradients), alpha, per-workgroup partials, the variance gradient."""
        cap = getattr(self, "_sdf_cap", 0)
        if not hasattr(self, "_sdf") or M > cap:
            cap = max(cap, int(M * 1.25) + 1024)
            dev = self.device
            f = lambda n: torch.empty(n, dtype=torch.float32, device=dev)
            old = getattr(self, "_sdf", None)
            self._sdf = {"pts": f(18 * cap), "pts01": f(18 * cap), "h6": f(16 * 6 * cap), "d_h6": f(16 * 6 * cap), "s6": f(6 * cap), "d_s6": f(6 * cap),
                         "alpha": f(cap), "d_sdf": f(cap), "x01": f(3 * cap), "eik": f((cap + 255) // 256 + 1), "varp": f((cap + 255) // 256 + 1),
                         "d_var": old["d_var"] if old is not None else torch.zeros(1, dtype=torch.float32, device=dev),      # (the Adam descriptor holds its address)
                         # folded copies (n2m_sdf_fold_*): per-sample flags, the compact list of the copies that keep the stacked pass
                         "fold_cnt": old["fold_cnt"] if old is not None else torch.zeros(2, 32, dtype=torch.int32, device=dev),
                         "fold_host": old["fold_host"] if old is not None else torch.zeros(32, dtype=torch.int32).pin_memory()}
            self._sdf.pop("fold", None)
            self._sdf_cap = cap
        return self._sdf

    def _step_sdf(self, b, M, N, w, xyzs, dirs, ts, shading, seed, pk):
        """The iteration of the SDF recipe (nerf/renderer.py:724-741, nerf/network.py:143-154, nerf/utils.py:651-655,740-743) as a fixed launch
        sequence: lookup -> field (raw sdf) -> six finite-difference copies through the density encoder + sigma_net (ONE stacked call each
        way) -> NeuS alpha (n2m_sdf_alpha_forward) -> compositing + loss in alpha mode -> n2m_sdf_alpha_backward (+ eikonal) -> field
        backward (batch and stacked copies) -> table backward (shared fill for the batch, the density table alone for the copies, TV) ->
        Adam (+ the variance) -> bookkeeping (loss = photometric + specular + eikonal terms).  trainer.Stage0Trainer is the parity baseline."""
        opt, model, dev, o = self.opt, self.model, self.device, self.optimizer
        s = L.stream()
        e1 = model.encoder
        sw = self.mlp_params
        ml = int(min(model.max_level, self.Lv))
        eps, car = float(opt.normal_anneal_epsilon), float(opt.cos_anneal_ratio)
        random_bg = b.bg is not None
        bg_t, bg_s = (b.bg, 0.0) if random_bg else (None, 1.0)
        lam_rgb, lam_mask = float(opt.lambda_rgb), float(max(opt.lambda_mask, 0.0))
        sb = self._sdf_buf(M)
        d_sigma, d_rgb = w["d_sr"][:max(M, 1)], w["d_sr"][max(M, 1):4 * max(M, 1)]
        extra = extra2 = None
        if M > 0:
            M6 = 6 * M
            if ml < self.Lv:          # progressive levels: the kernels leave the inactive levels alone, the field reads all sixteen
                w["h1"][ml * M:16 * M].zero_()
                w["h2"][2 * ml * M:32 * M].zero_()
                sb["h6"][ml * M6:16 * M6].zero_()
            self._wait_gather()
            L.call("n2m_grid_encode_forward_packed", _p(xyzs), _p(pk), _p(e1.offsets), _p(w["h1"]), _p(w["h2"]), M, self.Lv, ml, self.S,
                   self.H0, e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id, float(self.aff[0]), float(self.aff[1]), s)
            self._mid_step_fill()
            spec_reg = shading != 0 and opt.lambda_specular > 0
            L.call("n2m_field_forward_train", _p(xyzs), _p(dirs) if shading != 0 else None, _p(w["h1"]), _p(w["h2"]), *[_p(p) for p in sw], M, shading, 3,
                   _p(w["sigma"]), _p(w["rgb"]), None, _p(w["spec_partial"]) if spec_reg else None, s)
            # finite-difference normals: six offset copies, one encode + one sigma_net evaluation for all of them
            L.call("n2m_sdf_offsets", _p(xyzs), M, eps, float(model.bound), _p(sb["pts"]), _p(sb["pts01"]), s)
            # Table backward of the copies: once epsilon is a fraction of the finest active cell, a copy nearly always lies in its centre
            # sample's cell on a given level -- those (copy, level) pairs fold into the batch's own backward
            # (n2m_grid_encode_backward_binned_pair_fold), the others (2.4 % at the end of the schedule) take a density-only call over
            # per-level lists whose lengths the host reads back (known long before the backward is enqueued: the plan only needs the samples)
            finest = self.H0 * 2.0 ** (self.S * (ml - 1))
            # (the fold and the lists call are one-pass paths of the table backward: batches above 2^20 samples keep the stacked pass)
            fold = self.sdf_fold and eps / (2.0 * float(model.bound)) * finest < 0.25 and M <= (1 << 20)
            if fold:
                if "fold" not in sb:      # buffers of the fold, only once it is used: per-level flags and lists (capacity: every copy)
                    c6 = 6 * self._sdf_cap
                    sb["fold"] = {"cap": c6, "flags": torch.empty(16 * self._sdf_cap, dtype=torch.uint8, device=dev),
                                  "pts": torch.empty(16 * c6 * 3, dtype=torch.float32, device=dev),
                                  "src": torch.empty(16 * c6, dtype=torch.int32, device=dev), "g": torch.empty(16 * c6, dtype=torch.float32, device=dev)}
                fb = sb["fold"]
                par = self.global_step & 1
                # the plan kernel adds onto this parity's counters and clears the other parity's for the next step -- which only holds while
                # every step runs the plan; a step without it (no samples, the fold condition off for a step) would leave counts behind
                sb["fold_cnt"][par].zero_()
                L.call("n2m_sdf_fold_plan", _p(xyzs), M, eps, float(model.bound), self.Lv, ml, self.S, self.H0, int(bool(e1.align_corners)),
                       _p(fb["flags"]), _p(fb["pts"]), _p(fb["src"]), fb["cap"], _p(sb["fold_cnt"]), par, s)
                sb["fold_host"].copy_(sb["fold_cnt"][par], non_blocking=True)
                fold_ready = torch.cuda.Event()
                fold_ready.record()
            L.call("n2m_grid_encode_forward", _p(sb["pts01"]), _p(e1.embeddings), _p(e1.offsets), _p(sb["h6"]), M6, 3, 1, self.Lv, ml, self.S, self.H0,
                   None, e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id, L.F32, s)
            L.call("n2m_field_forward", _p(sb["pts"]), None, _p(sb["h6"]), None, *[_p(p) for p in sw], M6, 0, 2, _p(sb["s6"]), None, None, s)
            L.call("n2m_sdf_alpha_forward", _p(w["sigma"]), _p(sb["s6"]), _p(dirs), _p(ts), M, _p(model.variance), eps, car, _p(sb["alpha"]), None,
                   _p(sb["eik"]) if opt.lambda_eikonal > 0 else None, s)
        # compositing in alpha mode + loss head + their backward
        L.call("n2m_composite_loss_train_ex", _p(sb["alpha"]), _p(w["rgb"]), _p(ts), _p(b.rays), M, N, 1e-4, _p(b.rgba), _p(bg_t), bg_s, lam_rgb, lam_mask,
               _p(seed), None, None, _p(d_sigma), _p(d_rgb), _p(w["partial"]), None, None, None, 0.0, 1, s)
        if M > 0:
            M6 = 6 * M
            lam_eik = float(opt.lambda_eikonal) if opt.lambda_eikonal > 0 else 0.0
            L.call("n2m_sdf_alpha_backward", _p(d_sigma), _p(w["sigma"]), _p(sb["s6"]), _p(dirs), _p(ts), M, _p(model.variance), eps, car, _p(seed),
                   float(lam_eik * 2.0 / M), _p(sb["d_sdf"]), _p(sb["d_s6"]), _p(sb["varp"]), _p(sb["d_var"]), _p(o.found_inf), s)
            L.call("n2m_field_backward_train", _p(xyzs), _p(dirs) if shading != 0 else None, _p(w["h1"]), _p(w["h2"]), *[_p(p) for p in sw], M, shading, 3,
                   _p(sb["d_sdf"]), _p(d_rgb), None, _p(w["d_h1"]), _p(w["d_h2"]), *[_p(g) for g in self.dw_views], _p(o.found_inf),
                   float(2.0 * opt.lambda_specular / M) if spec_reg else 0.0, _p(seed) if spec_reg else None, s)
            L.call("n2m_field_backward", _p(sb["pts"]), None, _p(sb["h6"]), None, *[_p(p) for p in sw], M6, 0, 2, _p(sb["d_s6"]), None, None,
                   _p(sb["d_h6"]), None, *[_p(g) for g in self.dw_views], _p(o.found_inf), s)
            # table gradients: the batch through the shared fill (both tables, overwrite), the stacked copies onto the density table alone
            L.grid_backward_config(*self._bwd_cfg)
            need = max(L.lib().n2m_grid_binned_pair_workspace_bytes(M, ml, self.ho.ctypes.data),
                       L.lib().n2m_grid_binned_pair_workspace_bytes(M6, ml, self.ho.ctypes.data))
            ws = L.workspace(dev, need)
            tv_fold = opt.lambda_tv > 0 and ml == self.Lv
            tv_w, tv_wo = float(opt.lambda_tv), float(opt.lambda_tv * (10 if opt.bound > 1 else 1))
            geo = (self.Lv, ml, self.S, self.H0, e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id)
            geo_batch = geo
            if opt.lambda_tv > 0 and 6 <= ml < self.Lv and self.sdf_tv_all:
                # progressive levels: the TV term covers ALL sixteen levels (grid.py:170-192 has no max_level).  Instead of the batch's backward
                # over the active levels + the TV as its own 16-level pass (185 us), the batch's backward runs over all sixteen with the TV
                # folded in and ZERO feature gradients on the inactive levels (the encoder's backward ignores them, grid.py:71-95): one pass
                # on the 1-D XCD grid.  Pays from about six active levels on.
                w["d_h1"][ml * M:16 * M].zero_()
                w["d_h2"][2 * ml * M:32 * M].zero_()
                geo_batch = (self.Lv, self.Lv) + geo[2:]
                tv_fold = True
                ws = L.workspace(dev, max(need, L.lib().n2m_grid_binned_pair_workspace_bytes(M, self.Lv, self.ho.ctypes.data)))
            batch_args = (_p(w["d_h1"]), _p(w["d_h2"]), _p(xyzs), self.ho.ctypes.data, _p(self.g1), _p(self.g2), M, *geo_batch,
                          _p(e1.embeddings) if tv_fold else None, tv_w, tv_wo, float(0.5 / model.bound), _p(seed) if tv_fold else None, _p(o.found_inf),
                          float(self.aff[0]), float(self.aff[1]), 1, _p(ws), ws.numel())
            if fold:
                t0 = time.perf_counter()
                while not fold_ready.query():
                    if time.perf_counter() - t0 > 5e-3:
                        fold_ready.synchronize()
                        break
                K = int(sb["fold_host"][:ml].max())
                self.last_fold_left = int(sb["fold_host"][:ml].sum())
                if K > (1 << 20):      # a level's list of unfolded copies exceeds the one-pass limit of the lists call: this step keeps the stacked pass
                    fold = False
            if fold:
                L.call("n2m_grid_encode_backward_binned_pair_fold", *batch_args, _p(fb["flags"]), _p(sb["d_h6"]), eps, float(model.bound), s)
                if K > 0:
                    L.call("n2m_sdf_fold_gather", _p(sb["d_h6"]), M, ml, _p(fb["src"]), _p(fb["pts"]), fb["cap"],
                           sb["fold_cnt"].data_ptr() + 128 * par, K, _p(fb["g"]), s)
                    L.call("n2m_grid_encode_backward_binned_lists", _p(fb["g"]), _p(fb["pts"]), fb["cap"], self.ho.ctypes.data, _p(self.g1), K, self.Lv, ml,
                           self.S, self.H0, e1.gridtype_id, int(bool(e1.align_corners)), e1.interp_id, _p(o.found_inf), 0, _p(ws), ws.numel(), s)
            else:
                L.call("n2m_grid_encode_backward_binned_pair", *batch_args, s)
                L.call("n2m_grid_encode_backward_binned_pair", _p(sb["d_h6"]), None, _p(sb["pts01"]), self.ho.ctypes.data, _p(self.g1), None, M6, *geo,
                       None, 0.0, 0.0, 1.0, None, _p(o.found_inf), 1.0, 0.0, 0, _p(ws), ws.numel(), s)
            if opt.lambda_tv > 0 and not tv_fold:
                # progressive levels: the TV term covers ALL levels of the table (grid.py:170-192 has no max_level), as its own pass
                x01 = sb["x01"][:3 * M]
                torch.add(xyzs[:3 * M], float(model.bound), out=x01)
                x01.div_(2.0 * float(model.bound))
                need_tv = L.lib().n2m_grid_binned_workspace_bytes(M, 3, 1, self.Lv, self.ho.ctypes.data, L.F32, 1)
                ws_tv = L.workspace(dev, need_tv, 1)
                L.call("n2m_grad_total_variation_binned", _p(x01), _p(e1.embeddings), _p(self.g1), self.ho.ctypes.data, tv_w, tv_wo,
                       float(0.5 / model.bound), _p(seed), M, 3, 1, self.Lv, self.S, self.H0, e1.gridtype_id, int(bool(e1.align_corners)),
                       _p(ws_tv), ws_tv.numel(), s)
            if spec_reg:
                extra = (w["spec_partial"], float(opt.lambda_specular / M))
            if lam_eik > 0:
                extra2 = (sb["eik"], (M + 255) // 256, float(lam_eik / M))
        else:
            self.g1.zero_()
            self.g2.zero_()
            sb["d_var"].zero_()
        self._marker = torch.cuda.Event()
        self._marker.record()
        self._lr_step(True if shading != 0 else False, loss_out=(N, b.loss, extra, extra2))
        loss = b.loss.view(())
        self._fill_pipeline()
        return loss

    def _lr_step(self, full, loss_out=None, fused=None):
        if full is not None:
            self._optimizer_step(full, lr_lambda(self.global_step - 1, self.opt.iters), loss_out, fused)
        if self.ema is not None and self.global_step % self.epoch_len == 0:      # end of an epoch (nerf/utils.py:1213-1214)
            self.ema_update()

    def ema_update(self):
        """One update of the averaged weights from the current parameters (one launch, stream-ordered behind the optimizer pass)."""
        if self.shard:
            self.sync_parameters()          # each rank has advanced its own rows of the fp32 tables only (a collective, at a symmetric step)
        self.ema.update()

    def averaged_parameters(self):
        """Context: the model carries the EMA weights (store / copy_to ... restore, nerf/utils.py:1250-1252,1340-1341); no-op without EMA."""
        import contextlib
        if self.ema is None:
            return contextlib.nullcontext()
        self.sync_parameters()
        return self.ema.average_parameters()

    @torch.no_grad()
    def eval_psnr(self, cam=0, downscale=4, use_ema=False):
        """use_ema: evaluate the averaged weights, as the reference's evaluate_one_epoch does (nerf/utils.py:1250-1252)."""
        from .trainer import Stage0Trainer
        self.sync_parameters()
        if use_ema and self.ema is not None:
            with self.averaged_parameters():
                return Stage0Trainer.eval_psnr(self, cam, downscale)
        return Stage0Trainer.eval_psnr(self, cam, downscale)
