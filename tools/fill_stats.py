"""What the table backward's fill is fed in the steady state of the bench (after `--pretrain` steps of the lego recipe): per level, the share of
samples whose feature gradient is exactly zero (they deliver at most their TV term), the share of samples that start a same-cell run (what the
run merge of levels 0-8 keeps), and the log entries that follow.  Reads the step executor's own buffers after a step; no kernel is changed.

    python tools/fill_stats.py [--pretrain 1000]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pretrain", type=int, default=1000)
    args = ap.parse_args()
    from nerf2mesh_amd import synthetic
    from nerf2mesh_amd.engine import Stage0Engine
    from nerf2mesh_amd.network import NeRFNetwork
    from nerf2mesh_amd.options import make_options
    torch.manual_seed(0)
    opt = make_options(O=True, iters=30000, fused_mlp=True, bound=1, dt_gamma=0)
    dev = torch.device("cuda", 0)
    eng = Stage0Engine(NeRFNetwork(opt), opt, synthetic.make_cameras(100, seed=0), dev, seed=0)
    eng.mark_untrained()
    cap = {}
    orig = eng._finish

    def finish(b):
        M = orig(b)
        cap["b"], cap["M"] = b, M
        return M
    eng._finish = finish
    for _ in range(args.pretrain + 3):
        eng.train_step()
    torch.cuda.synchronize()
    b, M = cap["b"], cap["M"]
    w = eng._w
    stride = M                                             # level-major layouts use the step's own M as the stride
    d1 = w["d_h1"][:16 * stride].view(16, stride)[:, :M]
    d2 = w["d_h2"][:32 * stride].view(16, stride, 2)[:, :M]
    x = b.samples[:3 * b.cap_m][:3 * M].view(M, 3)
    x01 = (x + 1) * 0.5
    rays = b.rays[:b.N].cpu().numpy()
    print(f"M = {M} samples, N = {b.N} rays, {M / b.N:.1f} samples per ray")
    live = ((d1 != 0) | (d2.view(torch.int16).bitwise_and(0x7FFF) != 0).any(-1))          # [16, M]
    live_any = live.any(0)
    print(f"samples with a non-zero feature gradient on some level: {float(live_any.float().mean()):.3f}")
    # per ray: the first sample whose gradient is zero on every level -- how contiguous are the dead samples?
    la = live_any.cpu().numpy()
    dead_tail = 0
    for off, cnt in rays[:2000]:
        seg = la[off:off + cnt]
        k = len(seg)
        while k > 0 and not seg[k - 1]:
            k -= 1
        dead_tail += len(seg) - k
    tot = int(rays[:2000, 1].sum())
    print(f"first 2000 rays: dead samples {1 - la[:tot].mean():.3f} of all, of which in a ray's tail {dead_tail / max(1, (~la[:tot]).sum()):.3f}")
    S = float(np.log2(eng.model.encoder.per_level_scale))
    total_entries = 0
    for l in range(16):
        scale = 2.0 ** (l * S) * 16 - 1
        cell = torch.floor(x01 * scale + 0.5).to(torch.int32)
        same = (cell[1:] == cell[:-1]).all(-1)
        heads = 1 + int((~same).sum())
        lv = live[l]
        lf = float(lv.float().mean())
        # log entries: unmerged levels 8 per live sample + 1 (TV) per dead one; merged levels: per run head
        if l < 9:
            # a run delivers eight entries if any member is live, else one
            run_id = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), (~same).long().cumsum(0)])
            run_live = torch.zeros(heads, dtype=torch.bool, device=dev)
            run_live.index_put_((run_id,), lv, accumulate=False) if False else None
            rl = torch.zeros(heads, dtype=torch.int32, device=dev).index_add_(0, run_id, lv.int()) > 0
            ent = int(rl.sum()) * 8 + int((~rl).sum())
        else:
            ent = int(lv.sum()) * 8 + int((~lv).sum())
        total_entries += ent
        print(f"level {l:2d}: scale {scale:8.1f}  live {lf:.3f}  run heads {heads / M:.3f} of M  entries {ent / 1e6:.2f} M")
    print(f"log entries (estimate, 64-lane run limits ignored): {total_entries / 1e6:.1f} M = {total_entries * 10 / 1e6:.0f} MB")


if __name__ == "__main__":
    main()
