"""The multi-rank training step on real device code: two ranks share GPU 0 and sum their gradients through gloo
(N2M_DIST_BACKEND=gloo; on the 8-GPU node the same code runs one rank per GPU over RCCL).  Covers the FusedAdamAMP multi-rank path:
per-table collectives in their own dtype, the persistent dW buffer + found_inf flag in the flat bucket, 1/world folded into the
seed gradient, lock-step GradScaler decisions."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("driver", ["trainer", "engine"])
def test_two_ranks_stay_bit_identical_and_learn(driver):
    env = dict(os.environ, N2M_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517" if driver == "trainer" else "29519", os.path.join(ROOT, "tools", "dist_check.py"), "40", driver]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIST_CHECK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_sharded_optimizer_equals_the_all_reduce_path(tmp_path):
    """Step executor, two ranks: reduce-scattered gradient rows + Adam on the own half of the rows + all-gather of the packed table
    (N2M_SHARD_ADAM, the default) against the all-reduce path.  The arithmetic is the same (with two ranks a + b is the same sum either
    way, the TV stencil reads the same density values from the packed table); what differs run to run is the order of the fp16 / fp32
    atomics on the split dense levels, which Adam amplifies -- so the yardstick is the distance between two runs of the all-reduce path."""
    import torch
    dumps = {}
    for name, shard, port in (("shard", "1", "29523"), ("ar1", "0", "29525"), ("ar2", "0", "29527")):
        path = str(tmp_path / f"{name}.pt")
        env = dict(os.environ, N2M_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", N2M_SHARD_ADAM=shard, N2M_DIST_DUMP=path)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.join(ROOT, "tools", "dist_check.py"), "24", "engine"]
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "DIST_CHECK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("DIST_CHECK")][-1]
        assert f"shard={shard == '1'}" in line, line
        dumps[name] = torch.load(path)
    dist = lambda a, b: float((dumps[a] - dumps[b]).norm() / dumps[b].norm())
    d_rr, d_s1, d_s2 = dist("ar1", "ar2"), dist("shard", "ar1"), dist("shard", "ar2")
    print(f"relative distance: all-reduce run vs run {d_rr:.3e}, sharded vs all-reduce {d_s1:.3e} / {d_s2:.3e}")
    # 24 steps include the occupancy refresh at step 16 (the sharded path gathers the density table for it: without that the
    # distance is 1.2e-2 against ~9e-4 run to run)
    assert max(d_s1, d_s2) <= 2.5 * d_rr + 3e-4


@pytest.mark.gpu
@pytest.mark.parametrize("recipe", ["lego", "garden"])
def test_sharded_occupancy_refresh_leaves_the_replicated_bit_field(tmp_path, recipe):
    """the 128^3 x cascade density query of the occupancy refresh dealt to the ranks by Morton range + an all-gather of the
    densities (renderer.update_extra_state, N2M_SHARD_REFRESH, the default) against every rank querying every cell (=0).  Two ranks on the one
    GPU, 20 steps (refreshes in front of steps 1 and 17): density grid and bit field bit for bit, and the parameters with them -- lego (1 cascade, all
    valid cells in one list) and the outdoor recipe (5 cascades, untrained cells excluded: 37 % of the 10.5 M cells are queried)."""
    import torch
    out = {}
    for name, flag, port in (("shard", "1", 29561 if recipe == "lego" else 29565), ("repl", "0", 29563 if recipe == "lego" else 29567)):
        g, d = str(tmp_path / f"{name}_grid.pt"), str(tmp_path / f"{name}.pt")
        env = {"N2M_SHARD_REFRESH": flag, "N2M_DIST_DUMP_GRID": g}
        if recipe == "garden":
            env["N2M_DIST_RECIPE"] = "garden"
        line = _dist_check(port, 20, env, dump=d)
        assert f"refresh_sharded={flag == '1'}" in line, line
        out[name] = (torch.load(g), torch.load(d))
    (ga, pa), (gb, pb) = out["shard"], out["repl"]
    assert torch.equal(ga["bits"], gb["bits"]), f"{int((ga['bits'] != gb['bits']).sum())} differing bytes of the occupancy bit field"
    assert torch.equal(ga["grid"], gb["grid"])
    assert int(ga["bits"].count_nonzero()) > 0
    assert torch.equal(pa, pb), "parameters after 20 steps differ between the sharded and the replicated refresh"


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["touch", "mark"])
def test_sharded_refresh_decides_symmetrically_after_a_rank_local_grid_write(mode):
    """a torch-side write to density_grid on ONE rank used to make that rank alone enter the list check's collective (a hang
    over RCCL).  The check now runs on every rank at every refresh.  Rank 1 alone rewrites its grid after step 10 -- "touch": same values, new
    version counter (the sharded query stays on); "mark": one empty cell set to -1, so the ranks' lists differ and EVERY rank takes the
    replicated query from then on.  Either way the run ends, and the replicas stay bit-identical (DIST_CHECK OK)."""
    line = _dist_check(29571 if mode == "touch" else 29573, 20, {"N2M_SHARD_REFRESH": "1", "N2M_DIST_ASYM": mode})
    assert f"refresh_sharded={mode == 'touch'}" in line, line


def _dist_check(port, steps, extra_env, dump=None, timeout=400):
    env = dict(os.environ, N2M_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", **extra_env)
    if dump:
        env["N2M_DIST_DUMP"] = dump
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "dist_check.py"), str(steps), "engine"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0 and "DIST_CHECK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    return [l for l in r.stdout.splitlines() if l.startswith("DIST_CHECK")][0]


@pytest.mark.gpu
def test_peer_store_exchange_equals_the_collective_path(tmp_path):
    """N2M_PEER_STORE=1 (include/n2m_peer.h, parallel.PeerExchange): two PROCESSES on the one GPU map each other's staging buffers
    (hipIpcGetMemHandle / hipIpcOpenMemHandle), the table backward's flush stores gradient rows into their owner's slots, the owner sums
    the slots in rank order, Adam's packed rows are stored into both ranks' tables, epoch flags order it all.  24 steps (one occupancy
    refresh inside): replicas bit-identical (DIST_CHECK OK), no wait timed out, and the parameters are as close to the reduce-scatter /
    all-gather run as two of those runs are to each other -- with two ranks: identical bits (the peer path sums fp16 slots in fp32 and rounds
    once, gloo rounds per hop; for two addends that is the same number)."""
    import torch
    dumps = {}
    for name, port, env in (("rs1", 29551, {"N2M_SHARD_ADAM": "1"}), ("rs2", 29553, {"N2M_SHARD_ADAM": "1"}),
                            ("peer", 29555, {"N2M_SHARD_ADAM": "1", "N2M_PEER_STORE": "1"}),
                            ("peer_unfused", 29569, {"N2M_SHARD_ADAM": "1", "N2M_PEER_STORE": "1", "N2M_PEER_FUSED": "0"}),
                            # round 6: coarse chunks PADDED to a multiple of four rows, the last rank's shorter -- the layout W = 4 / W = 8 get
                            # at the standard table (split / W = 481 390 / 240 695 rows), forced here on two ranks by 52 rows of extra padding
                            ("peer_padded", 29575, {"N2M_SHARD_ADAM": "1", "N2M_PEER_STORE": "1", "N2M_PEER_PAD_ROWS": "52"})):
        path = str(tmp_path / f"{name}.pt")
        line = _dist_check(port, 24, env, dump=path)
        assert f"peer_store={name.startswith('peer')}" in line and "shard=True" in line, line
        dumps[name] = torch.load(path)
    dist = lambda a, b: float((dumps[a] - dumps[b]).norm() / dumps[b].norm())
    d_rr, d_p1, d_p2 = dist("rs1", "rs2"), dist("peer", "rs1"), dist("peer", "rs2")
    print(f"relative distance: collective run vs run {d_rr:.3e}, peer-store vs collective {d_p1:.3e} / {d_p2:.3e}")
    assert max(d_p1, d_p2) <= 2.5 * d_rr + 3e-4
    # round 5: the slot sum inside Adam's gradient load and the row push inside its packed-row store (n2m_adam_step_peer, the default in peer
    # mode) against the separate passes (n2m_peer_reduce_slices -> n2m_adam_step -> n2m_peer_copy): the same arithmetic, bit for bit
    assert torch.equal(dumps["peer"], dumps["peer_unfused"])
    # ... and where the chunk boundaries lie changes nothing either (rank 0 owns 52 rows more, rank 1 104 fewer than its slot holds)
    assert torch.equal(dumps["peer"], dumps["peer_padded"])
    if d_rr == 0.0:
        # the step is bit-reproducible (fixed-point table backward) and a sum of TWO ranks does not depend on its order or on where the fp16
        # rounding happens: the peer-store run must then reproduce the collective run bit for bit (measured: it does)
        assert d_p1 == 0.0 and d_p2 == 0.0


@pytest.mark.gpu
def test_peer_store_with_a_rank_without_samples():
    """The peer-store exchange with rank 1 looking away from the scene: it stores zeros into its slots and signals, takes the other rank's
    rows, and the replicas stay bit-identical -- no hang, no timeout."""
    _dist_check(29557, 24, {"N2M_SHARD_ADAM": "1", "N2M_PEER_STORE": "1", "N2M_DIST_BLIND_RANK": "1"})


@pytest.mark.gpu
def test_peer_store_primitives_and_timeout():
    """n2m_peer_* between two processes on one GPU (tools/peer_check.py): a buffer one process allocates is written by the other through
    the mapped pointer; signal / wait hand over in both directions; the slot sum is the rank-order sum; a wait for a signal that never
    comes returns after its timeout and leaves the error word -- the stream, and the GPU, go on."""
    env = dict(os.environ, N2M_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29559", os.path.join(ROOT, "tools", "peer_check.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "PEER_CHECK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("driver", ["trainer", "engine"])
def test_stage1_two_ranks_views_sharded(driver):
    """Stage 1 on two ranks (views shard), autograd trainer and step executor: the fused AMP optimizer stays on, replicas
    bit-identical, the per-face error accumulators summed over the ranks before a refinement, rank 0's new mesh taken over by everybody
    (tools/dist_check_stage1.py)."""
    env = dict(os.environ, N2M_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", {"trainer": "29541", "engine": "29543"}[driver], os.path.join(ROOT, "tools", "dist_check_stage1.py"), "6", driver]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIST_CHECK_S1 OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("shard", ["1", "0"])
def test_a_rank_without_samples_keeps_the_collectives_in_step(shard):
    """Rank 1 looks away from the scene: all of its batches are empty.  The step executor must neither hang nor let the replicas drift
    apart -- the empty rank issues the same reduce-scatters / all-reduces in the same order and applies the other rank's gradients
    (both optimizer layouts of the executor: sharded and all-reduce; the autograd trainer refuses such a batch loudly in render())."""
    driver = "engine"
    env = dict(os.environ, N2M_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", N2M_SHARD_ADAM=shard, N2M_DIST_BLIND_RANK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", {"1": "29531", "0": "29533"}[shard], os.path.join(ROOT, "tools", "dist_check.py"), "24", driver]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "DIST_CHECK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_bench_spawns_the_ranks_it_is_asked_for():
    """`python bench.py --gpus 2` with no launcher around it re-executes itself under torch.distributed.run: the JSON line says
    n_gpus = 2 and names the backend the ranks really used (here gloo, two ranks on the one GPU of the test box)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(N2M_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--pretrain", "20",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "gloo" in d["config"]["parallelism"]
    assert d["config"]["samples_per_step_per_gpu"] > 1e5 and d["value"] > 0
    # rank 0 evaluates a view after the other ranks have left: the sharded tables were gathered while everybody was still there
    assert isinstance(d["psnr_view0_quarter_res"], float), d["psnr_view0_quarter_res"]


@pytest.mark.gpu
def test_bench_refuses_a_rank_count_it_cannot_run():
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "N2M_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


def _needs_two_gpus():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: the RCCL path (backend nccl, one rank per GPU) cannot run on a one-GPU box")


@pytest.mark.gpu
@pytest.mark.parametrize("shard", ["1", "0"])
def test_rccl_two_gpus_replicas_stay_identical(shard):
    """The first multi-GPU lease exercises RCCL straight away: one rank per GPU, backend nccl (= RCCL), the step executor with the sharded
    optimizer (reduce_scatter_tensor / all_gather_into_tensor in place) and with the all-reduce layout.  Skips on one-GPU boxes."""
    _needs_two_gpus()
    env = {k: v for k, v in os.environ.items() if k != "N2M_DIST_BACKEND"}
    env.update(MASTER_ADDR="127.0.0.1", N2M_SHARD_ADAM=shard, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", {"1": "29541", "0": "29543"}[shard], os.path.join(ROOT, "tools", "dist_check.py"), "40", "engine"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "DIST_CHECK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DIST_CHECK")][-1]
    assert "backend=nccl" in line and f"shard={shard == '1'}" in line, line


@pytest.mark.gpu
def test_rccl_two_gpus_bench_line():
    """`python bench.py --gpus 2` on a box with two GPUs: n_gpus = 2 over nccl, weak scaling, finite PSNR."""
    _needs_two_gpus()
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "N2M_DIST_BACKEND")}
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--pretrain", "100",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "nccl" in d["config"]["parallelism"] and d["value"] > 0
    assert isinstance(d["psnr_view0_quarter_res"], float), d["psnr_view0_quarter_res"]


@pytest.mark.gpu
def test_sharded_checkpoint_holds_complete_optimizer_state(tmp_path):
    """with the optimizer sharded over the ranks each rank advances the Adam moments of its own rows only;
    FusedAdamAMP.state_dict() of such a run gathers them first (collective), so rank 0's checkpoint equals rank 1's and a resumed run
    continues with the moments of ALL rows (the reference saves complete optimizer state, nerf/utils.py:1336-1350)."""
    env = dict(os.environ, N2M_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", N2M_SHARD_ADAM="1", N2M_DIST_CKPT=str(tmp_path / "ck"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29545", os.path.join(ROOT, "tools", "dist_check.py"), "20", "engine"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIST_CHECK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    import torch
    a, b = torch.load(str(tmp_path / "ck.rank0.pt")), torch.load(str(tmp_path / "ck.rank1.pt"))
    for k in a:
        assert torch.equal(a[k], b[k]), f"rank 0 and rank 1 saved different {k}"
    # every row that has a parameter change also has moments: nothing of the other rank's shard was left at zero
    for name in ("exp_avg_sq.0", "exp_avg_sq.1"):
        nz = (a[name].reshape(a[name].shape[0], -1).abs().sum(-1) > 0)
        halves = nz.view(-1)
        lo, hi = halves[:halves.numel() // 2].float().mean().item(), halves[halves.numel() // 2:].float().mean().item()
        assert lo > 0 and hi > 0, (name, lo, hi)
