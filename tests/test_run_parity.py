"""RUN-LEVEL parity (north_star: "final PSNR within 0.1 dB of the reference"): whole training runs of the step executor against the reference's
OWN loop -- the unchanged Trainer.train_step / post_train_step + torch.optim.Adam + GradScaler + LambdaLR + EMA in train_one_epoch's order over
backends/_*.py (tests/run_parity.py) -- from the same initial parameters, evaluated on held-out views at full resolution through one
inference path, EMA weights (what the reference evaluates) and raw weights.

What the bar can be is set by the reference itself: two runs of the REFERENCE loop from the same initial state that differ only in their
random draws (pixels, backgrounds, jitter) end this far apart (profiles/r06_run_parity.txt, measured on MI355X):

    lego,   2 000 steps:  rms 1.5 dB (raw) / 2.3 dB (EMA)   -- the run is still recovering from the switch to full shading at step 1 000
    lego,   5 000 steps:  rms 0.09 dB (raw) / 0.12 dB (EMA)
    lego,  30 000 steps:  rms 0.27 dB (raw) / 0.16 dB (EMA)  -- the reference's full schedule
    garden, 2 000 steps:  rms 0.02 dB
    sdf,    2 000 steps:  rms 0.12 dB

and the executor necessarily draws differently from the reference loop (one [N,6] draw per batch instead of torch's global generator), so
|executor - reference| is compared with THAT spread: at the full 30 000 steps the paired difference over 6 seeds is -0.004 +- 0.058 dB (EMA),
-0.02 +- 0.06 dB (raw); at 5 000 steps over 4 seeds -0.04 +- 0.10 dB (EMA), -0.06 +- 0.07 dB (raw); garden at 30 000 steps -0.001 +- 0.004 dB.  The executor itself is bit-reproducible: two runs from one state end in identical bits (spread 0, asserted).

The driver-visible test below runs the lego recipe at 5 000 steps on 2 seeds (~2.5 minutes: the reference loop takes 9-10 ms per step) and
holds the mean difference to 0.3 dB = 0.1 dB + twice the standard error two seeds leave (0.1 dB per seed rms / sqrt(2) ~ 0.07);
tools/run_parity.py runs the larger tables."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_executor_run_ends_where_the_reference_loop_ends():
    import run_parity as RP_
    from oracle import ref_python
    if not ref_python.available():
        pytest.skip("reference Python not available (neither /root/reference nor oracle/_ref/pyref)")
    res = RP_.run("lego", seeds=2, steps=5000, views=4, log=lambda *_: None, engine_repeat=False)
    print("\n" + RP_.format_table(res))
    for kind in ("psnr_ema", "psnr_raw"):
        d = res["summary"][f"delta_{kind}"]
        assert abs(d["mean"]) <= 0.3, (kind, d)
        assert max(abs(x) for x in d["per_seed"]) <= 0.5, (kind, d)           # no single run falls out (a recipe-level bug costs dB, not tenths)
        assert res["summary"][f"engine_{kind}"]["mean"] >= 38.0, res["summary"]      # measured: 39.3-39.5 dB on both sides


def test_executor_garden_run_ends_where_the_reference_loop_ends():
    """BASELINE config 4's recipe (5 cascades, dt_gamma 1/256, per-camera near / far, entropy term, inner / outer TV): 2 000 steps, 2 seeds.
    Measured (profiles/r06_run_parity.txt): -0.007 +- 0.009 dB at 2 000 steps, -0.001 +- 0.004 dB at the full 30 000; two reference runs with
    different draws: 0.014-0.019 dB rms."""
    import run_parity as RP_
    from oracle import ref_python
    if not ref_python.available():
        pytest.skip("reference Python not available (neither /root/reference nor oracle/_ref/pyref)")
    res = RP_.run("garden", seeds=2, steps=2000, views=4, log=lambda *_: None)
    print("\n" + RP_.format_table(res))
    for kind in ("psnr_ema", "psnr_raw"):
        d = res["summary"][f"delta_{kind}"]
        assert abs(d["mean"]) <= 0.1, (kind, d)
        assert max(abs(x) for x in d["per_seed"]) <= 0.15, (kind, d)


def test_executor_runs_are_bit_reproducible():
    """The spread between two executor runs from one state is ZERO: every kernel of the step sums in a fixed order or in fixed point
    (DESIGN 4.4), the batches come from a seeded generator.  (tests/test_psnr_floor.py's old 1.2 dB window was justified by a 0.5 dB
    run-to-run spread that no longer exists.)"""
    import run_parity as RP_
    from nerf2mesh_amd import synthetic
    dev = torch.device("cuda", 0)
    poses = synthetic.make_cameras(100, seed=0).to(dev).float().contiguous()
    init = RP_.initial_state("lego", 0, 1200, dev)
    a, sa, _ = RP_.train_engine("lego", 0, 1200, poses, init, dev)       # crosses the occupancy refreshes, the EMA updates and the shading switch at 1 000
    b, sb, _ = RP_.train_engine("lego", 0, 1200, poses, init, dev)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    for k in sa:
        assert torch.equal(sa[k], sb[k]), "EMA " + k
