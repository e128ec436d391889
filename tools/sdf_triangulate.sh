# SDF run parity, which component carries the executor's +0.28 dB: the reference's loop with the fused field behind it, and the autograd trainer on
# the library's kernels, 16 seeds x 3000 steps each (the reference loop itself: profiles/r06_run_parity.txt, same seeds)
mkdir -p gpurun_out/r6y
for s in 0 8; do
  timeout 700 python tools/run_parity.py --recipe sdf --seeds 8 --first-seed $s --steps 3000 --views 8 --paths reference-fused,trainer --out gpurun_out/r6y/sdf3k_tri_$s > gpurun_out/r6y/log_$s.txt 2>&1
  tail -8 gpurun_out/r6y/log_$s.txt
done
