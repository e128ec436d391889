mkdir -p gpurun_out/r6ai
for v in 0 1 1 0; do
  N2M_TV_FWD=$v python bench.py --gpus 1 --no-cpu-baseline --steps 192 --warmup 16 --no-other-configs > gpurun_out/r6ai/b.json 2>gpurun_out/r6ai/b.err
  python -c "
import json; d=json.load(open('gpurun_out/r6ai/b.json')); print('N2M_TV_FWD=$v  192 steps (12 refreshes): %.4f ms/step = %.1f M samples/s' % (d['ms_per_step'], d['value']/1e6))"
done 2>&1 | tee gpurun_out/r6ai/tvfwd_long_run.txt
