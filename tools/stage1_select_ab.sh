mkdir -p gpurun_out/r6w
timeout 300 python -m pytest tests/test_hip_parity.py -q -m gpu -k "select_positive or compact" > gpurun_out/r6w/t1.log 2>&1; tail -3 gpurun_out/r6w/t1.log
for v in torch n2m n2m torch torch n2m; do
  N2M_S1_NONZERO=$v python bench.py --stage 1 --no-cpu-baseline > gpurun_out/r6w/s1_$v.json 2>gpurun_out/r6w/s1.err
  python -c "
import json,sys; d=json.load(open('gpurun_out/r6w/s1_$v.json')); print('N2M_S1_NONZERO=$v', round(d['ms_per_step'],4), 'ms/step', round(d['value']/1e6,1), 'M px/s')"
done 2>&1 | tee gpurun_out/r6w/stage1_select_ab.txt
timeout 900 python -m pytest tests/test_stage1.py tests/test_stage1_reference.py tests/test_pipeline.py -q -m gpu > gpurun_out/r6w/t2.log 2>&1; tail -3 gpurun_out/r6w/t2.log
