mkdir -p gpurun_out/r6ad
timeout 900 python -m pytest tests/test_optim.py tests/test_adam_tail.py -x -q -m gpu > gpurun_out/r6ad/tests.log 2>&1; tail -6 gpurun_out/r6ad/tests.log
bash tools/env_ab.sh N2M_ADAM_TAIL 0 1 r6ad/tail 2>&1 | tee gpurun_out/r6ad/adam_tail_ab.txt
