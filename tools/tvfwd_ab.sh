mkdir -p gpurun_out/r6s
timeout 600 python -m pytest tests/test_tv_fwd.py tests/test_tv_corners.py -x -q -m gpu > gpurun_out/r6s/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r6s/tests.log
tail -15 gpurun_out/r6s/tests.log
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "tv or TV or total_variation or backward" > gpurun_out/r6s/tests2.log 2>&1; echo "tests2 rc=$?" >> gpurun_out/r6s/tests2.log
tail -5 gpurun_out/r6s/tests2.log
timeout 900 bash tools/env_ab.sh N2M_TV_FWD 0 1 r6s/tvfwd 2>&1 | tee gpurun_out/r6s/tvfwd_ab.txt
