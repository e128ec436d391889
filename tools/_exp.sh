cd /root/repo; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do python bench.py --no-cpu-baseline > gpurun_out/bench49_$i.json 2>/dev/null; python tools/show_bench.py gpurun_out/bench49_$i.json | head -1; done; python tools/train_check.py 2>&1 | tail -1
