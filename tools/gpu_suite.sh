# The whole GPU suite + the smoke entry, as the driver runs them at round end (run on the GPU box from the repo root: gpurun -- bash tools/gpu_suite.sh [tag])
TAG=${1:-suite}; mkdir -p gpurun_out/$TAG
timeout 2400 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc $?"; tail -25 gpurun_out/$TAG/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
