"""Run-level parity of the step executor against the reference's own training loop: forwards to tests/run_parity.py (the checker lives
under tests/ because it imports the reference's Python through oracle/ref_python.py -- test infrastructure).

    python tools/run_parity.py --recipe lego --seeds 3 --steps 2000 --views 8 --engine-repeat --out profiles/r06_run_parity_lego
"""
import os
import runpy
import sys

if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    runpy.run_path(os.path.join(root, "tests", "run_parity.py"), run_name="__main__")
