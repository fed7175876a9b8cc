"""bench.py on the EXPERIMENTS build of the library (libmerlot_hip_exp.so): lets a switch of that build (MERLOT_P8_PH2, MERLOT_TN_PH2 ...)
be judged by what it does to the whole training step instead of to a tight loop of one launch -- the chip's clock depends on
what else runs (profiles/r03_a_ph2.txt).  Same arguments as bench.py; numbers are comparable with each other only."""
import _exp_lib  # noqa: F401
import os
import runpy
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv[0] = os.path.join(ROOT, 'bench.py')
runpy.run_path(sys.argv[0], run_name='__main__')
