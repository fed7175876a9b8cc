"""bench.py on another build of the PRODUCT library: AB_LIB=<path to .so> python scripts/bench_lib.py [bench.py arguments] -- a same-box
A/B of two builds judged by the whole training step (boxes of the pool differ by +-2 %)."""
import os
import runpy
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from merlot_amd import lib  # noqa: E402
if os.environ.get('AB_LIB'):
    lib.LIB.path = os.path.abspath(os.environ['AB_LIB'])
    lib.LIB.check_abi = False
sys.argv[0] = os.path.join(ROOT, 'bench.py')
runpy.run_path(sys.argv[0], run_name='__main__')
