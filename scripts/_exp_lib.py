"""scripts/ only: route merlot_amd through the EXPERIMENTS build of the library (libmerlot_hip_exp.so, compiled with
-DMERLOT_EXPERIMENTS by `merlot_amd/csrc/build.sh exp`), the only build in which MERLOT_DBG / MERLOT_NT_CFG_DYN /
MERLOT_NT_TILE_CG_DYN / MERLOT_TN_CFG / MERLOT_TN_SPLITS are read, and expose the probe library.  Import it first."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
from merlot_amd import lib  # noqa: E402

EXP = os.environ.get('EXP_LIB') or os.path.join(lib._HERE, 'libmerlot_hip_exp.so')
assert os.path.exists(EXP), "experiments build missing: run merlot_amd/csrc/build.sh exp"
lib.LIB.path = EXP
lib.LIB.protos['merlot_probe_persist_trace'] = ('int', [('void*', 'dst'), ('int64_t', 'bytes'), ('merlot_stream_t', 'stream')])
from probe_lib import PROBE  # noqa: E402,F401
