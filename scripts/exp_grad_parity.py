"""Round 4: per-tensor gradient agreement of the HIP path with (a) the torch emulation of the SAME bf16 policy
(tests/emu_ops.py, CPU) and (b) the fp32 oracle, by tensor class -- the measurements behind the bounds of
tests/test_grad_classes_gpu.py.  python scripts/exp_grad_parity.py [config1 | config2x8]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from grad_parity import run_all, tensor_class  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'config1'
res = run_all(which, verbose=True)
rows = []
for n in sorted(res['hip']):
    if n.endswith('key_layer/bias') or n not in res['emu']:
        continue
    gh, ge, go = res['hip'][n].double(), res['emu'][n].double(), res['oracle'][n].double()
    rows.append((tensor_class(n), n, float((gh - ge).norm() / (ge.norm() + 1e-30)), float(gh.norm() / (ge.norm() + 1e-30)) - 1.0,
                 float((gh - go).norm() / (go.norm() + 1e-30)), float(gh.norm() / (go.norm() + 1e-30)) - 1.0,
                 float((ge - go).norm() / (go.norm() + 1e-30)), float(ge.norm())))
print(f"{'class':8s} {'tensor':70s} {'hip/emu relL2':>13s} {'norm-1':>9s} | {'hip/orc relL2':>13s} {'norm-1':>9s} | {'emu/orc':>9s} {'|g|':>9s}")
for r in sorted(rows):
    print(f"{r[0]:8s} {r[1][-70:]:70s} {r[2]:13.2e} {r[3]:9.1e} | {r[4]:13.2e} {r[5]:9.1e} | {r[6]:9.2e} {r[7]:9.2e}")
for c in sorted(set(r[0] for r in rows)):
    rr = [r for r in rows if r[0] == c]
    print(f"class {c:8s} n={len(rr):3d}  hip/emu relL2 max {max(r[2] for r in rr):.2e} med {np.median([r[2] for r in rr]):.2e}  |norm-1| max "
          f"{max(abs(r[3]) for r in rr):.1e}   hip/oracle relL2 max {max(r[4] for r in rr):.2e} med {np.median([r[4] for r in rr]):.2e} "
          f"|norm-1| max {max(abs(r[5]) for r in rr):.1e}")
print('losses hip / emu / oracle:', res['loss'])
