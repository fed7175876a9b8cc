"""bench.py with the attention log taken from the forward (MERLOT_LOG_BWD=off: every joint-encoder forward launch re-walks Q K^T for the
four block sums) or from the backward (on, the Trainer's default since round 4): the same-box A/B of that change."""
import os
import runpy
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from merlot_amd import config as _cfg  # noqa: E402
_orig = _cfg.NeatConfig.from_yaml.__func__


def _from_yaml(cls, path, *a, **k):
    c = _orig(cls, path, *a, **k)
    c.model['attention_log_in_backward'] = os.environ.get('MERLOT_LOG_BWD', 'on') == 'on'
    return c


_cfg.NeatConfig.from_yaml = classmethod(_from_yaml)
sys.argv[0] = os.path.join(ROOT, 'bench.py')
runpy.run_path(sys.argv[0], run_name='__main__')
