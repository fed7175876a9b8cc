"""Which lines of the host code launch the torch (non-library) kernels of one training step, and what they cost: torch.profiler with stacks over ONE
step at the bench's batch; device time of every aten kernel, summed by the innermost merlot_amd/ source line on its Python stack."""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

from merlot_amd import NeatConfig  # noqa: E402
from merlot_amd.train import Trainer, synthetic_batch  # noqa: E402

EX = int(os.environ.get('EXAMPLES', 128))
config = NeatConfig.from_yaml(os.path.join(ROOT, 'merlot_amd', 'configs', 'pretrain_4seg_224.yaml'))
dev = torch.device('cuda', 0)
trainer = Trainer(config, dev, None, seed=0)
batch = synthetic_batch(config, EX, dev, seed=1234)
for _ in range(3):
    trainer.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    trainer.step(batch)
    torch.cuda.synchronize()
by_site = collections.defaultdict(lambda: [0.0, 0, collections.Counter()])
total = lib_total = 0.0
for ev in prof.events():
    dt = getattr(ev, 'device_time_total', 0) or getattr(ev, 'cuda_time_total', 0)
    if not ev.kernels:
        continue
    t = sum(k.duration for k in ev.kernels)
    site = None
    for fr in (ev.stack or []):
        if 'merlot_amd/' in fr and 'lib.py' not in fr:
            site = fr.split('merlot_amd/')[-1]
            break
    names = [k.name for k in ev.kernels]
    is_lib = any('anonymous namespace' in n or '_GLOBAL__N' in n for n in names)
    total += t
    if is_lib:
        lib_total += t
        continue
    d = by_site[site or '(no merlot_amd frame)']
    d[0] += t
    d[1] += len(names)
    d[2][ev.name] += 1
print(f'one step: {total / 1e3:.1f} ms of kernels, library {lib_total / 1e3:.1f} ms, torch {(total - lib_total) / 1e3:.2f} ms')
for site, (t, n, ops_) in sorted(by_site.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f'{t:9.1f} us {n:4d} kernels  {site}   {dict(ops_.most_common(3))}')

print('--- aten ops by input shape (device time of their own kernels)')
rows = []
for ka in prof.key_averages(group_by_input_shape=True):
    t = getattr(ka, 'self_device_time_total', None)
    if t is None:
        t = getattr(ka, 'self_cuda_time_total', 0)
    if ka.key.startswith('aten::') and t > 20:
        rows.append((t, ka.count, ka.key, str(ka.input_shapes)[:150]))
for t, n, k, shp in sorted(rows, reverse=True)[:45]:
    print(f'{t:9.1f} us {n:4d} x {k:28s} {shp}')
