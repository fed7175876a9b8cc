"""Where does the training step synchronise the host with the GPU?  torch's sync debug mode warns at every blocking call."""
import os
import sys
import warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from merlot_amd import NeatConfig  # noqa: E402
from merlot_amd.train import Trainer, synthetic_batch  # noqa: E402

dev = torch.device('cuda', 0)
config = NeatConfig.from_yaml(os.path.join(ROOT, 'merlot_amd', 'configs', 'pretrain_4seg_224.yaml'))
trainer = Trainer(config, dev, None, seed=0)
batch = synthetic_batch(config, 8, dev, seed=1)
trainer.step(batch)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode('warn')
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    trainer.step(batch)
torch.cuda.set_sync_debug_mode('default')
import traceback  # noqa: E402
print(len(w), 'synchronising calls in one step')
seen = {}
for x in w:
    key = f'{x.filename.replace(ROOT, "")}:{x.lineno}'
    seen[key] = seen.get(key, 0) + 1
for k, v in seen.items():
    print(v, k)
