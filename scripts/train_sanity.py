"""End-to-end sanity on the GPU: overfit one small synthetic batch for a few dozen steps with the full pipeline
(forward, backward into the arena, fused AdamW with bf16 state, bf16 weight refresh) -- the loss must fall."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
from merlot_amd import NeatConfig
from merlot_amd.train import Trainer, synthetic_batch

cfgd = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), '..', 'merlot_amd', 'configs', 'pretrain_4seg_224.yaml')))
cfgd['optimizer'].update(num_warmup_steps=5, num_train_steps=max(200, 2 * (int(sys.argv[4]) if len(sys.argv) > 4 else 60)), learning_rate=2e-4)
cfgd['model']['hidden_dropout_prob'] = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
if len(sys.argv) > 2 and sys.argv[2] != 'bf16':           # round 6: python scripts/train_sanity.py 0.1 w1,w2,fuse,noa,dgrad1 [examples] -- fp8 forward + the 8-bit backward
    cfgd['model']['fp8_forward'] = 'ln'
    cfgd['model']['fp8_backward'] = sys.argv[2]
    print('fp8_forward ln, fp8_backward', sys.argv[2])
config = NeatConfig.from_dict(cfgd)
tr = Trainer(config, 'cuda', None, seed=0)
batch = synthetic_batch(config, int(sys.argv[3]) if len(sys.argv) > 3 else 2, 'cuda', seed=7)
hist = []
NSTEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 60
for it in range(NSTEPS):
    out = tr.step(batch)
    if it % (5 if NSTEPS <= 60 else 20) == 0 or it == NSTEPS - 1:
        m = out['metrics']
        print(f"step {it:3d} loss {float(out['loss']):8.4f}  mlm {float(m['lang/loss']):7.4f} acc {float(m['lang/acc']):.3f}  "
              f"contr {float(m['contr/loss_all']):.4f}  temporal {float(m['temporal/loss']):.4f}  lr {tr.opt.current_lr():.2e}", flush=True)
    hist.append(float(out['loss']))
assert all(h == h for h in hist), "NaN in the loss"
print("first", hist[0], "last", hist[-1], "drop", hist[0] - hist[-1])
assert hist[-1] < hist[0] - 1.0, "loss did not fall"
print("TRAIN SANITY OK")
