"""How tight can the stem-gradient parity check be?  rel-L2 of the 56 stem gradients (HIP vs bf16-policy oracle) at several frame sizes."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch
from common import tiny_config, synth_batch, rel_l2
from oracle import merlot_oracle as mo
from merlot_amd import MerlotModel, ParamStore

for size in (64, 128, 224):
    cfg = tiny_config(resnet_layers=[1, 1, 2], image_size=[size, size])
    w = mo.init_weights(cfg, 2)
    for t in w.values():
        t.requires_grad_(True)
    b = synth_batch(cfg, E=1, num_chunks=4, seed=4)
    with mo.bf16_stem():
        m = mo.MerlotOracle(cfg, w, b['image'], b['input_ids'], mask_input=False, shuffled_idx_img=b['shuffled_idx_img'])
    cot = torch.randn(m.encoder_hidden_states['viz'].shape, generator=torch.Generator().manual_seed(0))
    (m.encoder_hidden_states['viz'] * cot).sum().backward()
    st = ParamStore(cfg, 'cuda', seed=0)
    st.load_tf_weights({k: v.detach() for k, v in w.items()})
    st.zero_grad()
    pm = MerlotModel(cfg, True, False, b['image'].cuda(), b['input_ids'].cuda(), mask_input=False,
                     shuffled_idx_img=torch.from_numpy(b['shuffled_idx_img']).cuda(), params=st)
    (pm.encoder_hidden_states['viz'] * cot.cuda()).sum().backward()
    torch.cuda.synchronize()
    gt = st.export_tf_grads()
    ks = [k for k, v in w.items() if v.grad is not None and ('resnet50lite' in k or 'conv_postresnet_proj' in k)]
    rels = np.array([rel_l2(gt[k], w[k].grad) for k in ks])
    cos = np.array([float(torch.dot(gt[k].flatten().float().cpu(), w[k].grad.flatten()) / (gt[k].float().norm().cpu() * w[k].grad.norm() + 1e-30)) for k in ks])
    allg = torch.cat([gt[k].flatten().float().cpu() for k in ks]); allo = torch.cat([w[k].grad.flatten() for k in ks])
    print(f'{size}^2: {len(ks)} tensors rel-L2 max {rels.max():.3f} median {np.median(rels):.3f} | cosine min {cos.min():.4f} median {np.median(cos):.4f} | '
          f'all stem gradients as one vector rel-L2 {rel_l2(allg, allo):.3f} | fwd viz {rel_l2(pm.encoder_hidden_states["viz"], m.encoder_hidden_states["viz"]):.4f}', flush=True)
