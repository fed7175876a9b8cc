"""Shared by tests/test_native_yaml_gpu.py and scripts/exp_native_shapes.py: the two problems of tests/golden/ref_shim_native.npz
(model/configs/merlot.yaml:30,36 -- a non-square frame on the patch stem; 192 x 352 with the ResNet-hybrid stem at its released depth
[3, 4, 9]) run through the HIP model, with everything the comparisons need."""
import os

import numpy as np
import torch

from common import tiny_config, synth_batch
from oracle import merlot_oracle as mo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
PROBLEMS = {'p64x96': dict(over=dict(image_size=[64, 96]), E=2),
            'r192x352': dict(over=dict(image_size=[192, 352], resnet_layers=[3, 4, 9]), E=1)}


def native_weights(cfg, seed):
    """oracle.init_weights, and for a ResNet-hybrid stem the gamma of every bottleneck block's LAST GroupNorm (GroupNorm_3, _6, ...:
    the one in front of the residual add) scaled by 0.2.  At depth [3, 4, 9] a randomly initialised stem is chaotic in bf16: the
    reference's own bf16 policy moves its output by 36 % against fp32 (3.7 % at depth [1, 1, 2]), so no comparison of a bf16 path with
    anything means much; with the residual branches damped -- what a trained network looks like, and what `zero-init residual` does on
    purpose -- the same measure is 3.2 %, and every one of the 164 stem variables still acts on the output."""
    import re
    w = mo.init_weights(cfg, seed=seed, perturb=True)
    if cfg.get('resnet_layers'):
        for n in w:
            m = re.search(r'block_group\d/GroupNorm_(\d+)/gamma$', n)
            if m and int(m.group(1)) % 3 == 0 and int(m.group(1)) > 0:
                w[n] = w[n] * 0.2
    return w


def is_stem(n):
    return 'resnet50lite' in n or 'conv_postresnet_proj' in n


def load(name):
    fx = np.load(os.path.join(GOLD, 'ref_shim_native.npz'), allow_pickle=False)
    p = name + '/'
    spec = PROBLEMS[name]
    cfg = tiny_config(**spec['over'])
    assert list(fx[p + 'image_size']) == cfg['image_size']
    wseed, bseed = (int(v) for v in fx[p + 'seeds'])
    batch = synth_batch(cfg, E=spec['E'], num_chunks=4, Lc=32, seed=bseed)
    w = native_weights(cfg, wseed)
    noise = {k: fx[p + 'noise/' + k] for k in ('gumbel', 'span_lower', 'span_upper', 'random_ids', 'option')}
    return cfg, batch, w, noise, {k[len(p):]: fx[k] for k in fx.files if k.startswith(p)}


def run_hip(cfg, batch, w, noise, backward=True):
    from merlot_amd import MerlotModel, ParamStore
    st = ParamStore(cfg, 'cuda', seed=0)
    st.load_tf_weights({k: v.detach() for k, v in w.items()})
    st.zero_grad()
    sidx = torch.from_numpy(batch['shuffled_idx_img']).cuda()
    pm = MerlotModel(cfg, True, False, batch['image'].cuda(), batch['input_ids'].cuda(), mask_input=True, shuffled_idx_img=sidx, params=st,
                     noise={k: torch.from_numpy(np.asarray(v)) for k, v in noise.items()})
    l1, i1 = pm.mask_loss()
    l2, i2 = pm.contrastive_loss()
    l3, i3 = pm.temporal_loss(sidx, torch.from_numpy(batch['video_src_ids']).cuda())
    grads = None
    if backward:
        (l1 + l2 + l3).backward()
        torch.cuda.synchronize()
        grads = {k: v.detach().float().cpu() for k, v in st.export_tf_grads().items()}
    return pm, (float(l1.detach()), float(i2['loss_all'].detach()), float(l3.detach())), grads


def run_oracle(cfg, batch, w, noise, bf16_stem):
    wo = {k: v.detach().clone().requires_grad_(True) for k, v in w.items()}
    ctx = mo.bf16_stem() if bf16_stem else torch.enable_grad()
    with ctx:
        m = mo.MerlotOracle(cfg, wo, batch['image'], batch['input_ids'], mask_input=True, shuffled_idx_img=batch['shuffled_idx_img'], noise=noise)
        loss, info = m.total_loss(batch['shuffled_idx_img'], batch['video_src_ids'])
        loss.backward()
    return m, float(loss), {k: v.grad.detach().float() for k, v in wo.items() if v.grad is not None}
