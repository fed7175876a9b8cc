"""Shared by tests/test_grad_classes_gpu.py and scripts/exp_grad_parity.py: one problem, three runs --
  hip    : the product (HIP kernels, GPU)
  emu    : the product's HOST code on tests/emu_ops.py (torch-CPU, the same bf16 rounding points as the kernels)
  oracle : oracle/merlot_oracle.py (fp32, the reference's graph)
and every parameter gradient of each, keyed by the reference's variable names.

Why the middle run: bf16 compute against an fp32 reference differs by 3-12 % per tensor (rounding noise of a 2 + 2 + 2-layer
model), which hides a wrong scale on a small tensor (a bias, a LayerNorm gamma).  The emulation rounds where the kernels round, so
HIP vs emulation agrees an order of magnitude tighter and a scale error cannot hide; the emulation's host logic is pinned against the
oracle on the CPU (tests/test_host_emulated.py), and here again in the same process."""
import re

import numpy as np
import pytest
import torch

from common import tiny_config, synth_batch
from oracle import merlot_oracle as mo


def tensor_class(name):
    """bias | ln (LayerNorm gamma / beta) | pos (position / CLS tables) | emb (word embeddings) | kernel (dense / conv kernels)"""
    leaf = name.split('/')[-1]
    if leaf == 'bias' or leaf.endswith('_bias') or leaf == 'output_bias':
        return 'bias'
    if leaf in ('gamma', 'beta'):
        return 'ln'
    if 'word_embeddings' in name:
        return 'emb'
    if re.search(r'(pos_emb|position_embeddings|cls_emb|img_idx_pe|final_pe|pe$|embeddings$)', name):
        return 'pos'
    return 'kernel'


def problem(which):
    if which == 'config1':                                  # BASELINE config #1: 64^2, 2 + 2 + 2 layers, 2 examples x 4 segments
        cfg = tiny_config()
        return cfg, synth_batch(cfg)
    if which == 'config2x8':                                # config #2's geometry (224^2, groups of 4, 16 chunks), 8 examples, 2 + 2 + 2 layers
        # masking_use_attn off: with 32 groups a near-tie between the 25th and 26th attention sum of SOME group flips between the
        # bf16 and the fp32 run, and a different mask is a different problem (the attention-guided choice is compared where it is
        # stable: config #1, config #2 at one example, the reference-run fixtures); the masks here come from the explicit noise only
        cfg = tiny_config(image_size=[224, 224], masking_use_attn=False)
        return cfg, synth_batch(cfg, E=8, num_chunks=16, seed=5)
    if which == 'config2d12':                               # BASELINE config #2 at FULL depth (12 + 12 + 12 layers, 224^2, 16 chunks), one example:
        import os                                           # the problem of tests/test_config2_depth12_gpu.py (attention-guided masking on)
        from merlot_amd import NeatConfig
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        config = NeatConfig.from_yaml(os.path.join(root, 'merlot_amd', 'configs', 'pretrain_4seg_224.yaml'))
        cfg = dict(config.model)
        cfg['hidden_dropout_prob'] = 0.0
        return cfg, synth_batch(cfg, E=1, num_chunks=16, seed=11, two_videos=True)
    raise ValueError(which)


def _run_product(cfg, w, b, device):
    from merlot_amd import MerlotModel, ParamStore
    st = ParamStore(cfg, device, seed=0)
    st.load_tf_weights({k: v.detach() for k, v in w.items()})
    dev = lambda t: t.to(device)
    sidx = dev(torch.from_numpy(b['shuffled_idx_img']))
    pm = MerlotModel(cfg, True, False, dev(b['image']), dev(b['input_ids']), mask_input=True, shuffled_idx_img=sidx, params=st,
                     noise={k: torch.from_numpy(v) for k, v in b['noise'].items()})
    l1, _ = pm.mask_loss()
    l2, _ = pm.contrastive_loss()
    l3, _ = pm.temporal_loss(sidx, dev(torch.from_numpy(b['video_src_ids'])))
    st.zero_grad()
    (l1 + l2 + l3).backward()
    if device != 'cpu':
        torch.cuda.synchronize()
    g = {k: v.detach().float().cpu() for k, v in st.export_tf_grads().items()}
    return g, float((l1 + l2 + l3).detach()), pm.lang_mask_info['masked_idx'].cpu().numpy()


def run_all(which, verbose=False):
    import emu_ops
    cfg, b = problem(which)
    w = mo.init_weights(cfg, 0)
    # oracle
    wo = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    m = mo.MerlotOracle(cfg, wo, b['image'], b['input_ids'], mask_input=True, shuffled_idx_img=b['shuffled_idx_img'], noise=b['noise'])
    lo, _ = m.total_loss(b['shuffled_idx_img'], b['video_src_ids'])
    lo.backward()
    go = {k: v.grad.detach().float() for k, v in wo.items() if v.grad is not None}
    if verbose:
        print('oracle done', flush=True)
    with pytest.MonkeyPatch.context() as mp:
        emu_ops.install(mp)
        ge, le, idx_e = _run_product(cfg, w, b, 'cpu')
    if verbose:
        print('emulation done', flush=True)
    gh, lh, idx_h = _run_product(cfg, w, b, 'cuda')
    assert np.array_equal(idx_h, m.lang_mask_info['masked_idx'].numpy()) and np.array_equal(idx_e, idx_h)
    return {'hip': gh, 'emu': ge, 'oracle': go, 'loss': (lh, le, float(lo))}
