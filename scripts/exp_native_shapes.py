"""Round 5: what the HIP model does on the as-shipped geometry (tests/golden/ref_shim_native.npz) -- every number the bounds of
tests/test_native_yaml_gpu.py were set from."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from common import rel_l2, head  # noqa: E402
from grad_parity import tensor_class  # noqa: E402
import native_shapes as ns  # noqa: E402

for name in ns.PROBLEMS:
    cfg, batch, w, noise, fx = ns.load(name)
    pm, losses, g = ns.run_hip(cfg, batch, w, noise)
    print(f'=== {name}: image {cfg["image_size"]}, P {pm.P}, L {pm.L}, resnet {cfg.get("resnet_layers")}')
    print('masked_idx equal', np.array_equal(pm.lang_mask_info['masked_idx'].cpu().numpy(), fx['masked_idx']),
          'masked_ids equal', np.array_equal(pm.lang_mask_info['masked_ids'].cpu().numpy(), fx['masked_ids']))
    hs = pm.vision_transformer_info['hidden_state'].float().cpu()
    print('vs the reference run: attention_summs %.2e | ViT hidden rows %.2e | img_trg_h %.2e lang_trg_h %.2e | encoder viz %.2e lang %.2e' % (
        rel_l2(pm.lang_transformer_info['attention_summs'].reshape(pm.B, pm.L), torch.from_numpy(fx['attention_summs'])),
        rel_l2(hs[:, torch.from_numpy(fx['vit_rows']).long(), :], torch.from_numpy(fx['vit_hidden_rows'])),
        rel_l2(pm.img_trg_h, torch.from_numpy(fx['img_trg_h'])), rel_l2(pm.lang_trg_h, torch.from_numpy(fx['lang_trg_h'])),
        rel_l2(pm.encoder_hidden_states['viz'], torch.from_numpy(fx['encoder_viz'])),
        rel_l2(pm.encoder_hidden_states['lang'], torch.from_numpy(fx['encoder_lang']))))
    print('losses hip', losses, 'reference', list(fx['losses']))
    norms = dict(zip([str(n) for n in fx['grad_names']], fx['grad_norms']))
    for grp, sel in (('outside the stem', lambda n: not ns.is_stem(n)), ('stem', ns.is_stem)):
        r = [abs(float(g[n].double().norm()) - v) / v for n, v in norms.items() if sel(n) and not n.endswith('key_layer/bias')]
        if r:
            print(f'gradient norms vs the reference run, {grp}: {len(r)} tensors, median {np.median(r):.2e} max {max(r):.2e}')
    for k in sorted(fx):
        if k.startswith('grad/'):
            n = k[5:]
            print('   sample %-90s %s rel-L2 %.2e' % (n, tensor_class(n), rel_l2(torch.from_numpy(head(g[n].numpy())), torch.from_numpy(fx[k]))))
    for bf16 in ((False, True) if cfg.get('resnet_layers') else (False,)):
        m, lo, go = ns.run_oracle(cfg, batch, w, noise, bf16)
        print(f'--- vs the oracle ({"bf16-policy stem" if bf16 else "fp32"}): loss {lo:.5f} vs hip {sum(losses):.5f}; ViT hidden %.2e encoder viz %.2e lang %.2e' % (
            rel_l2(hs, m.vision_transformer_info['hidden_state']), rel_l2(pm.encoder_hidden_states['viz'], m.encoder_hidden_states['viz']),
            rel_l2(pm.encoder_hidden_states['lang'], m.encoder_hidden_states['lang'])))
        by = {}
        for n, gr in go.items():
            if n.endswith('key_layer/bias') or float(gr.norm()) == 0:
                continue
            c = ('stem-' if ns.is_stem(n) else '') + tensor_class(n)
            rel = float((g[n].double() - gr.double()).norm() / gr.double().norm())
            ratio = float(g[n].double().norm() / gr.double().norm()) - 1
            cos = float(torch.dot(g[n].flatten().double(), gr.flatten().double()) / (g[n].double().norm() * gr.double().norm()))
            by.setdefault(c, []).append((rel, abs(ratio), cos, n))
        for c, v in sorted(by.items()):
            rels = [t[0] for t in v]
            print(f'   {c:12s} {len(v):3d} tensors: rel-L2 median {np.median(rels):.2e} max {max(rels):.2e} ({max(v)[3]}) | norm ratio max {max(t[1] for t in v):.2e} | cosine min {min(t[2] for t in v):.4f}')
