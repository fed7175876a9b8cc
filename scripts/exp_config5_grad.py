"""Round 6: the numbers behind the gradient bounds of tests/test_edge_cases_gpu.py::test_config5_geometry_gradients_against_the_oracle (384^2, 16-segment groups,
2 + 2 layers): per tensor class, rel-L2 and norm ratio of the HIP gradients against the fp32 oracle."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from common import tiny_config, synth_batch, rel_l2  # noqa: E402
from grad_parity import tensor_class  # noqa: E402
import test_edge_cases_gpu as T  # noqa: E402

cfg = tiny_config(image_size=[384, 384], num_chunks_in_group=16, max_position_embeddings=1024)
b = synth_batch(cfg, E=1, num_chunks=16, seed=5)
w, m, st, pm = T._both(cfg, b, grads=True)
loss, info = m.total_loss(b['shuffled_idx_img'], b['video_src_ids'])
loss.backward()
st.zero_grad()
l = pm.mask_loss()[0] + pm.contrastive_loss()[0] + pm.temporal_loss(torch.from_numpy(b['shuffled_idx_img']).cuda(), torch.from_numpy(b['video_src_ids']).cuda())[0]
l.backward()
torch.cuda.synchronize()
gt = st.export_tf_grads()
rows = {}
for k, v in w.items():
    if v.grad is None or k.endswith('key_layer/bias'):
        continue
    c = 'contrastive' if k.startswith('contrastive/') else tensor_class(k)
    rows.setdefault(c, []).append((rel_l2(gt[k], v.grad), abs(float(gt[k].float().norm().cpu() / v.grad.norm()) - 1.0), k))
for c, r in sorted(rows.items()):
    worst = max(r)
    print(f'class {c:12s} n={len(r):3d}  rel-L2 max {worst[0]:.2e} ({worst[2][-50:]}) median {np.median([x[0] for x in r]):.2e}   |norm - 1| max {max(x[1] for x in r):.1e}')
print('loss hip / oracle', float(l), float(loss))
