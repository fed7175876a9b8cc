"""Round 5: the persistent, prefetching single-pass forward (attention_pp.inc) against the resident one-shot forward
(attention_res.inc) on the ViT pass.  Experiments build: MERLOT_ATTN_PP=0/1 picks the kernel, MERLOT_ATTN_DBG=32 = data movement only."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops
from exp_attn_time import timeit


def ref(qkv, B, S):
    q, k, v = [t.reshape(B, S, 12, 64).permute(0, 2, 1, 3).float() for t in qkv.float().split(768, dim=1)]
    s = q @ k.transpose(-1, -2) * 0.125
    lse = torch.logsumexp(s, -1)
    o = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * S, 768)
    return o, lse


SC = int(os.environ.get('SCALE', 4))
for B, S in ((3, 198), (5, 65), (4, 96), (2, 128), (3, 161), (2, 224), (1, 198), (43, 198), (300, 198)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    outs = {}
    for k in ('0', '1'):
        os.environ['MERLOT_ATTN_PP'] = k
        outs[k] = ops.attention_fwd(qkv, B, S, 12, None)
    torch.cuda.synchronize()
    o_ref, lse_ref = ref(qkv, B, S)
    e = [float((outs[k][0].float() - o_ref).abs().max()) for k in ('0', '1')]
    el = [float((outs[k][1] - lse_ref).abs().max()) for k in ('0', '1')]
    d = float((outs['0'][0].float() - outs['1'][0].float()).abs().max())
    print(f'B {B:4d} S {S:4d}: |o - ref| res {e[0]:.2e} pp {e[1]:.2e} | |lse - ref| res {el[0]:.2e} pp {el[1]:.2e} | res vs pp {d:.2e}', flush=True)
    assert e[1] < 2.5e-2 and el[1] < 2e-3, 'pp forward off'

for B, S in ((512 * SC, 198), (128 * SC, 198), (512 * SC, 129)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    row = []
    for k, dbg in (('0', '0'), ('1', '0'), ('0', '0'), ('1', '0'), ('1', '32')):
        os.environ['MERLOT_ATTN_PP'] = k
        os.environ['MERLOT_ATTN_DBG'] = dbg
        t = timeit(lambda: ops.attention_fwd(qkv, B, S, 12, None))
        row.append(f'{ {"0": "res", "1": "pp"}[k] }{ {"0": "", "32": "(data only)", "64": "(nt)", "96": "(nt, data only)"}[dbg] } {t:7.1f} us')
    os.environ['MERLOT_ATTN_DBG'] = '0'
    gb = B * S * 12 * 64 * 2 * 4 / 1e9
    print(f'fwd B {B:5d} S {S:4d} ({gb:.2f} GB): ' + ' | '.join(row), flush=True)

# ---- backward: persistent (attention_pp.inc) against the one-launch fused kernel (attention_fb.inc) and fp32 autograd
def ref_bwd(qkv, do, B, S):
    x = qkv.float().requires_grad_(True)
    q, k, v = [t.reshape(B, S, 12, 64).permute(0, 2, 1, 3) for t in x.split(768, dim=1)]
    s = q @ k.transpose(-1, -2) * 0.125
    o = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * S, 768)
    o.backward(do.float())
    return x.grad


for B, S in ((3, 198), (5, 65), (4, 96), (2, 128), (3, 161), (2, 224), (1, 198), (43, 198), (300, 198)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    os.environ['MERLOT_ATTN_PP'] = '0'
    o, lse = ops.attention_fwd(qkv, B, S, 12, None)
    do = torch.randn_like(o)
    outs = {}
    for k in ('0', '1'):
        os.environ['MERLOT_ATTN_PP'] = k
        outs[k] = ops.attention_bwd(qkv, o, do, lse, B, S, 12, None)
    torch.cuda.synchronize()
    g = ref_bwd(qkv, do, B, S)
    rel = [float((outs[k].float() - g).norm() / g.norm()) for k in ('0', '1')]
    mx = [float((outs[k].float() - g).abs().max()) for k in ('0', '1')]
    d = float((outs['0'].float() - outs['1'].float()).abs().max())
    print(f'bwd B {B:4d} S {S:4d}: rel-L2 vs fp32 fused {rel[0]:.2e} pp {rel[1]:.2e} | max abs fused {mx[0]:.2e} pp {mx[1]:.2e} | fused vs pp {d:.2e}', flush=True)
    assert rel[1] < 1e-2 and torch.isfinite(outs['1'].float()).all(), 'pp backward off'

for B, S in ((512 * SC, 198), (128 * SC, 198), (512 * SC, 129)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    os.environ['MERLOT_ATTN_PP'] = '0'
    o, lse = ops.attention_fwd(qkv, B, S, 12, None)
    do = torch.randn_like(o)
    row = []
    PROBES = {1: 'p2 without dK / dV MFMAs + transposing reads', 2: 'p2 without transposing reads', 3: 'p2 without exp', 4: 'p2 without vector arithmetic',
              5: 'p2 without row-vector reads', 6: 'p2 without S / dP MFMAs + fragment reads'}
    for k, dbg in (('0', '0'), ('1', '0'), ('0', '0'), ('1', '0'), ('0', '1'), ('1', '32'), ('1', '64'), ('1', '128')) + \
            (tuple(('1', str(q << 16)) for q in PROBES) + (('1', '0'),) if S == 198 else ()):
        os.environ['MERLOT_ATTN_PP'] = k
        os.environ['MERLOT_ATTN_DBG'] = dbg
        t = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, S, 12, None))
        tag = {"0": "", "1": "(data only)", "32": "(data only)", "64": "(no pass 2)", "128": "(no pass 1)"}.get(dbg) if int(dbg) < 65536 else f'({PROBES[int(dbg) >> 16]})'
        row.append(f'{ {"0": "fused", "1": "pp"}[k] }{tag} {t:7.1f} us')
    os.environ['MERLOT_ATTN_DBG'] = '0'
    print(f'bwd B {B:5d} S {S:4d}: ' + ' | '.join(row), flush=True)

# ---- the masked forward of 257 .. 352 tokens (joint encoder, training step): persistent two-half kernel against the tiled kernel
from emu_ops import attention_fwd as emu_fwd  # noqa: E402  (fp32 restatement with the reference's masking semantics)
for B, S in ((3, 328), (2, 257), (2, 300), (5, 352), (30, 328), (64, 289)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    valid = (torch.rand(B, S, device='cuda') > 0.2).to(torch.uint8)
    valid[:, 0] = 1
    valid[0, S // 2:] = 0                                 # a long padded tail
    outs = {}
    for k in ('0', '1'):
        os.environ['MERLOT_ATTN_PP'] = k
        outs[k] = ops.attention_fwd(qkv, B, S, 12, valid)
    torch.cuda.synchronize()
    o_ref, lse_ref = emu_fwd(qkv.cpu(), B, S, 12, valid.cpu())
    e = [float((outs[k][0].float().cpu() - o_ref.float()).abs().max()) for k in ('0', '1')]
    el = [float((outs[k][1].cpu() - lse_ref).abs().max()) for k in ('0', '1')]
    d = float((outs['0'][0].float() - outs['1'][0].float()).abs().max())
    print(f'masked fwd B {B:4d} S {S:4d}: |o - ref| tiled {e[0]:.2e} pp {e[1]:.2e} | |lse - ref| tiled {el[0]:.2e} pp {el[1]:.2e} | tiled vs pp {d:.2e}', flush=True)
    assert e[1] < 2.5e-2 and el[1] < 2e-2, 'masked pp forward off'
for B, S in ((128 * SC, 328), (128 * SC, 289)):
    qkv = (torch.randn(B * S, 2304, device='cuda') * 0.7).bfloat16()
    valid = (torch.rand(B, S, device='cuda') > 0.1).to(torch.uint8)
    valid[:, 0] = 1
    row = []
    for k, dbg in (('0', '0'), ('1', '0'), ('0', '0'), ('1', '0'), ('1', '32')):
        os.environ['MERLOT_ATTN_PP'] = k
        os.environ['MERLOT_ATTN_DBG'] = dbg
        t = timeit(lambda: ops.attention_fwd(qkv, B, S, 12, valid))
        row.append(f'{ {"0": "tiled", "1": "pp"}[k] }{"(data only)" if dbg != "0" else ""} {t:7.1f} us')
    os.environ['MERLOT_ATTN_DBG'] = '0'
    print(f'masked fwd B {B:5d} S {S:4d}: ' + ' | '.join(row), flush=True)
