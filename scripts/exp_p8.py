"""Round 2: the ping-pong persistent NT kernel (id 22, csrc/gemm_p8.inc) against the round-1 persistent kernel (id 21):
correctness on ragged / small / large shapes for every epilogue (bit-identical to id 21: same MFMA order per accumulator;
and against the torch fp32 GEMM), then timing is scripts/exp_skew.py with CFGS=21,22,22,21."""
import _exp_lib  # noqa: F401
import os
import torch
from merlot_amd import ops

dev = 'cuda'
torch.manual_seed(0)


def run(cfg, fn):
    os.environ['MERLOT_NT_CFG_DYN'] = str(cfg)
    return fn()


bad = 0
for (M, N, K) in [(256, 256, 128), (512, 768, 768), (4000, 768, 768), (300, 2304, 768), (16384, 3072, 768), (10100, 768, 3072),
                  (41984, 2304, 768), (777, 1000, 128), (70000, 3072, 768), (101376, 768, 3072)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev) * 0.1
    res = torch.randn(M, N, device=dev).bfloat16()
    cases = {
        'bias': lambda: (ops.gemm_nt(a, b, bias=bias),),
        'f32': lambda: (ops.gemm_nt(a, b, bias=bias, out_dtype=torch.float32, alpha=0.5),),
        'gelu': lambda: (lambda u: (ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_GELU, aux_out=u), u))(torch.zeros(M, N, device=dev, dtype=torch.bfloat16)),
        'res': lambda: (ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res),),
        'resdrop': lambda: (ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1, dropout_seed=5),),
        'dgelu': lambda: (ops.gemm_nt(a, b, epilogue=ops.EPI_DGELU, aux_in=res),),
    }
    ref32 = torch.addmm(bias, a.float(), b.float().t())
    for name, fn in cases.items():
        if N % 2 and name == 'resdrop':
            continue
        o21 = run(11 if M * N < 256 * 256 * 8 else 21, fn)
        for rep in range(3):
            o22 = run(22, fn)
            same = all(torch.equal(x, y) for x, y in zip(o21, o22))
            if not same:
                d = max(float((x.float() - y.float()).abs().max()) for x, y in zip(o21, o22))
                nbad = sum(int((x != y).sum()) for x, y in zip(o21, o22))
                print(f'MISMATCH {M}x{N}x{K} {name} rep {rep}: max abs diff {d:.4g}, {nbad} elements', flush=True)
                bad += 1
    got = run(22, cases['bias'])[0].float()
    rel = float((got - ref32).norm() / ref32.norm())
    print(f'{M}x{N}x{K}: id 22 vs torch fp32 rel-L2 {rel:.2e}', flush=True)
    assert rel < 6e-3
print('p8 correctness:', 'OK' if bad == 0 else f'{bad} MISMATCHES')
