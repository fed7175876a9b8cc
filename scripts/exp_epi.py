"""Cost of each NT epilogue relative to the bare main loop (MERLOT_DBG=1 skips the epilogue).
gpurun: python scripts/exp_epi.py"""
import _exp_lib  # noqa: F401  (experiments build of the library + probes)
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_amd import ops

dev = 'cuda'
torch.manual_seed(0)
T = 101376


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    for N, K, name in [(768, 768, 'proj'), (768, 3072, 'fc2'), (3072, 768, 'fc1/dfc2'), (2304, 768, 'qkv')]:
        a = torch.randn(T, K, device=dev).bfloat16()
        b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        bias = torch.zeros(N, device=dev)
        res = torch.randn(T, N, device=dev).bfloat16()
        aux = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
        flops = 2.0 * T * N * K
        cases = {
            'none': lambda: ops.gemm_nt(a, b, bias=bias),
            'gelu(+preact out)': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_GELU, aux_out=aux),
            'residual p=0': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res),
            'residual p=0.1': lambda: ops.gemm_nt(a, b, bias=bias, epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1,
                                                  dropout_seed=123),
            'dgelu': lambda: ops.gemm_nt(a, b, epilogue=ops.EPI_DGELU, aux_in=res),
        }
        os.environ['MERLOT_DBG'] = '1'
        t_loop = bench(cases['none'])
        os.environ['MERLOT_DBG'] = '0'
        print(f'{name:9s} N={N} K={K}: main loop only {t_loop:7.1f} us {flops / t_loop / 1e6:6.0f} TF', flush=True)
        for k, fn in cases.items():
            t = bench(fn)
            os.environ['MERLOT_DBG'] = '8'
            t8 = bench(fn)
            os.environ['MERLOT_DBG'] = '16'
            t16 = bench(fn)
            os.environ['MERLOT_DBG'] = '0'
            print(f'    {k:20s} {t:7.1f} us {flops / t / 1e6:6.0f} TF   epilogue +{t - t_loop:6.1f} us (+{100 * (t / t_loop - 1):4.1f} %)'
                  f' | no-mem epilogue {t8:7.1f} us | relaxed post-epilogue waits {t16:7.1f} us', flush=True)


if __name__ == '__main__':
    main()
