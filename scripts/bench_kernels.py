"""Per-kernel microbenchmarks on the shapes of BASELINE config #2 (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlot_amd import ops

BF16, F32 = torch.bfloat16, torch.float32


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    seg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    T = seg * 198
    g = torch.Generator(device='cuda').manual_seed(0)
    rnd = lambda *s: (torch.randn(s, device='cuda', generator=g) * 0.5).to(BF16)
    print(f"T={T} tokens (ViT pass of {seg} segments)")
    for (N, K, name) in [(2304, 768, 'qkv'), (768, 768, 'proj'), (3072, 768, 'fc1'), (768, 3072, 'fc2')]:
        a, bt = rnd(T, K), rnd(N, K)
        bias = torch.zeros(N, device='cuda')
        t = timeit(lambda: ops.gemm_nt(a, bt, bias=bias))
        print(f"gemm_nt {name:5s} M={T} N={N} K={K}: {t*1e6:8.1f} us  {2*T*N*K/t/1e12:7.1f} TF")
    res = rnd(T, 768)
    a, bt = rnd(T, 3072), rnd(768, 3072)
    t = timeit(lambda: ops.gemm_nt(a, bt, bias=torch.zeros(768, device='cuda'), epilogue=ops.EPI_RESIDUAL, aux_in=res, dropout_p=0.1, dropout_seed=1))
    print(f"gemm_nt fc2+res+dropout: {t*1e6:8.1f} us  {2*T*768*3072/t/1e12:7.1f} TF")
    for (M, N, name) in [(2304, 768, 'dWqkv'), (768, 768, 'dWproj'), (3072, 768, 'dW1'), (768, 3072, 'dW2')]:
        a, b = rnd(T, M), rnd(T, N)
        out = torch.zeros((M, N), device='cuda')
        t = timeit(lambda: ops.gemm_tn(a, b, out, accumulate=True))
        print(f"gemm_tn {name:6s} M={M} N={N} R={T}: {t*1e6:8.1f} us  {2*T*N*M/t/1e12:7.1f} TF")
    # attention
    for (B, S, nm) in [(seg, 198, 'vit'), (seg // 4, 328, 'joint'), (seg // 16, 512, 'text')]:
        qkv = rnd(B * S, 2304)
        o, lse = ops.attention_fwd(qkv, B, S, 12)
        do = rnd(B * S, 768)
        fl = 4.0 * B * 12 * S * S * 64
        t = timeit(lambda: ops.attention_fwd(qkv, B, S, 12))
        t2 = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, B, S, 12))
        print(f"attention {nm:5s} B={B} S={S}: fwd {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF | bwd {t2*1e6:8.1f} us {2.5*fl/t2/1e12:6.1f} TF(5 matmuls)")
    # layernorm / colsum / cast
    x = rnd(T, 768)
    gam, bet = torch.ones(768, device='cuda'), torch.zeros(768, device='cuda')
    t = timeit(lambda: ops.ln_fwd(x, gam, bet))
    print(f"ln_fwd  rows={T}: {t*1e6:8.1f} us  {2*T*768*2/t/1e9:7.1f} GB/s")
    _, _, mean, rstd = ops.ln_fwd(x, gam, bet)
    dg, db = torch.zeros(768, device='cuda'), torch.zeros(768, device='cuda')
    t = timeit(lambda: ops.ln_bwd(x, x, mean, rstd, gam, dg, db, dres=x))
    print(f"ln_bwd  rows={T}: {t*1e6:8.1f} us  {4*T*768*2/t/1e9:7.1f} GB/s")
    y = rnd(T, 3072)
    ob = torch.zeros(3072, device='cuda')
    t = timeit(lambda: ops.colsum_bf16(y, ob))
    print(f"colsum  [{T},3072]: {t*1e6:8.1f} us  {T*3072*2/t/1e9:7.1f} GB/s")
    t = timeit(lambda: ops.dropout_apply(x, 0.1, 5))
    print(f"dropout_apply [{T},768]: {t*1e6:8.1f} us  {2*T*768*2/t/1e9:7.1f} GB/s")
    m = torch.randn(223_000_000, device='cuda')
    mb = torch.empty(223_000_000, device='cuda', dtype=BF16)
    t = timeit(lambda: ops.cast_bf16(m, mb), iters=5)
    print(f"cast 223M: {t*1e6:8.1f} us  {223e6*6/t/1e9:7.1f} GB/s")
    gr = torch.randn(223_000_000, device='cuda') * 0.01
    mm, vv = torch.zeros(223_000_000, device='cuda', dtype=BF16), torch.zeros(223_000_000, device='cuda', dtype=BF16)
    t = timeit(lambda: ops.adamw_step(m, gr, mm, vv, 1e-4, 0.9, 0.98, 1e-6, 0.1), iters=5)
    print(f"adamw 223M: {t*1e6:8.1f} us  {223e6*20/t/1e9:7.1f} GB/s")


if __name__ == '__main__':
    main()
