"""Shared builders for the parity tests: configs, seeded synthetic inputs, explicit noise."""
import numpy as np
import torch

from oracle import index_oracle as ix


def tiny_config(**over):
    """BASELINE config #1: merlot.yaml 4-segment, 2-layer ViT + 2-layer joint (+2-layer text-only), 64x64 frames."""
    cfg = dict(num_chunks_in_group=4, masking_use_attn=True, masking_rate=0.2, masking_do_spanbert=True,
               masking_choose_topk_prob=0.5, image_shuffle_prob=0.4, masking_spanbert_len_probs=[0.625, 0.25, 0.125],
               do_projection=True, do_bias=True, image_size=[64, 64], patch_size=16, spatial_pool_size=2,
               use_bfloat16=True, vocab_size=50370, hidden_size=768, contrastive_size=768, contrast_coef=0.25,
               contrast_temp=0.05, attention_probs_dropout_prob=0.0, hidden_dropout_prob=0.0, initializer_range=0.02,
               intermediate_size=3072, max_position_embeddings=1024, num_attention_heads=12, num_hidden_layers=2,
               num_vision_transformer_hidden_layers=2, num_lang_transformer_hidden_layers=2, share_params=True)
    cfg.update(over)
    return cfg


def synth_batch(cfg, E=2, num_chunks=4, Lc=32, seed=1, two_videos=True):
    """Synthetic inputs of SURVEY.md 8(d): images U[0,1) (bf16-representable), START + 8..31 tokens + padding."""
    g = torch.Generator().manual_seed(seed)
    H, W = cfg['image_size']
    img = torch.rand(E * num_chunks, H, W, 3, generator=g).to(torch.bfloat16).float()
    ids = torch.zeros(E, num_chunks, Lc, dtype=torch.long)
    for e in range(E):
        for c in range(num_chunks):
            ln = int(torch.randint(8, Lc, (1,), generator=g))
            ids[e, c, 0] = 2
            ids[e, c, 1:ln] = torch.randint(100, 50354, (ln - 1,), generator=g)
    n = cfg['num_chunks_in_group']
    B = E * num_chunks // n
    L = Lc * n
    rng = np.random.RandomState(seed)
    nm = int(L * cfg.get('masking_rate', 0.2))
    noise = dict(gumbel=(-np.log(-np.log(rng.uniform(size=(B, L))))).astype(np.float32),
                 span_lower=rng.choice(3, size=(B, nm), p=[0.625, 0.25, 0.125]).astype(np.int32),
                 span_upper=rng.choice(3, size=(B, nm), p=[0.625, 0.25, 0.125]).astype(np.int32),
                 random_ids=rng.randint(100, cfg['vocab_size'], size=B * L).astype(np.int32),
                 option=rng.choice(3, size=B * L, p=[0.1, 0.8, 0.1]).astype(np.int32))
    pr = np.array(ix.num_shuffle_probs(n, cfg.get('image_shuffle_prob', 0.4)))
    pr = pr / pr.sum()
    num_shuffle = rng.choice(len(pr), size=B, p=pr).astype(np.int32)
    num_shuffle[0] = n - 1                                   # make sure the shuffled branch is exercised
    u_sel, u_perm = rng.uniform(size=(B, n)).astype(np.float32), rng.uniform(size=(B, n)).astype(np.float32)
    sidx = ix.shuffled_idx_img(B, n, cfg.get('image_shuffle_prob', 0.4), num_shuffle, u_sel, u_perm)
    vsrc = np.zeros((E, num_chunks), np.int32)
    if two_videos:
        vsrc[0, num_chunks // 2:] = 1
    return dict(image=img, input_ids=ids, noise=noise, shuffled_idx_img=sidx, video_src_ids=vsrc,
                num_shuffle=num_shuffle, u_select=u_sel, u_perm=u_perm, B=B, L=L)


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def head(a, rows=16):
    """how the shim fixtures store big tensors: [.., last] flattened to 2-D, first `rows` rows (small ones whole)."""
    a = np.asarray(a)
    return a.reshape(-1, a.shape[-1])[:rows] if a.size > 65536 else a
