"""CPU: libmerlot_hip.so loads and exports every symbol include/merlot_hip.h declares (no compute calls)."""
import ctypes
import os

from merlot_amd import lib


def test_header_declares_the_survey_export_list():
    protos = lib.parse_header()
    need = ['merlot_patch_embed_fwd', 'merlot_patch_embed_wgrad', 'merlot_gemm_bf16_nt', 'merlot_gemm_bf16_tn',
            'merlot_ln_fwd', 'merlot_ln_bwd', 'merlot_attention_fwd', 'merlot_attention_bwd', 'merlot_attention_colsum', 'merlot_attention_workspace_bytes',
            'merlot_gather_add4', 'merlot_scatter_add_rows', 'merlot_softmax_ce', 'merlot_cls_avgpool_fwd',
            'merlot_cls_avgpool_bwd', 'merlot_adamw_step', 'merlot_mask_inputs', 'merlot_temporal_labels',
            'merlot_shuffled_idx', 'merlot_last_error', 'merlot_gemm_bf16_nt_plan']
    for n in need:
        assert n in protos, n


# SURVEY.md 8(b), "Minimum export list", name by name -> the entry points of include/merlot_hip.h that provide it.  The survey wrote
# its list before any kernel existed; where this ABI factors the work differently the row says how (INTEGRATION.md section B has the
# same table with the reference lines).
SURVEY_8B = {
    'merlot_patch_embed_fwd': ['merlot_patch_embed_fwd', 'merlot_im2col_patches'],
    'merlot_patch_embed_bwd': ['merlot_patch_embed_wgrad'],                 # the image is not differentiated: weight gradient only
    'merlot_gemm_bf16_nt': ['merlot_gemm_bf16_nt', 'merlot_gemm_nt_workspace_bytes', 'merlot_gemm_bf16_nt_plan'],
    'merlot_gemm_bf16_tn': ['merlot_gemm_bf16_tn', 'merlot_gemm_bf16_tn_cs', 'merlot_gemm_bf16_tn_workspace_bytes'],
    # dgrad dX = dY . W reads W through its transposed bf16 working copy (refreshed once per step with the cast): NT + the transposes
    'merlot_gemm_bf16_nn': ['merlot_gemm_bf16_nt', 'merlot_cast_transpose_f32_bf16', 'merlot_cast_transpose_batched'],
    # epilogue enum {none, bias, bias_gelu, bias_residual, bias_dropout_residual} = merlot_epilogue + bias / dropout_p arguments
    # round 6 (ABI v8): ONE entry -- the residual GEMM's launch also emits LayerNorm(h) (merlot_gemm_bf16_nt_ln); merlot_ln_fwd stays for the stand-alone sites
    'merlot_ln_residual_fwd': ['merlot_gemm_bf16_nt_ln', 'merlot_gemm_nt_ln_workspace_bytes', 'merlot_gemm_bf16_nt_ln_plan', 'merlot_ln_fwd'],
    'merlot_ln_residual_bwd': ['merlot_ln_bwd'],                            # dres / branch gradient / bias column sums fused
    'merlot_qkv_attention_fwd': ['merlot_attention_fwd', 'merlot_attention_workspace_bytes'],                   # colsum_out / blocksum_out = colsum_lo / colsum_hi
    'merlot_qkv_attention_bwd': ['merlot_attention_bwd'],
    'merlot_bias_gelu_fwd': ['merlot_gelu_fwd'],
    'merlot_bias_gelu_bwd': ['merlot_gelu_bwd'],
    'merlot_gather_rows_fwd': ['merlot_gather_add4'],
    'merlot_gather_rows_bwd': ['merlot_scatter_add_rows', 'merlot_scatter_add_sorted'],   # many rows onto few table rows: sorted + per-run reduction
    'merlot_vocab_ce_fwd': ['merlot_vocab_ce_fwd', 'merlot_vocab_ce_scratch_bytes'],
    'merlot_vocab_ce_bwd': ['merlot_gemm_bf16_nt', 'merlot_gemm_bf16_tn', 'merlot_colsum_bf16'],   # dlogits come out of the forward
    'merlot_contrastive_logits_ce_fwd': ['merlot_l2norm_fwd', 'merlot_gemm_bf16_nt', 'merlot_softmax_ce'],
    'merlot_contrastive_logits_ce_bwd': ['merlot_gemm_bf16_nt', 'merlot_l2norm_bwd'],
    'merlot_avgpool_posemb_ln_fwd': ['merlot_cls_avgpool_fwd', 'merlot_gather_add4', 'merlot_ln_fwd'],
    'merlot_avgpool_posemb_ln_bwd': ['merlot_ln_bwd', 'merlot_cls_avgpool_bwd'],
    'merlot_adamw_bf16state_step': ['merlot_adamw_step'],                   # state_bf16 = 1
}


def test_survey_8b_export_list_is_covered():
    """every name of SURVEY 8(b)'s own minimum export list is provided by exported entry points."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    survey = open(os.path.join(root, 'SURVEY.md')).read()
    line = [l for l in survey.splitlines() if l.startswith('`merlot_patch_embed_{fwd,bwd}`')][0]
    names = []
    for stem, alts in re.findall(r'`(merlot_[a-z0-9_]+?)(?:_\{([a-z,]+)\})?`', line):
        names += [f'{stem}_{a}' for a in alts.split(',')] if alts else [stem]
    assert len(names) >= 20
    assert sorted(set(names)) == sorted(SURVEY_8B), sorted(set(names) ^ set(SURVEY_8B))
    protos = lib.parse_header()
    dll = ctypes.CDLL(lib.LIB_PATH)
    for want, have in SURVEY_8B.items():
        for h in have:
            assert h in protos and hasattr(dll, h), (want, h)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(lib.LIB_PATH), "build the extension first: python -c 'import __graft_entry__ as g; g.build()'"
    dll = ctypes.CDLL(lib.LIB_PATH)
    for name in lib.parse_header():
        assert hasattr(dll, name), f"{name} declared in include/merlot_hip.h but not exported"
    d = lib.LIB.load()
    assert d.merlot_abi_version() == 11
    assert d.merlot_last_error() is not None


def test_product_library_carries_no_experiment_hooks():
    """the benchmarked library must not be steerable from the environment and must not export
    diagnostics.  The experiment switches live behind -DMERLOT_EXPERIMENTS (libmerlot_hip_exp.so, scripts/ only) and
    the probes in libmerlot_probe.so."""
    dll = ctypes.CDLL(lib.LIB_PATH)
    for name in ('merlot_probe_mfma32', 'merlot_probe_tr16', 'merlot_probe_tr8', 'merlot_probe_cvt8', 'merlot_probe_cu_hog', 'merlot_probe_mfma_rate',
                 'merlot_probe_persist_trace'):
        assert not hasattr(dll, name), f"{name} exported from the product library"
    blob = open(lib.LIB_PATH, 'rb').read()
    assert b'g_persist_ctr' not in blob and b'g_persist_seq' not in blob      # no library-owned device / host state
    for knob in (b'MERLOT_DBG', b'MERLOT_NT_CFG', b'MERLOT_NT_TILE_CG', b'MERLOT_NT_PERSIST_ID', b'MERLOT_TN_CFG',
                 b'MERLOT_TN_SPLITS'):
        assert knob not in blob, f"environment knob {knob.decode()} compiled into the product library"
    probe_path = os.path.join(os.path.dirname(lib.LIB_PATH), 'libmerlot_probe.so')
    assert os.path.exists(probe_path)
    pdll = ctypes.CDLL(probe_path)
    hdr = os.path.join(os.path.dirname(os.path.dirname(lib.LIB_PATH)), 'include', 'merlot_probe.h')
    for name in lib.parse_header(hdr):
        assert hasattr(pdll, name), f"{name} declared in include/merlot_probe.h but not exported"


def test_nt_plan_is_a_pure_function_of_the_shape():
    d = lib.LIB.load()
    assert d.merlot_gemm_bf16_nt_plan(101376, 2304, 768) == 22        # ViT QKV at the bench batch: persistent ping-pong
    assert d.merlot_gemm_bf16_nt_plan(101376, 768, 3072) == 22        # fc2
    assert d.merlot_gemm_bf16_nt_plan(101376, 768, 768) == 22         # out-projection: ping-pong persistent too (round 2)
    assert d.merlot_gemm_bf16_nt_plan(4000, 768, 768) == 11
    assert d.merlot_gemm_bf16_nt_plan(50176, 64, 576) == 14
    assert d.merlot_gemm_bf16_nt_plan(128, 128, 100) == -1            # K % 64 != 0 is rejected by the entry point


def test_argument_validation_happens_before_any_launch():
    """error convention: negative status + message, no exception from C, no GPU needed for the shape checks."""
    d = lib.LIB.load()
    rc = d.merlot_gemm_bf16_nt(None, 8, None, 8, None, 8, 4, 4, 64, 1.0, 0, 0, 0, None, None, 0, None, 0, 0.0, 0, None, None, 0, None)
    assert rc == -1 and b'null operand' in d.merlot_last_error()
    # the persistent kernels' tile-claim counters are the CALLER's: a shape that runs one is refused
    # without the workspace -- before any launch -- and the library exports no counter pool of its own
    fake = 4096
    assert d.merlot_gemm_nt_workspace_bytes() == 64
    rc = d.merlot_gemm_bf16_nt(fake, 768, fake, 768, fake, 768, 101376, 768, 768, 1.0, 0, 0, 0, None, None, 0, None, 0, 0.0, 0, None,
                               None, 0, None)
    assert rc == -1 and b'workspace' in d.merlot_last_error()
    rc = d.merlot_gemm_bf16_nt(fake, 768, fake, 768, fake, 768, 101376, 768, 768, 1.0, 0, 0, 0, None, None, 0, None, 0, 0.0, 0, None,
                               fake, 16, None)
    assert rc == -1 and b'need 64' in d.merlot_last_error()
    rc = d.merlot_ln_fwd(1, 0, 1, 1, 1, None, None, None, 4, 700, 1e-5, None)
    assert rc == -1 and b'H=700' in d.merlot_last_error()
    rc = d.merlot_attention_fwd(None, 2304, None, 768, None, None, None, 1, 4, 12, 0.125, None, None, 4, 0, 1.0, None, 0, None)
    assert rc == -1
    # ABI v7: the persistent attention kernels claim their items from CALLER-owned counters; a workspace of the wrong size is refused
    assert d.merlot_attention_workspace_bytes() == 64
    rc = d.merlot_attention_fwd(fake, 2304, fake, 768, fake, None, None, 1, 198, 12, 0.125, None, None, 198, 0, 1.0, fake, 16, None)
    assert rc == -1 and b'workspace' in d.merlot_last_error()
    # the frame-kernel job table is checked on its HOST copy before anything is launched
    import numpy as np
    from merlot_amd.input_pipeline import JOB_DTYPE
    jobs = np.zeros(2, JOB_DTYPE)
    jobs[0] = (0, 8, 8, 16, 16, 0, 0, 0, 0, (1, 1, 1), 0)
    jobs[1] = (192, 8, 8, 16, 16, 5, 0, 0, 0, (1, 1, 1), 0)            # resize method 5 does not exist
    fake = 4096                                                      # never dereferenced: validation fails first
    rc = d.merlot_image_frames(fake, 384, jobs.ctypes.data, fake, 2, fake, 16, 16, fake, 1 << 20, None)
    assert rc == -1 and b'frame 1: resize method 5' in d.merlot_last_error()
    jobs[1]['method'] = 0
    rc = d.merlot_image_frames(fake, 300, jobs.ctypes.data, fake, 2, fake, 16, 16, fake, 1 << 20, None)
    assert rc == -1 and b'frame 1 lies outside the source buffer' in d.merlot_last_error()
    rc = d.merlot_image_frames(fake, 384, jobs.ctypes.data, fake, 2, fake, 16, 16, fake, 16, None)
    assert rc == -1 and b'workspace too small' in d.merlot_last_error()
    rc = d.merlot_im2col_patches(fake, fake, 1, 60, 60, 12, -0.5, None)        # 3 * 12 * 12 = 432: not a multiple of 64
    assert rc == -1 and b'patch_size must be a multiple of 8' in d.merlot_last_error()


def test_product_has_no_cpu_fallback():
    import pytest
    import torch
    from merlot_amd import ops
    with pytest.raises(ValueError, match="no CPU fallback"):
        ops.gemm_nt(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(4, 64, dtype=torch.bfloat16))


def test_abi_v9_8bit_entries_validate_without_a_gpu():
    """round 6 (judge row g1): the 8-bit backward's entry points exist with the declared signatures and refuse bad arguments before any launch."""
    d = lib.LIB.load()
    protos = lib.parse_header()
    for name in ('merlot_quantize_f8', 'merlot_gemm_f8_tn', 'merlot_gemm_f8_tn_workspace_bytes', 'merlot_gemm_f8_nt', 'merlot_gemm_bf16_nt_q8',
                 'merlot_gemm_fp8_nt_q8', 'merlot_ln_fwd_q8t', 'merlot_ln_bwd_q8', 'merlot_f8_scale_rotate'):
        assert name in protos and hasattr(d, name), name
    # the split plan is a pure function of the shape: tiles x chunks fill one round of 256 workgroups, chunks of >= 8 K-tiles of 128 rows
    assert d.merlot_gemm_f8_tn_workspace_bytes(768, 3072, 443904) == 7 * 768 * 3072 * 4
    assert d.merlot_gemm_f8_tn_workspace_bytes(768, 768, 443904) == 28 * 768 * 768 * 4
    assert d.merlot_gemm_f8_tn_workspace_bytes(768, 3072, 1000) == 0            # refused shapes need nothing
    assert d.merlot_gemm_f8_tn(None, 768, 0, None, None, 768, 0, None, None, 768, 768, 768, 4096, 1.0, 0, None, 0, None) == -1
    assert b'null operand' in d.merlot_last_error()
    import ctypes as C
    buf = (C.c_char * 64)()
    p = C.cast(buf, C.c_void_p)
    assert d.merlot_gemm_f8_tn(p, 768, 2, p, p, 768, 0, p, p, 768, 768, 768, 4096, 1.0, 0, None, 0, None) == -1      # format 2 does not exist
    assert d.merlot_gemm_f8_tn(p, 768, 0, p, p, 768, 0, p, p, 768, 768, 768, 4000, 1.0, 0, None, 0, None) == -1      # R % 128
    assert b'R %% 128' in d.merlot_last_error() or b'128' in d.merlot_last_error()
    assert d.merlot_gemm_bf16_nt_q8(p, 768, p, 768, None, 0, 1024, 3000, 768, 1.0, 3, None, p, 3000, None, p, 3000, 1, p, p, 64, None) == -1   # N % 256
