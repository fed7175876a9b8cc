"""CPU: libmerlot_hip.so loads and exports every symbol include/merlot_hip.h declares (no compute calls)."""
import ctypes
import os

from merlot_amd import lib


def test_header_declares_the_survey_export_list():
    protos = lib.parse_header()
    need = ['merlot_patch_embed_fwd', 'merlot_patch_embed_wgrad', 'merlot_gemm_bf16_nt', 'merlot_gemm_bf16_tn',
            'merlot_ln_fwd', 'merlot_ln_bwd', 'merlot_attention_fwd', 'merlot_attention_bwd', 'merlot_attention_colsum',
            'merlot_gather_add4', 'merlot_scatter_add_rows', 'merlot_softmax_ce', 'merlot_cls_avgpool_fwd',
            'merlot_cls_avgpool_bwd', 'merlot_adamw_step', 'merlot_mask_inputs', 'merlot_temporal_labels',
            'merlot_shuffled_idx', 'merlot_last_error']
    for n in need:
        assert n in protos, n


def test_library_exports_every_declared_symbol():
    assert os.path.exists(lib.LIB_PATH), "build the extension first: python -c 'import __graft_entry__ as g; g.build()'"
    dll = ctypes.CDLL(lib.LIB_PATH)
    for name in lib.parse_header():
        assert hasattr(dll, name), f"{name} declared in include/merlot_hip.h but not exported"
    d = lib.LIB.load()
    assert d.merlot_abi_version() == 1
    assert d.merlot_last_error() is not None


def test_argument_validation_happens_before_any_launch():
    """error convention: negative status + message, no exception from C, no GPU needed for the shape checks."""
    d = lib.LIB.load()
    rc = d.merlot_gemm_bf16_nt(None, 8, None, 8, None, 8, 4, 4, 64, 1.0, 0, 0, 0, None, None, 0, None, 0, 0.0, 0, None)
    assert rc == -1 and b'null operand' in d.merlot_last_error()
    rc = d.merlot_ln_fwd(1, 0, 1, 1, 1, None, None, None, 4, 700, 1e-5, None)
    assert rc == -1 and b'H=700' in d.merlot_last_error()
    rc = d.merlot_attention_fwd(None, 2304, None, 768, None, None, None, 1, 4, 12, 0.125, None)
    assert rc == -1
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
    rc = d.merlot_im2col_patches(fake, fake, 1, 64, 64, 8, -0.5, None)
    assert rc == -1 and b'patch_size 16' in d.merlot_last_error()


def test_product_has_no_cpu_fallback():
    import pytest
    import torch
    from merlot_amd import ops
    with pytest.raises(ValueError, match="no CPU fallback"):
        ops.gemm_nt(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(4, 64, dtype=torch.bfloat16))
