"""The hot loops of the library, read off the compiled gfx950 code (CPU only: hipcc cross-compiles).  Round 5 found the same defect in four kernels: LDS
addresses that the compiler rebuilt with a v_add_u32 per read (operands beyond the 64 KiB of a ds_read immediate; the LDS symbol's own address added per
access), and, in the attention backward's dK / dV pass, row vectors held in 32 - 48 registers only to be subtracted.  The fixes are invisible in the
results -- so this test pins what the compiled loops look like: instruction counts of the loop bodies and the register counts that go with them."""
import os
import re
import shutil
import subprocess
import tempfile
from collections import Counter

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'merlot_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# The counts below are properties of ONE compiler's output: they are asserted for the toolchain they were read from and reported,
# not enforced, for any other (a newer LLVM may schedule the same source differently and still be right).
PINNED_HIPCC = '7.2.26015'


def _hipcc_version():
    try:
        out = subprocess.check_output([HIPCC, '--version'], text=True)
    except Exception:
        return ''
    m = re.search(r'HIP version:\s*(\S+)', out)
    return m.group(1) if m else ''


pytestmark = [pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not available'),
              pytest.mark.skipif(os.path.exists(HIPCC) and not _hipcc_version().startswith(PINNED_HIPCC),
                                 reason=f'instruction / register counts are pinned for hipcc {PINNED_HIPCC} only')]


def _compile(src):
    tmp = tempfile.mkdtemp(prefix='loop_isa_')
    out = os.path.join(tmp, 'k.s')
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-w', '-S', '--cuda-device-only', os.path.join(CSRC, src), '-o', out],
                          cwd=CSRC)
    text = open(out).read().split('\n')
    shutil.rmtree(tmp, ignore_errors=True)
    return text


def _kernel(lines, pattern):
    """-> (body lines, vgpr count) of the ONE kernel whose mangled name matches."""
    starts = [i for i, l in enumerate(lines) if re.match(r'^_ZN\S*' + pattern + r'\S*:', l)]
    assert len(starts) == 1, (pattern, len(starts))
    body = []
    for l in lines[starts[0] + 1:]:
        body.append(l)
        if l.startswith('.Lfunc_end'):
            break
    name = lines[starts[0]].split(':')[0]
    vg = None
    for i, l in enumerate(lines):
        if l.strip().startswith('.amdhsa_kernel ' + name):
            for m in lines[i:i + 80]:
                mm = re.match(r'\s*\.amdhsa_next_free_vgpr (\d+)', m)
                if mm:
                    vg = int(mm.group(1))
                    break
    assert vg is not None, name
    return body, vg


def _loops(body, n_mfma, inner_labels=False, max_len=400):
    """loops (label ... backward branch to the same label; inner_labels: other labels -- skipped-over side blocks -- may lie inside) that hold exactly
    n_mfma MFMAs in at most max_len instructions -> [Counter of mnemonics]"""
    out = []
    labels = [(k, b.split(':')[0]) for k, b in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', b)]
    for k, lab in labels:
        code, closed = [], False
        for b in body[k + 1:]:
            c = b.split(';')[0].strip()
            if re.match(r'^\.LBB', c):
                if inner_labels:
                    continue
                break
            if not c or c.startswith('.'):
                continue
            code.append(c)
            if (c.startswith('s_cbranch') or c.startswith('s_branch')) and c.endswith(lab):
                closed = True
                break
            if len(code) > max_len:
                break
        if closed and sum('v_mfma' in c for c in code) == n_mfma:
            out.append(Counter(c.split()[0] for c in code))
    return out


def _valu(c):
    return sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)


def test_attention_backward_loops():
    lines = _compile('attention.hip')
    # the persistent ViT backward: pass 2 (16 MFMAs per chunk) and pass 1 (12)
    body, vg = _kernel(lines, r'attn_bwd_pp_kernelILi7ELi0E')
    assert vg <= 224, vg                                         # 248 before the row vectors became the accumulators' initial value
    p2 = _loops(body, 16)
    assert len(p2) == 1 and _valu(p2[0]) <= 76 and p2[0]['v_add_u32_e32'] <= 10, p2       # 115 vector-ALU / 38 address additions before
    assert p2[0]['ds_read_b128'] == 16 and p2[0]['ds_read_b64_tr_b16'] == 16
    # the fused backward of the joint encoder (masked + attention log): two copies of the dK / dV chunk loop (the wave's two key blocks)
    # (both forms of the image arrival are in the product since the per-token-count rule of fb_bwd: <.., BPW = 2, CH = false | true>, the same chunk loops)
    for ch in (0, 1):
        body, vg = _kernel(lines, r'attn_bwd_fused_kernelILi512ELb1ELb1ELi2ELb%dE' % ch)
        assert vg <= 244, (ch, vg)
        p2 = _loops(body, 16)
        # (the chunked form's first copy carries the counted waits for the arriving row groups between its MFMAs: only its second copy is a plain 16-MFMA loop)
        assert len(p2) == 2 - ch and all(_valu(c) <= 145 and c['v_add_u32_e32'] <= 10 for c in p2), (ch, p2)      # 164 / 187 and 34 address additions before
        body, vg = _kernel(lines, r'attn_bwd_fused_kernelILi512ELb1ELb0ELi2ELb%dE' % ch)
        p2 = _loops(body, 16)
        assert len(p2) == 2 - ch and all(_valu(c) <= 108 and c['v_add_u32_e32'] <= 10 for c in p2), (ch, p2)      # 131 / 156 before

def test_gemm_main_loops():
    lines = _compile('gemm.hip')
    # ping-pong NT kernel, two phases per K-tile: 32 MFMAs per wave and K-tile, 24 fragment reads; 56 vector-ALU instructions before the bases were pinned
    for epi in range(4):
        body, vg = _kernel(lines, r'gemm_nt_p8_kernelILi%dELb0ELb0ELb1ELb0ELi0ELi0E' % epi)   # <EPI, bf16 out, bf16 operands, two phases, no LayerNorm fold, no 8-bit copy, A format 0>
        steady = _loops(body, 32, inner_labels=True)
        assert len(steady) == 1 and _valu(steady[0]) <= 50 and steady[0]['ds_read_b128'] == 24, (epi, steady)
        assert 'scratch_load_dwordx4' not in steady[0] and 'scratch_store_dwordx4' not in steady[0]
    # the RESIDUAL kernel that also emits LayerNorm(C) (round 6): the same main loop -- no scratch, 24 fragment reads; its extra epilogue state costs address
    # arithmetic in the loop (66 vector-ALU instructions when first built)
    body, vg = _kernel(lines, r'gemm_nt_p8_kernelILi2ELb0ELb0ELb1ELb1ELi0ELi0E')
    steady = _loops(body, 32, inner_labels=True)
    assert len(steady) == 1 and _valu(steady[0]) <= 70 and steady[0]['ds_read_b128'] == 24, steady
    assert 'scratch_load_dwordx4' not in steady[0] and 'scratch_store_dwordx4' not in steady[0] and 'scratch_load_dword' not in steady[0]
    # round 6, the 8-bit kernels.  The epilogues that also write an 8-bit copy (Q8 = 1 / 2 / 5 / 6) must not spill: their first version carried four more kernel
    # arguments and a 64-bit lane pointer through an epilogue already at 250 registers -- 384 .. 448 bytes of scratch per lane, launches 39 - 60 % slower
    # (profiles/r06_n_f8_producers.txt); with the epilogue operands re-read from the kernel-argument segment and a 32-bit lane offset they have the base kernels' 48 bytes
    for pat in (r'ILi1ELb0ELb1ELb1ELb0ELi1ELi0E', r'ILi1ELb0ELb1ELb1ELb0ELi5ELi0E', r'ILi3ELb0ELb0ELb1ELb0ELi2ELi0E', r'ILi3ELb0ELb0ELb1ELb0ELi6ELi0E',
                r'ILi3ELb0ELb0ELb1ELb0ELi1ELi0E', r'ILi3ELb0ELb0ELb1ELb0ELi5ELi0E'):
        starts = [l for l in lines if re.match(r'\s*\.set _ZN\S*gemm_nt_p8_kernel' + pat + r'\S*\.private_seg_size, (\d+)', l)]
        assert len(starts) == 1 and int(starts[0].rsplit(',', 1)[1]) <= 48, (pat, starts)
    # the 8-bit weight gradient: one K-tile = 48 transposing byte reads + 16 scaled MFMAs, nothing else in the vector ALU, no scratch
    body, vg = _kernel(lines, r'gemm_tn_q8_kernelILi1ELi0E')
    steady = [c for c in _loops(body, 16, inner_labels=True) if c['ds_read_b64_tr_b8'] == 48 and sum(c.values()) < 200]
    assert len(steady) == 1 and _valu(steady[0]) <= 6 and steady[0]['v_mfma_scale_f32_32x32x64_f8f6f4'] == 16, steady
    assert vg <= 240 and not any('scratch_' in l for l in body)
    # one-phase TN kernel: 48 transposing reads, 32 MFMAs; 18 vector-ALU instructions before the B base was pinned
    body, vg = _kernel(lines, r'gemm_tn_p1_kernelILb0E')
    steady = [c for c in _loops(body, 32, inner_labels=True) if c['ds_read_b64_tr_b16'] == 48 and sum(c.values()) < 200]
    assert len(steady) == 1 and _valu(steady[0]) <= 6, steady


def test_one_launch_groupnorm_kernels_keep_two_workgroups_per_cu():
    """csrc/conv.hip (GN_FUSED_MAX_SLICES): the wait of the one-launch GroupNorm kernels terminates only while a sample has no more slices than the kernel has resident workgroups
    per XCD (measured: profiles/r06_z7_gn_slices.txt).  The 40-slice guard leans on >= 2 workgroups per CU = 64 per XCD: every instantiation at <= 256 registers (256-thread
    workgroups, 512 registers per SIMD lane), no scratch, LDS a few hundred bytes."""
    text = '\n'.join(_compile('conv.hip'))
    found = 0
    for blk in text.split('- .agpr_count:')[1:]:
        name = re.search(r'\.name:\s+(\S+)', blk).group(1)
        if 'gn_fwd_fused_kernel' in name or 'gn_bwd_fused_kernel' in name:
            found += 1
            val = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1))      # noqa: E731
            assert val('vgpr_count') <= 256 and val('private_segment_fixed_size') == 0 and val('group_segment_fixed_size') <= 64, (name, blk[:400])
    assert found == 5, found                               # forward <8, res, hold>, <16, res>, <16>; backward with / without the stored output
