"""The persistent attention kernels (csrc/attention_pp.inc) count their own memory waits: every LDS-DMA request and every own-row
load is inline assembly the compiler does not see, and the `s_waitcnt vmcnt(N)` immediates are derived by hand from the number of
vector-memory operations a wave issues per item.  That only holds while the COMPILED code keeps three properties, which this test
reads off the gfx950 assembly (CPU only: hipcc cross-compiles):
  1. no scratch: a spill is a vector-memory operation the counts do not know (and a reload is waited for with a drain);
  2. no compiler-inserted `s_waitcnt vmcnt` inside the kernels (hipcc drains the queue for a pending LDS-DMA it can see in front of
     every ds_read_b64_tr_b16, and for any ordinary load beside one -- the reason the requests are assembly) -- except the loader
     wave's waits for its own compiler-issued atomics (the two item claims of the prologue, the departure ticket at the end);
  3. the destination registers of a hand-issued load are neither read nor written between the load and the kernel's own wait
     (round 5: with issue and use on two sides of the loop's back edge hipcc moved the registers while the load was in flight)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'merlot_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def _regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def check_kernel(name, body):
    """-> list of violations for one kernel's assembly lines.  (attn_fwd_ppm_kernel's loader fetches the validity bytes with ordinary
    loads, consumed at once, before any request is outstanding: the compiler's own waits for those are expected there.)"""
    own_waits_only = 'ppm_kernel' not in name
    bad, in_asm, pending = [], False, {}
    n_dma = n_wait = 0
    own_atomic = False                                   # a compiler-issued returning atomic since the last wait (claim / departure)
    for i, ln in enumerate(body):
        if '#ASMSTART' in ln:
            in_asm = True
            continue
        if '#ASMEND' in ln:
            in_asm = False
            continue
        code = ln.split(';')[0].strip()
        if not code or code.endswith(':') or code.startswith('.'):
            continue
        if 'scratch_' in code:
            bad.append(f'{name}: scratch access {code!r}')
        if code.startswith('global_atomic_') and not in_asm:
            own_atomic = True
        if 's_waitcnt' in code and 'vmcnt' in code:
            if not in_asm and own_waits_only and not own_atomic:
                bad.append(f'{name}: compiler-inserted {code!r} (line {i})')
            n_wait += 1
            pending = {}
            own_atomic = False
            continue
        m = re.match(r'global_atomic_add (v\d+), ', code)
        if m and in_asm:                                 # the hand-issued item claim: its answer register is in flight until the next wait
            for r in _regs(m.group(1)):
                pending[r] = i
            continue
        if re.match(r'buffer_load_dword(x4)? ', code) and code.endswith('lds'):
            assert in_asm, f'{name}: an LDS-DMA the compiler can see: {code!r}'
            n_dma += 1
            continue
        m = re.match(r'buffer_load_dwordx4 (v\[\d+:\d+\]), ', code)
        if m:
            assert in_asm
            for r in _regs(m.group(1)):
                pending[r] = i
            continue
        if pending:
            used = _regs(code) & set(pending)
            if used:
                bad.append(f'{name}: line {i} {code!r} touches registers {sorted(used)} of a load in flight')
    assert n_dma > 0 and n_wait > 0, f'{name}: not a persistent attention kernel?'
    return bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not available')
def test_persistent_attention_kernels_keep_their_wait_counts():
    tmp = tempfile.mkdtemp(prefix='pp_isa_')
    try:
        out = os.path.join(tmp, 'attention.s')
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-function', '-Wno-unused-variable',
                               '-w', '-S', '--cuda-device-only', os.path.join(CSRC, 'attention.hip'), '-o', out], cwd=CSRC)
        kernels, cur = {}, None
        for ln in open(out).read().split('\n'):
            m = re.match(r'^(_ZN\S*attn_(fwd|bwd)_ppm?_kernel\S*):', ln)
            if m:
                cur = m.group(1)
                kernels[cur] = []
            if cur is not None:
                kernels[cur].append(ln)
                if 's_endpgm' in ln and not ln.strip().startswith(';'):
                    pass
            if cur is not None and ln.startswith('.Lfunc_end'):
                cur = None
        assert len(kernels) == 13, sorted(kernels)             # forward and backward, 3 .. 7 key chunks; masked forward, 9 .. 11
        bad = [b for name, body in kernels.items() for b in check_kernel(name, body)]
        assert not bad, '\n'.join(bad)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
