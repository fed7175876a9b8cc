"""The persistent attention kernels (csrc/attention_pp.inc) count their own memory waits: every LDS-DMA request and every own-row
load is inline assembly the compiler does not see, and the `s_waitcnt vmcnt(N)` immediates are derived by hand from the number of
vector-memory operations a wave issues per item.  That only holds while the COMPILED code keeps three properties, which this test
reads off the gfx950 assembly (CPU only: hipcc cross-compiles):
  1. no scratch: a spill is a vector-memory operation the counts do not know (and a reload is waited for with a drain);
  2. no compiler-inserted `s_waitcnt vmcnt` inside the kernels (hipcc drains the queue for a pending LDS-DMA it can see in front of
     every ds_read_b64_tr_b16, and for any ordinary load beside one -- the reason the requests are assembly) -- except the loader
     wave's waits for its own compiler-issued atomics (the two item claims of the prologue, the departure ticket at the end);
  3. the destination registers of a hand-issued load are neither read nor written between the load and the kernel's own wait
     (round 5: with issue and use on two sides of the loop's back edge hipcc moved the registers while the load was in flight);
  4. (round 6) the LDS-DMA statements write m0 inside their assembly.  m0 cannot be named as a clobber (hipcc: "reserved
     register ... may not be preserved"), so the invariant is checked instead: NO compiler-generated instruction of these kernels reads
     or writes m0 -- nothing of the compiler's can be live in it across a request;
  5. every hand-written `s_waitcnt vmcnt(N)` leaves at most as many operations in flight as the wave has issued since the kernel's
     previous full drain on that straight-line path: an immediate larger than the operations in front of it would wait for nothing."""
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
        if not in_asm and re.search(r'\bm0\b', code):
            bad.append(f'{name}: compiler-generated use of m0 {code!r} (line {i}): the LDS-DMA assembly overwrites it')
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


def check_counted_wait(name, body):
    """The ViT backward's `pp_wait_vm<4>` (attention_pp.inc, "A": own K / V rows landed; younger: the dQ stores of item k - 1) is only right if the
    wave issued EXACTLY the four dQ stores -- and nothing else -- between its eight own-row loads and that wait: with fewer younger operations
    vmcnt(4) would leave own-row loads in flight, with an extra load the count would be off as well.  Read off the listing (loads, stores and
    the wait sit in that order in the compiled loop; if a future compiler lays the loop out differently this fails and has to be re-derived)."""
    in_asm, seq = False, []
    for ln in body:
        if '#ASMSTART' in ln:
            in_asm = True
            continue
        if '#ASMEND' in ln:
            in_asm = False
            continue
        code = ln.split(';')[0].strip()
        if in_asm and re.match(r's_waitcnt vmcnt\(4\)', code):
            seq.append('WAIT4')
        elif in_asm and re.match(r'buffer_load_dwordx4 v\[', code) and not code.endswith('lds'):
            seq.append('OWN')
        elif re.match(r'(buffer|global|flat|scratch)_(load|store|atomic)', code):
            seq.append(('ASM ' if in_asm else 'CC ') + code.split()[0])
    bad = []
    waits = [k for k, e in enumerate(seq) if e == 'WAIT4']
    if len(waits) != 1:
        return [f'{name}: {len(waits)} hand-written vmcnt(4) waits, expected 1']
    k = waits[0] - 1
    between = []
    while k >= 0 and seq[k] != 'OWN':
        between.append(seq[k])
        k -= 1
    n_own = 0
    while k >= 0 and seq[k] == 'OWN':
        n_own += 1
        k -= 1
    if n_own != 8:
        bad.append(f'{name}: {n_own} own-row loads in front of vmcnt(4), expected 8')
    if between != ['CC global_store_dwordx4'] * 4:
        bad.append(f'{name}: vector-memory operations between the own-row loads and vmcnt(4): {between[::-1]} (expected the four dQ stores)')
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
        bad += [b for name, body in kernels.items() if 'attn_bwd_pp_kernel' in name for b in check_counted_wait(name, body)]
        assert not bad, '\n'.join(bad)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _fused_ch_sequence(body):
    """text-order sequence of one kernel's vector-memory events: D = LDS-DMA request, g = load, s = store, a = atomic, W<n> / w<n> = s_waitcnt vmcnt(n) inside
    / outside hand-written assembly, | = barrier; plus the violations check_kernel's first rules would report."""
    seq, bad, in_asm = [], [], False
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
            bad.append(f'scratch access {code!r}')
        if not in_asm and re.search(r'\bm0\b', code):
            bad.append(f'compiler-generated use of m0 {code!r} (line {i})')
        if re.match(r'buffer_load_dword(x4)? ', code) and code.endswith('lds'):
            if not in_asm:
                bad.append(f'an LDS-DMA the compiler can see: {code!r}')
            seq.append('D')
        elif code.startswith('global_load'):
            seq.append('g')
        elif code.startswith('global_store'):
            seq.append('s')
        elif code.startswith('global_atomic'):
            seq.append('a')
        elif code.startswith('s_waitcnt') and 'vmcnt' in code:
            seq.append(('W' if in_asm else 'w') + re.search(r'vmcnt\((\d+)\)', code).group(1))
        elif code.startswith('s_barrier'):
            seq.append('|')
    return seq, bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not available')
def test_chunked_fused_backward_keeps_its_requests_out_of_the_compilers_sight():
    """attn_bwd_fused_kernel<512, .., CH = true> (attention_fb.inc, round 6): the K | V and Q | dO images arrive in groups of 128 rows under the passes that read
    them.  That holds while (1) every request is assembly and nothing of the compiler's touches m0, (2) a wave issues FOUR requests per group (the wait ladder's
    immediates 12 / 8 / 4 / 0 count them), (3) every ordinary load is issued where no later group is outstanding or is the youngest operation (in front of the
    first walk: behind group 0 and drained by a wait the compiler sees; inside it: at chunk 4, drained -- visibly -- behind the walk), so that (4) the compiler
    inserts NO wait of its own between pass 1's stores and the refill requests: the first version drained those stores on every wave there."""
    tmp = tempfile.mkdtemp(prefix='fb_isa_')
    try:
        out = os.path.join(tmp, 'attention.s')
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-w', '-S', '--cuda-device-only',
                               os.path.join(CSRC, 'attention.hip'), '-o', out], cwd=CSRC)
        kernels, cur = {}, None
        for ln in open(out).read().split('\n'):
            m = re.match(r'^(_ZN\S*attn_bwd_fused_kernelILi512E\S*Li2ELb1EEE\S*):', ln)
            if m:
                cur = m.group(1)
                kernels[cur] = []
            if cur is not None:
                kernels[cur].append(ln)
                if ln.startswith('.Lfunc_end'):
                    cur = None
        assert len(kernels) == 3, sorted(kernels)              # unmasked, masked, masked + attention log
        for name, body in kernels.items():
            seq, bad = _fused_ch_sequence(body)
            assert not bad, name + ': ' + '; '.join(bad)
            s = ' '.join(seq)
            # three request sites (group 0 of K | V; its later groups, one rolled loop; the refill, one rolled loop), four requests each
            assert seq.count('D') == 12 and s.count('D D D D') == 3, (name, s)
            # group 0, then ordinary loads only, then the wait the compiler sees, then the later groups and the barrier that opens pass 1
            assert re.search(r'D D D D( g)+ w0 D D D D \|', s), (name, s)
            # every counted wait is the whole ladder, hand-written (the barrier behind it may sit elsewhere in the listing: block layout)
            assert 'W' in s and all(x in ('W12', 'W8', 'W4', 'W0') for x in seq if x.startswith('W')), (name, s)
            assert s.count('W12 W8 W4 W0') == s.count('W12') == s.count('W0') >= 3, (name, s)
            # nothing of the compiler's between the loads of the second block (drained visibly behind the walk), pass 1's stores and the refill
            m = re.search(r'g w0( s)+ \| D D D D W12', s)
            assert m, (name, s)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
