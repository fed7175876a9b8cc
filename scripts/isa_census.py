"""Round 5: a census of the compiled gfx950 code of the library's kernels (CPU only: hipcc cross-compiles) -- registers, scratch, and for every loop that holds MFMAs
the instruction mix of its body (MFMAs, other vector-ALU instructions, of those the integer additions that are address arithmetic, LDS reads, scalar instructions).
The look that found the pinned-address / accumulator-start changes of round 5; kept as a tool.   python scripts/isa_census.py > profiles/r05_isa_census.txt"""
import os
import re
import subprocess
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'merlot_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def demangle(n):
    try:
        return subprocess.check_output(['c++filt', n], text=True).strip().replace('(anonymous namespace)::', '')
    except Exception:
        return n


for src in ('gemm.hip', 'attention.hip', 'conv_gemm.hip', 'layernorm.hip'):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-w', '-S', '--cuda-device-only', os.path.join(CSRC, src), '-o', out], cwd=CSRC)
        lines = open(out).read().split('\n')
    meta = {}
    for i, l in enumerate(lines):
        m = re.match(r'\s*\.amdhsa_kernel (\S+)', l)
        if m:
            d = {}
            for x in lines[i:i + 90]:
                mm = re.match(r'\s*\.amdhsa_(next_free_vgpr|private_segment_fixed_size|group_segment_fixed_size) (\d+)', x)
                if mm:
                    d[mm.group(1)] = int(mm.group(2))
            meta[m.group(1)] = d
    print(f'==== {src}')
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\S+):', l)
        if not m or m.group(1) not in meta:
            continue
        name = m.group(1)
        body = []
        for x in lines[i + 1:]:
            body.append(x)
            if x.startswith('.Lfunc_end'):
                break
        code = [b.split(';')[0].strip() for b in body]
        n_mfma = sum('v_mfma' in c for c in code)
        d = meta[name]
        short = demangle(name).split('(')[0][:110]
        print(f'{short}: VGPRs {d.get("next_free_vgpr")} scratch {d.get("private_segment_fixed_size")} B, {sum(1 for c in code if c and not c.startswith("."))} instructions, {n_mfma} MFMAs')
        labels = [(k, c.split(':')[0]) for k, c in enumerate(code) if re.match(r'^\.LBB\d+_\d+:', c)]
        for k, lab in labels:
            loop, closed = [], False
            for c in code[k + 1:]:
                if not c or c.startswith('.'):
                    continue
                loop.append(c)
                if (c.startswith('s_cbranch') or c.startswith('s_branch')) and c.endswith(lab):
                    closed = True
                    break
                if len(loop) > 450:
                    break
            nm = sum('v_mfma' in c for c in loop)
            if closed and nm >= 8:
                cnt = Counter(c.split()[0] for c in loop)
                valu = sum(v for kk, v in cnt.items() if kk.startswith('v_') and 'mfma' not in kk)
                addr = sum(v for kk, v in cnt.items() if kk in ('v_add_u32_e32', 'v_xad_u32', 'v_lshl_add_u32', 'v_add3_u32', 'v_lshl_add_u64', 'v_mad_u64_u32', 'v_mul_lo_u32'))
                lds = sum(v for kk, v in cnt.items() if kk.startswith('ds_'))
                salu = sum(v for kk, v in cnt.items() if kk.startswith('s_'))
                print(f'    loop {lab}: {len(loop):4d} instructions | MFMA {nm:3d} | vector-ALU {valu:4d} (integer add / mul: {addr:3d}) | LDS {lds:3d} | scalar {salu:4d} | scratch {sum(v for kk, v in cnt.items() if kk.startswith("scratch_"))}')
