"""CPU: `__graft_entry__.build()`'s recipe from a tree WITHOUT build/ and without any .so -- every object is compiled by hipcc /
g++ for gfx950 here (cross-compilation, no GPU), the two libraries link, and the product library exports every function the header
declares.  (build.sh is timestamp-incremental and the built libraries travel to the GPU box although git ignores them, so a stale
in-tree .so would otherwise never be noticed: VERDICT r3 weak #9.)"""
import ctypes
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_build_from_a_clean_tree_compiles_every_object(tmp_path):
    from merlot_amd.lib import parse_header
    dst = tmp_path / 'repo'
    (dst / 'merlot_amd').mkdir(parents=True)
    shutil.copytree(os.path.join(ROOT, 'include'), dst / 'include')
    shutil.copytree(os.path.join(ROOT, 'merlot_amd', 'csrc'), dst / 'merlot_amd' / 'csrc',
                    ignore=shutil.ignore_patterns('build', 'build_exp', '*.o', '*.so', '.pytest_cache'))
    assert not list(dst.rglob('*.o')) and not list(dst.rglob('*.so'))
    env = dict(os.environ, HIPCC=os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'))
    out = subprocess.run(['bash', str(dst / 'merlot_amd' / 'csrc' / 'build.sh')], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    man = (dst / 'merlot_amd' / 'csrc' / 'build' / 'BUILD_MANIFEST').read_text().split()
    objs = dict(zip(man[1::2], man[0::2]))
    srcs = [f[:-4] for f in os.listdir(os.path.join(ROOT, 'merlot_amd', 'csrc')) if f.endswith('.hip') and f != 'probe.hip']
    srcs += [f[:-4] for f in os.listdir(os.path.join(ROOT, 'merlot_amd', 'csrc')) if f.endswith('.cpp')]
    assert sorted(objs) == sorted(s + '.o' for s in srcs), (sorted(objs), sorted(srcs))       # every source is part of the library
    assert all(v == 'compiled' for v in objs.values()), objs
    lib = dst / 'merlot_amd' / 'libmerlot_hip.so'
    assert lib.exists() and (dst / 'merlot_amd' / 'libmerlot_probe.so').exists()
    dll = ctypes.CDLL(str(lib))
    for name in parse_header(os.path.join(ROOT, 'include', 'merlot_hip.h')):
        assert hasattr(dll, name), name
    # the code object really is gfx950
    dump = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '--offloading', str(lib)], capture_output=True, text=True, cwd=str(tmp_path))
    assert 'gfx950' in dump.stdout
