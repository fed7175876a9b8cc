"""ctypes binding of libmerlot_hip.so (the C-ABI in include/merlot_hip.h).

The prototypes are parsed from the header itself, so the Python side can never drift from the ABI.
There is NO fallback: if the shared library is missing the import of any op raises, loudly.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'merlot_hip.h')
LIB_PATH = os.path.join(_HERE, 'libmerlot_hip.so')

_CTYPES = {
    'const void*': ctypes.c_void_p, 'void*': ctypes.c_void_p,
    'const float*': ctypes.c_void_p, 'float*': ctypes.c_void_p,
    'const int32_t*': ctypes.c_void_p, 'int32_t*': ctypes.c_void_p, 'int64_t*': ctypes.c_void_p,
    'const uint8_t*': ctypes.c_void_p, 'uint8_t*': ctypes.c_void_p, 'const merlot_image_job_t*': ctypes.c_void_p,
    'const int16_t*': ctypes.c_void_p, 'int16_t*': ctypes.c_void_p,
    'const merlot_jpeg_info_t*': ctypes.c_void_p, 'merlot_jpeg_info_t*': ctypes.c_void_p,
    'int64_t': ctypes.c_int64, 'uint64_t': ctypes.c_uint64, 'int': ctypes.c_int, 'float': ctypes.c_float,
    'double': ctypes.c_double,
    'merlot_stream_t': ctypes.c_void_p,
}


class MerlotHipError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype_str, [(ctype_str, argname), ...])} for every function the header declares."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'\b(int64_t|int|const char\*)\s+(merlot_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        arglist = []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                mm = re.match(r'^(.*?)(\w+)$', a)
                ty = mm.group(1).strip().replace(' *', '*')
                arglist.append((ty, mm.group(2)))
        protos[name] = (ret, arglist)
    return protos


class _Lib(object):
    """One shared library + the header that declares it.  The product uses exactly one instance, `LIB`
    (libmerlot_hip.so / include/merlot_hip.h); tests and scripts make another for libmerlot_probe.so."""

    def __init__(self, header=HEADER, path=None):
        self._dll = None
        self.path = path
        self.header = header
        self.check_abi = True             # scripts/ab_lib.py loads an OLDER build beside the current one and switches this off
        self.protos = parse_header(header)
        self.on_error = []                # callables run before a failed call raises

    def load(self):
        if self._dll is not None:
            return self._dll
        path = self.path or LIB_PATH
        if not os.path.exists(path):
            raise MerlotHipError(
                f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (or merlot_amd/csrc/build.sh). There is no CPU fallback for the product path.")
        dll = ctypes.CDLL(path)
        for name, (ret, args) in self.protos.items():
            fn = getattr(dll, name)
            fn.restype = {'int': ctypes.c_int, 'int64_t': ctypes.c_int64}.get(ret, ctypes.c_char_p)
            fn.argtypes = [_CTYPES[t] for t, _ in args]
        if self.check_abi and 'merlot_abi_version' in self.protos:
            # a stale library of another ABI version would take shifted arguments (v4 inserted workspace pointers in front of
            # `stream`) without any error: refuse it here, before the first call
            m = re.search(r'#define\s+MERLOT_ABI_VERSION\s+(\d+)', open(self.header).read())
            want = int(m.group(1)) if m else None
            got = int(dll.merlot_abi_version())
            if want is not None and got != want:
                raise MerlotHipError(f"{path} implements ABI v{got}, {self.header} declares v{want}: rebuild the library "
                                     f"(merlot_amd/csrc/build.sh)")
        self._dll = dll
        return dll

    def query(self, name, *args):
        """call a function whose return value is data, not a status (e.g. *_workspace_bytes)."""
        return getattr(self.load(), name)(*args)

    def call(self, name, *args):
        dll = self.load()
        rc = getattr(dll, name)(*args)
        if rc != 0:
            msg = dll.merlot_last_error()
            for hook in self.on_error:                   # e.g. ops.reset_workspaces: claim counters a failed launch may have left non-zero
                try:
                    hook()
                except Exception:
                    pass
            raise MerlotHipError(f"{name} failed ({rc}): {msg.decode() if msg else '?'}")


LIB = _Lib()


def call(name, *args):
    LIB.call(name, *args)
