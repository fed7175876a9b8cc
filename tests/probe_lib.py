"""ctypes binding of libmerlot_probe.so (include/merlot_probe.h): hardware-layout probes and experiment helpers.
Test / script infrastructure only -- the product library exports none of these symbols."""
import os

import torch

from merlot_amd.lib import _Lib, _HERE

PROBE = _Lib(header=os.path.join(os.path.dirname(_HERE), 'include', 'merlot_probe.h'),
             path=os.path.join(_HERE, 'libmerlot_probe.so'))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def probe_mfma32(a, b):
    d = torch.empty((64, 16), device=a.device, dtype=torch.float32)
    PROBE.call('merlot_probe_mfma32', a.data_ptr(), b.data_ptr(), d.data_ptr(), _stream())
    return d


def probe_tr16(tile):
    out = torch.empty_like(tile)
    PROBE.call('merlot_probe_tr16', tile.data_ptr(), out.data_ptr(), _stream())
    return out


def cu_hog(blocks, lds_bytes, cycles, sink):
    PROBE.call('merlot_probe_cu_hog', int(blocks), int(lds_bytes), int(cycles), sink.data_ptr(), _stream())
