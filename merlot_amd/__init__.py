"""merlot_amd -- MI355X-native MERLOT pretraining hot path (drop-in for rowanz/merlot's MerlotModel path).

Host code: Python on PyTorch-ROCm (device memory, streams, torch.distributed only).
Compute: hand-written gfx950 HIP kernels in libmerlot_hip.so behind the C-ABI of include/merlot_hip.h.
"""
from .config import NeatConfig  # noqa: F401
from .params import ParamStore  # noqa: F401
from .modeling import MerlotModel, model_fn_builder  # noqa: F401
from . import checkpoint, input_pipeline, optimization, sort_story, train  # noqa: F401,E402
