"""hrnet_hip: the MI355X (gfx950) HighRes-net hot path behind a C ABI (include/hrnet_hip.h).

`binding` wraps libhrnet_hip.so for PyTorch-ROCm tensors; `build.build_library()` compiles it in-tree.
The sibling modules `DeepNetworks.HRNet`, `DeepNetworks.ShiftNet` and `lanczos` re-expose the reference's
module names on top of it so that the reference's train.py / predict.py import them unchanged.
"""
from . import binding  # noqa: F401
from .binding import BF16, F32, HrnetHipError, load_library  # noqa: F401
