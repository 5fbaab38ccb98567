"""ofighters_amd - MI355X-native batched Ofighters arena engine.

Hot path only: arena step, observation rasteriser, scratch-NN
and bi-head policy forwards as hand-written HIP kernels behind the C-ABI of
include/ofx.h, plus the host-side mirror of the reference's
Battleground / Observation / Action / Agent interface.
"""
from . import _native
from ._native import OfxError, BEHAVIOURS
from .engine import ArenaBatch, DeviceBuffer, pack_actions, ACTION_DTYPE

__all__ = ["ArenaBatch", "DeviceBuffer", "pack_actions", "ACTION_DTYPE", "OfxError", "BEHAVIOURS", "_native"]
