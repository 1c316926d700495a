"""
zkhip -- host-side mirror of the reference's `dist-primitive` API over libzkhip.so, the
MI355X-native (gfx950) implementation of its MSM / sumcheck / product-tree hot path.
"""
from ._lib import LIB_PATH, build, lib  # noqa: F401
from .api import Ctx, DeviceBuffer, MsmLengthError, Srs, ZkError, comm_init_all  # noqa: F401
