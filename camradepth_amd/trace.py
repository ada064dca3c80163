"""Optional ROCTX ranges (tracing) around the host-side phases of a step -- plan forward / backward segments in
eager mode, graph replays in graph mode -- for `rocprofv3 --marker-trace --kernel-trace`.  Off unless CRD_ROCTX=1: a no-op costs
one attribute lookup.  (Kernel launches inside a captured HIP graph carry no markers: in graph mode the ranges bracket the replays.)"""
import contextlib
import ctypes
import os

_lib = None
if os.environ.get("CRD_ROCTX") == "1":
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        try:
            _lib = ctypes.CDLL(name)
            _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            _lib.roctxRangePushA.restype = ctypes.c_int
            _lib.roctxRangePop.restype = ctypes.c_int
            break
        except (OSError, AttributeError):
            _lib = None

enabled = _lib is not None


@contextlib.contextmanager
def _range(name):
    _lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        _lib.roctxRangePop()


_null = contextlib.nullcontext()


def range(name):     # noqa: A001  (the ROCTX name)
    """with trace.range("backward:dec"): ..."""
    return _range(name) if enabled else _null
