"""Layer-level tracing ranges (tracing row): with EQV_ROCTX=1 every `Module.__call__` and every C-ABI call is
bracketed by a roctx range (`libroctx64.so`: roctxRangePushA / roctxRangePop), so that `rocprofv3 --marker-trace` shows which module
issued which launch.  Host-side markers: meaningful for eager forwards and for the recording call of `filter_jit`; a hipGraph replay
has no host activity to mark.  Off (the default) costs one attribute test per call."""
import ctypes
import functools
import os

enabled = os.environ.get("EQV_ROCTX", "0") not in ("", "0")
_lib = None
depth = 0            # open ranges (tests)
pushed = 0           # ranges opened so far (tests)


def _load():
    global _lib
    if _lib is None:
        for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so"):
            try:
                _lib = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _lib is None:
            raise RuntimeError("EQV_ROCTX=1 but libroctx64.so cannot be loaded")
        _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
        _lib.roctxRangePushA.restype = ctypes.c_int
        _lib.roctxRangePop.restype = ctypes.c_int
    return _lib


def push(name: str):
    global depth, pushed
    _load().roctxRangePushA(name.encode())
    depth += 1
    pushed += 1


def pop():
    global depth
    _load().roctxRangePop()
    depth -= 1


def wrap_call(cls_name, fn):
    """`Module.__call__` of a sub-class, bracketed by a range named after the class (no-op unless tracing is on at call time)."""
    @functools.wraps(fn)
    def traced(self, *a, **kw):
        if not enabled:
            return fn(self, *a, **kw)
        push(cls_name)
        try:
            return fn(self, *a, **kw)
        finally:
            pop()
    return traced
