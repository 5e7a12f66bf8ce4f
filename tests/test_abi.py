"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol `include/eqxvision_amd.h`
declares with the argument count the ctypes binding uses (no compute calls without a GPU)."""
import os
import re

from eqxvision_amd import _lib

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "eqxvision_amd.h")


def _declared():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:int|int64_t|const char\*)\s+(mv_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("void", "") else len(args.split(","))
    return out


def test_header_symbols_exported_and_bound(built_lib):
    decl = _declared()
    assert len(decl) >= 25
    for name, nargs in decl.items():
        assert hasattr(built_lib, name), f"{name} declared in the header but not exported by the library"
        assert name in _lib.PROTOTYPES, f"{name} has no ctypes prototype"
        assert len(_lib.PROTOTYPES[name]) == nargs, f"{name}: header has {nargs} args, binding {len(_lib.PROTOTYPES[name])}"
    for name in _lib.PROTOTYPES:
        assert name in decl, f"{name} bound in _lib.py but not declared in the header"


def test_abi_version_and_error_channel(built_lib):
    assert built_lib.mv_abi_version() == _lib.ABI_VERSION
    assert built_lib.mv_last_error() is not None
    _lib.set_flag("force_generic", 1)
    assert _lib.get_flag("force_generic") == 1
    _lib.set_flag("force_generic", 0)


def test_argument_errors_do_not_need_a_gpu(built_lib):
    # NULL pointers are rejected before any HIP call: negative MV_E_INVALID + message, no abort
    rc = built_lib.mv_linear_fwd(None, None, None, None, None, None, 4, 4, 4, 0, 1, 1, None)
    assert rc == -1 and b"NULL" in built_lib.mv_last_error()
    rc = built_lib.mv_swin_window_attn_fwd(1, 1, 1, 1, 13, 13, 32, 2, 7, 7, 0, 0, 1, None)
    assert rc == -1 and b"multiple of the window" in built_lib.mv_last_error()


def test_comm_argument_errors_do_not_need_a_gpu(built_lib):
    # no communicator yet: the collective refuses loudly instead of copying locally (and librccl is not even loaded)
    assert built_lib.mv_comm_size() == 0 and built_lib.mv_comm_rank() == -1
    rc = built_lib.mv_allgather(1, 1, 16, None)
    assert rc == -1 and b"mv_comm_init" in built_lib.mv_last_error()
    rc = built_lib.mv_comm_init(3, 2, 1)
    assert rc == -1 and b"bad rank" in built_lib.mv_last_error()
    assert built_lib.mv_comm_destroy() == 0
