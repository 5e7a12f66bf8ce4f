"""`-m gpu`: HIP path (through the C ABI) vs the CPU oracle, op level and model level."""
import pytest
import torch

from tests import _cases, _grad_cases, _model_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from eqxvision_amd import _lib
    _lib.load()     # fails loudly if the HIP library was not built / shipped
    _lib.check_device_status()
    yield
    _lib.check_device_status()          # no kernel of the whole module recorded a broken run-time protocol (split-K hand-over)


@pytest.mark.parametrize("name,fn", _cases.all_cases(), ids=[n for n, _ in _cases.all_cases()])
def test_op(name, fn):
    info = fn()
    assert info["ok"], f"{name}: {info}"


@pytest.mark.parametrize("name,fn", _model_cases.all_cases(), ids=[n for n, _ in _model_cases.all_cases()])
def test_model(name, fn):
    info = fn()
    assert info["ok"], f"{name}: {info}"


@pytest.mark.parametrize("name,fn", _grad_cases.all_cases(), ids=[n for n, _ in _grad_cases.all_cases()])
def test_grad(name, fn):
    info = fn()
    assert info["ok"], f"{name}: {info}"


def test_device_status_word_reads_and_clears():
    """mv_device_status: 0 after clean launches; the Python wrapper raises on a recorded failure (none is provoked here: the
    spin-out path needs a partner block that never arrives)."""
    import ctypes
    from eqxvision_amd import _lib
    v = ctypes.c_uint(123)
    assert _lib.load().mv_device_status(0, ctypes.byref(v)) == 0 and v.value == 0
    assert _lib.check_device_status() == 0
