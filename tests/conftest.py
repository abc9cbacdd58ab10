import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import __graft_entry__ as entry  # noqa: E402

entry.load_package()
entry.load_oracle()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def product_lib():
    from godot_whisper_amd import runtime
    lib = runtime.require_gpu()
    runtime.silence_logs(lib)
    return lib


@pytest.fixture(scope="session")
def ref_lib():
    from oracle import reflib
    if not reflib.available():
        pytest.skip("oracle/_ref/libwhisper_ref.so not built (make -C oracle ref needs /root/reference)")
    lib = reflib.lib()
    import ctypes as C
    from godot_whisper_amd import abi
    cb = abi.ggml_log_callback(lambda lvl, txt, ud: None)
    lib.whisper_log_set(C.cast(cb, C.c_void_p), None)
    lib._cb = cb
    return lib


@pytest.fixture(scope="session")
def checker_lib():
    """The checker of the -m gpu parity tests: the compiled reference when its prebuilt library travelled with the snapshot
    (oracle/_ref/libwhisper_ref.so), else None = the CPU restatement (oracle/libwhisper_port.so)."""
    import ctypes as C
    from godot_whisper_amd import abi
    from oracle import port, reflib
    if reflib.available():
        lib = reflib.lib()
        cb = abi.ggml_log_callback(lambda lvl, txt, ud: None)
        lib.whisper_log_set(C.cast(cb, C.c_void_p), None); lib._cb = cb
        return lib
    assert port.available(), "no checker available: build oracle/libwhisper_port.so (python __graft_entry__.py build)"
    return None


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The margin of every floating-point parity bound this run held (tests/stage_compare.py: hold): worst measured value as a
    fraction of its bound — a regression that stays inside a bound still shows here."""
    import json
    import os
    import stage_compare as sc
    if not sc.MARGINS:
        return
    tr = terminalreporter
    tr.write_sep("-", "parity margins: worst measured / bound per kind (1.0 = at the bound)")
    for kind, m in sorted(sc.MARGINS.items()):
        tr.write_line(f"{m['worst_ratio']:6.3f}  {kind}: measured {m['measured']:.3e} of {m['limit']:.3e} over {m['n']} checks")
    out = os.environ.get("WMI_MARGINS_OUT")
    if out:
        with open(out, "w") as f:
            json.dump(sc.MARGINS, f, indent=1, sort_keys=True)
