#!/usr/bin/env python
"""On-GPU parity sweep of every C-ABI op against the CPU oracle (numpy).  Runs ALL cases (does not
stop at the first failure) and writes gpurun_out/gpu_check.json.  Used during development; the
pytest `-m gpu` suite runs the same cases through `tests/_cases.py`."""
import json
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _cases, _grad_cases, _model_cases  # noqa: E402


def main():
    only = sys.argv[1:]                       # substrings; a case runs if it contains any of them
    results = []
    nfail = 0
    for name, fn in _cases.all_cases() + _model_cases.all_cases() + _grad_cases.all_cases():
        if only and not any(o in name for o in only):
            continue
        t = time.time()
        try:
            info = fn()
            ok = bool(info.pop("ok"))
            results.append({"case": name, "ok": ok, "s": round(time.time() - t, 3), **info})
        except Exception as e:  # noqa: BLE001
            ok = False
            results.append({"case": name, "ok": False, "error": f"{type(e).__name__}: {e}",
                            "tb": traceback.format_exc()[-1500:]})
        nfail += (not ok)
        print(("PASS " if ok else "FAIL ") + name + "  " + json.dumps({k: v for k, v in results[-1].items()
                                                                       if k not in ("case", "ok", "tb")}), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/gpu_check.json", "w") as f:
        json.dump(results, f, indent=1)
    print(f"{len(results) - nfail}/{len(results)} passed")
    return 1 if nfail else 0


if __name__ == "__main__":
    sys.exit(main())
