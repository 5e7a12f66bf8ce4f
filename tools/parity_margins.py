"""Print the measured error and the limit of the parity cases DESIGN.md section 4 quotes (full-size models, both compute modes, the
exported factories), one line per case.   usage: python tools/parity_margins.py [substring ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _model_cases as MC

WANT = sys.argv[1:] or ["golden/committed_", "_fp32", "model/factory_", "_full_config", "B256", "B128", "layer1_recompute", "trained_scale",
                        "resnet50_B2", "vit_base_B2", "swin_t_B1", "alexnet_B4"]
for name, fn in MC.all_cases():
    if not any(w in name for w in WANT):
        continue
    info = fn()
    keys = ("ok", "err", "lim", "refmax", "l2_err", "err_plan_off", "plan_vs_off", "guard")
    print(f"{name:62s} " + "  ".join(f"{k}={info[k]:.3e}" if isinstance(info.get(k), float) else f"{k}={info.get(k)}" for k in keys if k in info), flush=True)
