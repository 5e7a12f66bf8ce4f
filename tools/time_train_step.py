"""Wall time of the training step of the reference's gradient test (tests/test_grads.py:35-47: filter_value_and_grad + adam +
apply_updates, model in training mode) on the HIP path, fp32.  The backward half is VALU code written for correctness (section 8 f4
asks "loss is not NaN"), so this is a record of where it stands, not a bench line.
usage: time_train_step.py [MODEL=resnet18] [BATCH=8] [STEPS=5]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import eqxvision_amd as eqv
from oracle import state as S

model = sys.argv[1] if len(sys.argv) > 1 else "resnet18"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
net = getattr(eqv.models, model)(num_classes=10, key=eqv.random.PRNGKey(0))
x = S.synthetic_images(B, 224, seed=0)
y = np.arange(B) % 10
keys = eqv.random.split(eqv.random.PRNGKey(1), B)


@eqv.filter_value_and_grad
def compute_loss(m, xx, yy):
    out = eqv.vmap(m, axis_name="batch")(xx, key=keys)
    return eqv.optim.softmax_cross_entropy(out, eqv.optim.one_hot(yy, 10)).mean()


opt = eqv.optim.adam(1e-3)
st = opt.init(eqv.filter(net, eqv.is_array))
ts, losses = [], []
for i in range(steps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, grads = compute_loss(net, x, y)
    updates, st = opt.update(grads, st)
    net = eqv.apply_updates(net, updates)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0); losses.append(float(loss))
ms = 1e3 * float(np.median(ts[1:]))
print(f"{model} B={B}: training step {ms:.1f} ms ({B / ms * 1e3:.1f} img/s), first step {1e3 * ts[0]:.0f} ms; loss {losses[0]:.4f} -> {losses[-1]:.4f}")
