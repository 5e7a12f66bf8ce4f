"""ms per TRAINING-MODE ViT-B/16 forward with every Dropout live (drop_rate / attn_drop_rate / drop_path_rate = 0.1), next to the
inference forward of the same weights (eager launches both); `swin_t`: dropout / attention_dropout = 0.1 (drawn in every mode,
swin.py:17-20) against the same architecture without them.  usage: time_vit_dropout.py [B] [vit_base|swin_t]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eqxvision_amd as eqv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eqv.set_compute_dtype("bf16")
x = torch.rand((B, 3, 224, 224), dtype=torch.float32).cuda()
keys = eqv.random.split(eqv.random.PRNGKey(0), B)


def clock(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


if len(sys.argv) > 2 and sys.argv[2] == "swin_t":
    import warnings
    warnings.simplefilter("ignore")
    plain = eqv.tree_inference(eqv.models.swin_t(key=eqv.random.PRNGKey(1)), True)
    net = eqv.models.swin_t(dropout=0.1, attention_dropout=0.1, key=eqv.random.PRNGKey(1))
    inf, trn = eqv.tree_inference(net, True), eqv.tree_inference(net, False)
    run = lambda m: clock(lambda: eqv.vmap(m, axis_name='batch')(x, key=keys))
    print(f"swin_t B={B}: no dropout, inference (eager) {run(plain):.2f} ms; dropout / attention_dropout = 0.1: inference mode "
          f"{run(inf):.2f} ms, training mode (+ MLP dropouts, stochastic depth) {run(trn):.2f} ms")
    sys.exit(0)
net = eqv.models.vit_base(num_classes=1000, drop_rate=0.1, attn_drop_rate=0.1, drop_path_rate=0.1, key=eqv.random.PRNGKey(1))
inf = eqv.tree_inference(net, True)
trn = eqv.tree_inference(net, False)
print(f"vit_base B={B}: inference (eager) {clock(lambda: eqv.vmap(inf, axis_name='batch')(x, key=keys)):.2f} ms, "
      f"training mode with live dropouts {clock(lambda: eqv.vmap(trn, axis_name='batch')(x, key=keys)):.2f} ms")
