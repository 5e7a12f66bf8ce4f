"""ms per TRAINING-MODE ViT-B/16 forward with every Dropout live (drop_rate / attn_drop_rate / drop_path_rate = 0.1), next to the
inference forward of the same weights (eager launches both).  usage: time_vit_dropout.py [B]"""
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


net = eqv.models.vit_base(num_classes=1000, drop_rate=0.1, attn_drop_rate=0.1, drop_path_rate=0.1, key=eqv.random.PRNGKey(1))
inf = eqv.tree_inference(net, True)
trn = eqv.tree_inference(net, False)
print(f"vit_base B={B}: inference (eager) {clock(lambda: eqv.vmap(inf, axis_name='batch')(x, key=keys)):.2f} ms, "
      f"training mode with live dropouts {clock(lambda: eqv.vmap(trn, axis_name='batch')(x, key=keys)):.2f} ms")
