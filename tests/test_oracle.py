"""CPU: the numpy oracle (`oracle/np_ops.py`, `oracle/models.py`) against (a) torch.nn.functional -- the
ground truth the reference's own tests use (tests/test_models/test_resnet.py:24) -- and (b) the committed
golden vectors.  The reference itself cannot be imported here (no jax / equinox): parity is otherwise unpinned."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import models as OM
from oracle import np_ops as O
from oracle import state as S
from oracle import torch_ref as TR

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "hotpath_small.npz"))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(1e-12, np.abs(b).max())


def rng(seed=0):
    return np.random.Generator(np.random.PCG64(seed))


@pytest.mark.parametrize("cfg", [
    dict(C=3, K=8, R=7, S=7, stride=2, pad=3, H=23, W=19),
    dict(C=16, K=12, R=3, S=3, stride=1, pad=1, H=9, W=9),
    dict(C=16, K=12, R=3, S=3, stride=2, pad=2, dil=2, H=11, W=13),
    dict(C=8, K=8, R=3, S=3, stride=1, pad=1, groups=4, H=7, W=7),
    dict(C=6, K=10, R=1, S=1, stride=2, pad=0, H=8, W=8),
    dict(C=3, K=5, R=11, S=11, stride=4, pad=2, H=35, W=35),
])
def test_conv2d_matches_torch(cfg):
    r = rng(1)
    g = cfg.get("groups", 1)
    x = r.standard_normal((cfg["C"], cfg["H"], cfg["W"])).astype(np.float32)
    w = r.standard_normal((cfg["K"], cfg["C"] // g, cfg["R"], cfg["S"])).astype(np.float32)
    b = r.standard_normal((cfg["K"], 1, 1)).astype(np.float32)
    got = O.conv2d(x, w, b, cfg["stride"], cfg["pad"], cfg.get("dil", 1), g)
    ref = F.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), torch.from_numpy(b.reshape(-1)), cfg["stride"],
                   cfg["pad"], cfg.get("dil", 1), g)[0].numpy()
    assert got.shape == ref.shape and rel(got, ref) < 1e-5


def test_pool_norm_activation_match_torch():
    r = rng(2)
    x = r.standard_normal((5, 13, 12)).astype(np.float32)
    t = torch.from_numpy(x)[None]
    assert rel(O.maxpool2d(x, 3, 2, 1), F.max_pool2d(t, 3, 2, 1)[0].numpy()) == 0
    assert rel(O.maxpool2d(x, 3, 2, 0), F.max_pool2d(t, 3, 2, 0)[0].numpy()) == 0
    x2 = r.standard_normal((4, 12, 6)).astype(np.float32)
    assert rel(O.adaptive_avgpool2d(x2, (6, 6)), F.adaptive_avg_pool2d(torch.from_numpy(x2)[None], (6, 6))[0].numpy()) < 1e-6
    assert rel(O.adaptive_avgpool2d(x2, (1, 1)), x2.mean((1, 2), keepdims=True)) < 1e-6
    rows = r.standard_normal((7, 20)).astype(np.float32)
    g_, b_ = r.uniform(0.5, 1.5, 20).astype(np.float32), r.standard_normal(20).astype(np.float32)
    ref = F.layer_norm(torch.from_numpy(rows), (20,), torch.from_numpy(g_), torch.from_numpy(b_), 1e-5).numpy()
    assert rel(O.layernorm_rows(rows, g_, b_), ref) < 1e-5
    assert rel(O.layernorm(rows[0], g_, b_), ref[0]) < 1e-5
    v = (3 * r.standard_normal(1000)).astype(np.float32)
    assert rel(O.gelu_tanh(v), F.gelu(torch.from_numpy(v), approximate="tanh").numpy()) < 1e-6
    assert np.abs(O.gelu_tanh(v) - F.gelu(torch.from_numpy(v)).numpy()).max() > 1e-4      # NOT the erf form
    assert rel(O.softmax(rows), torch.from_numpy(rows).softmax(-1).numpy()) < 1e-6
    m, var = r.standard_normal(5).astype(np.float32), r.uniform(0.5, 1.5, 5).astype(np.float32)
    gb, bb = r.uniform(0.5, 1.5, 5).astype(np.float32), r.standard_normal(5).astype(np.float32)
    ref = F.batch_norm(t, torch.from_numpy(m), torch.from_numpy(var), torch.from_numpy(gb), torch.from_numpy(bb), False, 0., 1e-5)[0].numpy()
    assert rel(O.batchnorm_inference(x, gb, bb, m, var), ref) < 1e-5


def test_adaptive_avgpool_equinox_rule_when_not_divisible():
    # first n%t outputs use chunks of n//t+1, the rest n//t (differs from torch)
    x = np.arange(7, dtype=np.float32).reshape(1, 7, 1)
    out = O.adaptive_avgpool2d(x, (3, 1)).reshape(-1)
    np.testing.assert_allclose(out, [np.mean([0, 1, 2]), np.mean([3, 4]), np.mean([5, 6])])


def test_patch_embed_is_a_gemm_and_attention_matches_sdpa():
    r = rng(3)
    img = r.standard_normal((3, 32, 32)).astype(np.float32)
    w = r.standard_normal((24, 3, 8, 8)).astype(np.float32)
    b = r.standard_normal((24, 1, 1)).astype(np.float32)
    tok = O.patch_embed(img, w, b, 8)
    gemm = img.reshape(3, 4, 8, 4, 8).transpose(1, 3, 0, 2, 4).reshape(16, 192) @ w.reshape(24, 192).T + b.reshape(-1)
    assert tok.shape == (16, 24) and rel(tok, gemm) < 1e-5
    x = r.standard_normal((9, 32)).astype(np.float32)
    wq, bq = r.standard_normal((96, 32)).astype(np.float32) / 6, r.standard_normal(96).astype(np.float32)
    wp, bp = r.standard_normal((32, 32)).astype(np.float32) / 6, r.standard_normal(32).astype(np.float32)
    y, attn = O.vit_attention(x, wq, bq, wp, bp, 4)
    assert y.shape == (9, 32) and attn.shape == (1, 4, 9, 9)
    qkv = torch.from_numpy(x @ wq.T + bq).reshape(9, 3, 4, 8).permute(1, 2, 0, 3)
    ref = F.scaled_dot_product_attention(qkv[0][None], qkv[1][None], qkv[2][None])[0].permute(1, 0, 2).reshape(9, 32)
    assert rel(y, ref.numpy() @ wp.T + bp) < 1e-5


def test_bf16_round_matches_torch():
    v = (rng(4).standard_normal(10000) * 100).astype(np.float32)
    ref = torch.from_numpy(v).to(torch.bfloat16).float().numpy()
    np.testing.assert_array_equal(O.bf16_round(v), ref)


def test_models_numpy_vs_torch_and_golden():
    x = S.synthetic_images(2, 64, seed=0)
    sd = S.resnet_state(1, "bottleneck", (1, 1, 1, 1), 10)
    a = O.vmap(lambda im: OM.resnet_forward(sd, im, "bottleneck", (1, 1, 1, 1)))(x)
    assert rel(a, TR.resnet_forward(sd, x, "bottleneck", (1, 1, 1, 1)).numpy()) < 1e-5
    assert rel(a, GOLD["resnet_bottleneck_1111_64px"]) < 1e-5
    sd = S.resnet_state(1, "basic", (2, 2, 2, 2), 10)
    a = O.vmap(lambda im: OM.resnet_forward(sd, im, "basic", (2, 2, 2, 2)))(x)
    assert rel(a, GOLD["resnet18_64px"]) < 1e-5
    x32 = S.synthetic_images(3, 32, seed=0)
    sd = S.vit_state(1, 32, 8, 64, 2, 2, 4, 10)
    a = O.vmap(lambda im: OM.vit_forward(sd, im, 8, 2, 2))(x32)
    assert rel(a, GOLD["vit_32px_p8_d64_h2_depth2"]) < 1e-5
    a = O.vmap(lambda im: OM.vit_last_self_attention(sd, im, 8, 2, 2))(x32[:2])
    assert a.shape == (2, 1, 2, 17, 17) and rel(a, GOLD["vit_32px_last_attn"]) < 1e-5
    x56 = S.synthetic_images(2, 56, seed=0)
    sd = S.swin_state(1, (4, 4), 32, (2, 2), (2, 4), (7, 7), 4.0, 10)
    a = O.vmap(lambda im: OM.swin_forward(sd, im, (4, 4), (2, 2), (2, 4), (7, 7)))(x56)
    assert rel(a, GOLD["swin_56px_e32_d22"]) < 1e-5
    assert rel(a, TR.swin_forward(sd, x56, (4, 4), (2, 2), (2, 4), (7, 7)).numpy()) < 1e-5


def test_alexnet_golden_and_bf16_emulation_is_close():
    x = S.synthetic_images(2, 224, seed=0)
    sd = S.alexnet_state(1, 1000)
    lg = O.vmap(lambda im: OM.alexnet_forward(sd, im))(x)
    assert rel(lg[:, :16], GOLD["alexnet_224_logits_head16"]) < 1e-4
    assert rel(np.linalg.norm(lg, axis=1), GOLD["alexnet_224_logits_l2"]) < 1e-5
    emu = O.vmap(lambda im: OM.alexnet_forward(sd, im, bf16=True))(x[:1])
    assert np.abs(emu - lg[:1]).max() <= 1e-2 * max(1.0, np.abs(lg).max())


def test_full_size_vit_and_swin_checksums():
    """the numpy oracle reproduces the committed full-size checksums (first 16 logits + L2 norm of the
    logits, written by the torch restatement) of vit_base and swin_t -- first image of the B = 2 fixture."""
    x = S.synthetic_images(2, 224, seed=0)[:1]
    lg = O.vmap(lambda im: OM.vit_forward(S.vit_state(1), im))(x)
    assert rel(lg[:, :16], GOLD["vit_base_224_logits_head16"][:1]) < 1e-4
    assert rel(np.linalg.norm(lg, axis=1), GOLD["vit_base_224_logits_l2"][:1]) < 1e-5
    lg = O.vmap(lambda im: OM.swin_forward(S.swin_state(1), im))(x)
    assert rel(lg[:, :16], GOLD["swin_t_224_logits_head16"][:1]) < 1e-4
    assert rel(np.linalg.norm(lg, axis=1), GOLD["swin_t_224_logits_l2"][:1]) < 1e-5


def test_swin_shift_mask_and_window_edge_cases():
    r = rng(5)
    C, H, heads = 16, 14, 2
    x = r.standard_normal((C, H, H)).astype(np.float32)
    wq, wp = r.standard_normal((3 * C, C)).astype(np.float32) / 4, r.standard_normal((C, C)).astype(np.float32) / 4
    bq, bp = r.standard_normal(3 * C).astype(np.float32), r.standard_normal(C).astype(np.float32)
    bias = r.standard_normal((heads, 49, 49)).astype(np.float32)
    sd = {"p.attn.qkv.weight": torch.from_numpy(wq), "p.attn.qkv.bias": torch.from_numpy(bq),
          "p.attn.proj.weight": torch.from_numpy(wp), "p.attn.proj.bias": torch.from_numpy(bp)}
    for shift in ([3, 3], [0, 0]):
        got = O.shifted_window_attention(x, wq, wp, bias, [7, 7], heads, shift, bq, bp)
        # independent torch restatement with an explicit (heads, n, n) bias through a fake table
        table = torch.from_numpy(bias.transpose(1, 2, 0).reshape(49 * 49, heads).copy())
        sd2 = dict(sd, **{"p.attn.relative_position_bias_table": table,
                          "p.attn.relative_position_index": torch.arange(49 * 49)})
        ref = TR._swin_attn(sd2, torch.from_numpy(x).permute(1, 2, 0)[None], "p", heads, [7, 7], shift)[0].permute(2, 0, 1).numpy()
        assert rel(got, ref) < 1e-5
    with pytest.raises(ValueError):
        O.shifted_window_attention(x[:, :13], wq, wp, bias, [7, 7], heads, [0, 0], bq, bp)


def test_resize_bilinear_restatement_matches_torch_upsampling():
    """jax.image.resize "bilinear" (restated from jax._src.image.scale) == torch align_corners=False when up-sampling."""
    import torch
    import torch.nn.functional as F
    from oracle import np_ops as O
    rng = np.random.default_rng(0)
    for h, w, H, W in ((28, 28, 224, 224), (1, 1, 28, 28), (7, 5, 20, 33), (14, 14, 14, 14)):
        x = rng.standard_normal((3, h, w)).astype(np.float32)
        ref = F.interpolate(torch.from_numpy(x)[None], size=(H, W), mode="bilinear", align_corners=False)[0].numpy()
        np.testing.assert_allclose(O.resize_bilinear(x, (H, W)), ref, atol=2e-6)
    # down-sampling: the antialias window makes it a box-ish average, NOT torch's 2-tap interpolation
    np.testing.assert_allclose(O._resize_weights(8, 4)[:, 1], [0, .125, .375, .375, .125, 0, 0, 0], atol=1e-12)
    np.testing.assert_allclose(O.resize_bilinear(np.full((1, 8, 8), 3.0, np.float32), (4, 4)), 3.0, atol=1e-6)


def test_segmentation_and_vgg_restatements_agree_numpy_vs_torch():
    from oracle import models as OM
    from oracle import state as S
    from oracle import torch_ref as TR
    x = S.synthetic_images(1, 64, seed=0)
    for kind in ("fcn", "deeplabv3"):
        sd = S.segmentation_state(1, kind, (1, 1, 1, 1), 5)
        a, o = OM.segmentation_forward(sd, x[0], kind, (1, 1, 1, 1))
        ta, to = TR.segmentation_forward(sd, x, kind, (1, 1, 1, 1))
        np.testing.assert_allclose(o, to[0].numpy(), atol=1e-5)
        np.testing.assert_allclose(a, ta[0].numpy(), atol=1e-5)
    setting = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 1, 1))
    sd = S.mobilenet_v2_state(1, 10, setting, last=64)
    xs = S.synthetic_images(2, 32, seed=0)
    np.testing.assert_allclose(np.stack([OM.mobilenet_v2_forward(sd, im, setting) for im in xs]),
                               TR.mobilenet_v2_forward(sd, xs, setting).numpy(), atol=1e-5)
    conf, _ = S.mobilenet_v3_conf("small")
    sd = S.mobilenet_v3_state(1, conf[:5], 64, 10)
    xs = S.synthetic_images(2, 64, seed=0)
    np.testing.assert_allclose(np.stack([OM.mobilenet_v3_forward(sd, im, conf[:5]) for im in xs]),
                               TR.mobilenet_v3_forward(sd, xs, conf[:5]).numpy(), atol=1e-5)
    conf_d, _ = S.mobilenet_v3_conf("large", dilated=True)
    sd = S.lraspp_state(1, conf_d, (4, 16), 5)
    np.testing.assert_allclose(OM.lraspp_forward(sd, xs[0], conf_d), TR.lraspp_forward(sd, xs[:1], conf_d)[0].numpy(), atol=1e-5)
    st = [(0, 1, 3, 1, 16, 8, 1), (0, 6, 5, 2, 8, 16, 2), (1, 4, 3, 2, 16, 24, 2), (1, 1, 3, 1, 24, 24, 1)]
    sd = S.efficientnet_state(1, st, 64, 10)
    np.testing.assert_allclose(np.stack([OM.efficientnet_forward(sd, im, st) for im in xs]),
                               TR.efficientnet_forward(sd, xs, st).numpy(), atol=1e-5)
    sd = S.regnet_state(1, (48, 120), (1, 2), (8, 24), 0.25, 32, 10)
    np.testing.assert_allclose(np.stack([OM.regnet_forward(sd, im, (48, 120), (1, 2), (8, 24), 0.25) for im in xs]),
                               TR.regnet_forward(sd, xs, (48, 120), (1, 2), (8, 24), 0.25).numpy(), atol=1e-5)
    plan = (8, "M", 16, "M")
    for bn in (False, True):
        sd = S.vgg_state(1, plan, bn, 10)
        xs = S.synthetic_images(2, 28, seed=0)
        np.testing.assert_allclose(np.stack([OM.vgg_forward(sd, im, plan, bn) for im in xs]),
                                   TR.vgg_forward(sd, xs, plan, bn).numpy(), atol=1e-5)


def test_oracle_jax_random_known_answers():
    """oracle.np_ops' restatement of jax.random (not under /root/reference): Random123 Threefry-2x32 known-answer vectors and the
    PRNGKey(0) values printed in JAX's documentation; dropout / drop_path follow eqx.nn.Dropout and drop_path.py:51-61."""
    from oracle import np_ops as O
    assert O.threefry_2x32([0, 0], [0, 0]).tolist() == [0x6B200159, 0x99BA4EFE]
    assert O.threefry_2x32([0xFFFFFFFF, 0xFFFFFFFF], [0xFFFFFFFF, 0xFFFFFFFF]).tolist() == [0x1CB996FC, 0xBB002BE7]
    assert O.threefry_2x32([0x13198A2E, 0x03707344], [0x243F6A88, 0x85A308D3]).tolist() == [0xC4923A9C, 0x483DF7A0]
    assert O.jax_split(np.array([0, 0], np.uint32)).tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    assert float(O.jax_uniform(np.array([0, 0], np.uint32))) == np.float32(0.41845703)
    key = O.jax_split(np.array([0, 42], np.uint32), 3)[1]
    x = np.arange(1, 61, dtype=np.float32).reshape(3, 4, 5)
    y = O.dropout(x, 0.25, key)
    kept = y != 0
    assert 0.5 < kept.mean() < 0.95 and np.allclose(y[kept], x[kept] / 0.75)
    g = O.drop_path(x, 0.5, "global", key)
    assert (g == 0).all() or np.allclose(g, x * 2)
    l = O.drop_path(x, 0.5, "local", key)
    per_row = (l.reshape(3, -1) != 0).all(1) | (l.reshape(3, -1) == 0).all(1)      # whole first-axis slices are kept or dropped
    assert per_row.all()



def test_gradient_oracle_against_finite_differences():
    """oracle/torch_grad.py (the reference of the GPU gradient cases) checked against central differences of its own loss on a
    small ViT (32 px, patch 16, width 32, 2 blocks): a few entries of different parameter kinds."""
    from oracle import torch_grad as TG
    sd = S.vit_state(3, 32, 16, 32, 2, 4, 4, 5)
    x = S.synthetic_images(3, 32, seed=4)
    labels = [0, 3, 1]
    fn = lambda sd_, x_, y_: TG.vit(sd_, x_, y_, 16, 4, 2, dtype=torch.float64)
    loss, grads = fn(sd, x, labels)
    assert np.isfinite(loss)
    for name, idx in (("fc.bias", 2), ("blocks.1.mlp.fc1.weight", 17), ("blocks.0.attn.qkv.weight", 40), ("pos_embed", 9),
                      ("patch_embed.proj.weight", 100), ("blocks.0.norm1.weight", 3)):
        fd = TG.finite_difference(fn, sd, x, labels, name, idx)
        g = float(np.asarray(grads[name]).reshape(-1)[idx])
        assert abs(fd - g) <= 2e-3 * max(1.0, abs(g)) + 2e-3, (name, fd, g)
