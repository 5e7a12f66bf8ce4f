"""Kernel symbol (as rocprofv3 prints it) -> the bench name of its family.  One bench name per kernel SYMBOL wherever a launch
reports one (igemm8 / igemm8s template instances); shared by tools/profile_session.sh (which writes profiles/traffic.json and
the `_rocprof*` tables) and tests/test_profiles.py (which re-derives those tables from the committed warm-stats files)."""
import re


def fam(name):
    n = name
    m = re.search(r"igemm8_kernel<(float|unsigned short), (true|false), (\d)(?:, (\d))?>", n)   # <OutT, DUAL, MODE[, LNF]>: one bench name per symbol
    if m:
        f32 = "_f32out" if m.group(1) == "float" else ""
        lnf = int(m.group(4) or 0)              # the LayerNorm-fold forms (round 6; profiles older than that print three arguments)
        if lnf: return "igemm8_bf16_256x256_" + {1: "lin_lnout_f32res", 2: "dense_lnin", 3: "lin_lnout", 4: "lin_f32out_splitres"}[lnf]
        if m.group(2) == "true": return "igemm8_dual_bf16_256x256" + f32
        return "igemm8_bf16_256x256_" + ("conv", "dense", "lin")[int(m.group(3))] + f32
    m = re.search(r"igemm8s_kernel<(float|unsigned short), (\d), (true|false), (true|false)>", n)   # <OutT, ARR, DUAL, DENSE>
    if m:
        f32 = "_f32out" if m.group(1) == "float" else ""
        arr = "128x256" if m.group(2) == "0" else "256x128"
        if m.group(3) == "true": return "igemm8_dual_bf16_" + arr + f32
        return "igemm8_bf16_" + arr + ("_dense" if m.group(4) == "true" else "_conv") + f32
    for sub, f in (("unsigned short, true>", "igemm2_dual_bf16_256x256"), ("igemm2_kernel<4, 2, 2, 2, 3", "igemm2_bf16_256x128"),
                   ("igemm2_kernel<2, 4, 4, 2, 2", "igemm2_bf16_256x256"), ("igemm2_kernel<8, 1, 1, 2, 3", "igemm2_bf16_256x64"),
                   ("igemm_bf16_kernel<128, 128", "igemm_bf16_128x128"), ("igemm_bf16_kernel<128, 64", "igemm_bf16_128x64"),
                   ("stream1x1_kernel", "stream1x1"), ("chain1x1_kernel<64, 8, false", "chain1x1_bf16_64_256_64"),
                   ("chain1x1_kernel<128, 6, false, true", "chain1x1_bf16_64_256_128_ysub2"), ("chain1x1_kernel<128", "chain1x1_bf16_64_256_128"), ("chain1x1_kernel<64, 6, true", "chain1x1_dual_bf16_64+64_256_64"), ("chain_stream_kernel", "chain_stream_bf16_128_512_128"),
                   ("chain_rc_kernel<0", "chain_rc0_bf16_64x2_256_64"), ("chain_rc_kernel<1", "chain_rc1_bf16_64x3_256_64"),
                   ("chain_res_kernel<128, true", "chain_res_bf16_64_256_128_ysub2"), ("chain_res_kernel<128, false", "chain_res_bf16_64_256_128"),
                   ("conv3x3c64_v2_kernel", "conv3x3c64_halo"), ("stem_pool_kernel<float, true, 11", "stem_pool11_mfma_f32in"), ("stem_pool_kernel", "stem_pool_mfma_f32in"),
                   ("patch_embed_kernel", "patch_embed_mfma_f32in"), ("mha_mfma_kernel", "mha_mfma_dh64_hm"),
                   ("layernorm_vec_kernel", "layernorm_vec"), ("layernorm_slim_kernel", "layernorm_slim"), ("ln_mlp96_kernel", "ln_mlp96_f32stream"), ("swin_attn_mfma", "swin_attn_mfma"), ("skinny_f32_kernel", "skinny_linear_f32_mfma"),
                   ("bneck_tail_kernel", "bneck_tail_bf16_14x14_256_1024"), ("ln_mlp_stream_kernel", "ln_mlp_stream_c384_f32stream"), ("ln_mlp_stream192_kernel", "ln_mlp_stream_c192_f32stream"),
                   ("swin_win96_kernel", "swin_win96"), ("swin_block_attn_kernel<384", "swin_block_attn_c384"), ("swin_block_attn_kernel<192", "swin_block_attn_c192"), ("swin_block_attn_kernel<96", "swin_block_attn_c96"),
                   ("patch_merge_ln_kernel", "patch_merge_ln_f32in"), ("swin_stem_ln_kernel", "swin_stem_ln_k96"),
                   ("fc_stream_kernel", "fc_stream_bf16"), ("maxpool_nhwc_bf16x8_kernel", "maxpool_nhwc_bf16x8")):
        if sub in n: return f
    return None
