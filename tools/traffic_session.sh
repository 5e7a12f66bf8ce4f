#!/bin/bash
# HBM traffic per launch from the L2 memory-side counters (MI355X_MICROARCH.md, HBM section): separate --pmc
# passes for FETCH_SIZE and WRITE_SIZE, FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), KB -> bytes.
# usage: tools/traffic_session.sh TAG MODEL [BATCH]   -> gpurun_out/TAG/traffic_MODEL.json
TAG=${1:-tr}; MODEL=${2:-resnet50}; BATCH=${3:-256}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -o t -- python bench.py --model $MODEL --batch $BATCH --steps 3 --warmup 3 --no-cpu > $O/$C.log 2>&1
done
find $O -name "*.db" -delete
O=$O MODEL=$MODEL python - <<'PY'
import csv, glob, collections, json, os, re
O, MODEL = os.environ["O"], os.environ["MODEL"]
FAM = [("unsigned short, true>", "igemm2_dual_bf16_256x256"), ("igemm2_kernel<4, 2, 2, 2, 3", "igemm2_bf16_256x128"), ("igemm2_kernel<2, 4, 4, 2, 2", "igemm2_bf16_256x256"),
       ("igemm2_kernel<8, 1, 1, 2, 3", "igemm2_bf16_256x64"), ("igemm3_kernel", "igemm3_bf16_256x256"),
       ("igemm_bf16_kernel<128, 128", "igemm_bf16_128x128"), ("igemm_bf16_kernel<128, 64", "igemm_bf16_128x64"),
       ("stream1x1_kernel", "stream1x1"), ("chain1x1_kernel<64, 8, false", "chain1x1_bf16_64_256_64"), ("chain1x1_kernel<128", "chain1x1_bf16_64_256_128"),
       ("chain1x1_kernel<64, 6, true", "chain1x1_dual_bf16_64+64_256_64"),
       ("conv3x3c64_v2_kernel", "conv3x3c64_halo"), ("stem_pool_kernel", "stem_pool_mfma_f32in"), ("conv3x3c64_kernel", "conv3x3c64_stream"), ("stem_patch_kernel", "stem_patch_mfma_f32in"),
       ("mha_mfma_kernel", "mha_mfma_dh64_hm"), ("layernorm_vec_kernel", "layernorm_vec"), ("swin_attn_mfma", "swin_attn_mfma"),
       ("maxpool_nhwc_bf16x8", "maxpool_nhwc_bf16x8")]
def fam(name):
    for sub, f in FAM:
        if sub in name: return f
    return None
acc = {"FETCH_SIZE": collections.defaultdict(list), "WRITE_SIZE": collections.defaultdict(list)}
for C in acc:
    for f in glob.glob(f"{O}/{C}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == C:
                k = fam(r["Kernel_Name"])
                if k: acc[C][k].append(float(r["Counter_Value"]))
out = {}
lines = []
for k in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
    fs, ws = acc["FETCH_SIZE"].get(k, []), acc["WRITE_SIZE"].get(k, [])
    rd = 2.0 * 1024.0 * sum(fs) / max(1, len(fs))           # KB, doubled (guide: gfx950 FETCH_SIZE = half of a coalesced stream)
    wr = 1024.0 * sum(ws) / max(1, len(ws))                 # KB (uncalibrated per the guide)
    out[k] = round(rd + wr)
    lines.append(f"{k:28s} launches {len(fs):5d}  read {rd/1e6:9.2f} MB  write {wr/1e6:9.2f} MB  total {(rd+wr)/1e6:9.2f} MB per launch")
json.dump({MODEL: out}, open(f"{O}/traffic_{MODEL}.json", "w"), indent=1)
open(f"{O}/traffic_{MODEL}.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
os.system(f"find {O} -size +3M -delete")
PY
