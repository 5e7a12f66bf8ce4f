#!/bin/bash
# PMC counter collection (own runs, --kernel-trace only): usage tools/pmc_session.sh TAG MODEL
TAG=${1:-pmc}; MODEL=${2:-vit_base}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/p1 -o p1 -- python bench.py --model $MODEL --steps 2 --warmup 3 --no-cpu --no-graph > $O/p1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $O/p2 -o p2 -- python bench.py --model $MODEL --steps 2 --warmup 3 --no-cpu --no-graph > $O/p2.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p3 -o p3 -- python bench.py --model $MODEL --steps 2 --warmup 3 --no-cpu --no-graph > $O/p3.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/p4 -o p4 -- python bench.py --model $MODEL --steps 2 --warmup 3 --no-cpu --no-graph > $O/p4.log 2>&1
find $O -name "*.db" -delete
python - <<PY
import csv, glob, collections, os
for pdir in sorted(glob.glob("$O/p?")):
    f = glob.glob(pdir + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(pdir, "no counter file", os.listdir(pdir)); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"][:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
    with open(pdir + "_summary.txt", "w") as out:
        for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:12]:
            out.write(k + "\n")
            for c, v in d.items():
                out.write(f"    {c:32s} total {v:16.0f}  per-dispatch {v / max(1, cnt[(k, c)]):14.1f}  n={cnt[(k, c)]}\n")
    print(open(pdir + "_summary.txt").read()[:3000])
    os.system(f"find {pdir} -size +3M -delete")
PY
