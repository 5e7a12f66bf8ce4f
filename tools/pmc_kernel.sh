#!/bin/bash
# PMC passes for ONE kernel family: usage [CMD='python tools/one_gemm.py ...'] tools/pmc_kernel.sh TAG MODEL KERNEL_SUBSTR [BATCH]
TAG=${1:-pk}; MODEL=${2:-vit_base}; KSUB=${3:-mha}; BATCH=${4:-256}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CYCLES"
P3="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE"
P4="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  if [ -n "$CMD" ]; then RUN="$CMD"; else RUN="python bench.py --model $MODEL --batch $BATCH --steps 1 --warmup 3 --no-cpu --no-graph"; fi
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p$i -o p$i -- $RUN > $O/p$i.log 2>&1
done
find $O -name "*.db" -delete
KSUB=$KSUB O=$O python - <<'PY'
import csv, glob, collections, os
O, KSUB = os.environ["O"], os.environ["KSUB"]
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(O + "/p?/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if KSUB in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
with open(O + "/summary_" + KSUB + ".txt", "w") as out:
    for c in sorted(tot):
        out.write(f"{c:36s} per-dispatch {tot[c] / max(1, n[c]):16.1f}   n={n[c]}\n")
print(open(O + "/summary_" + KSUB + ".txt").read())
os.system(f"find {O} -size +2M -delete")
PY
