#!/bin/bash
# Hardware counters of ONE kernel of a command, a few counters per pass (rocprofv3 --pmc, kernel trace only -- never combined with
# the sys / hip / hsa traces).  usage: tools/pmc_kernel.sh OUTDIR KERNEL_SUBSTRING CMD...   -> OUTDIR/pmc_summary.txt (mean per launch)
O=$1; K=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $O/p$i -o t -- "$@" > $O/p$i.log 2>&1
done <<'PASSES'
SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_RES_STALL_CSN
PASSES
find $O -name "*.db" -delete
O=$O K="$K" python - <<'PY'
import csv, glob, os, collections
O, K = os.environ["O"], os.environ["K"]
tot = collections.OrderedDict()
for f in sorted(glob.glob(f"{O}/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for c, (v, n) in acc.items():
        tot[c] = (v / n, n)
with open(f"{O}/pmc_summary.txt", "w") as f:
    f.write(f"# kernel *{K}*: mean per launch\n")
    for c, (v, n) in tot.items():
        f.write(f"{c:44s} {v:16.0f}   (n={n})\n")
print(open(f"{O}/pmc_summary.txt").read())
PY
