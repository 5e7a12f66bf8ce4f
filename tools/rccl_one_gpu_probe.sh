#!/bin/bash
# Round-3 review item 6(ii): does ncclCommInitRank(nranks = 2) come up through mv_comm_init when both ranks sit on ONE device?
# (A 1-GPU box cannot host the real thing; either it works -- then the native gather runs with 2 ranks -- or the exact RCCL error is
# recorded so that the first 8-GPU run is not the first time anybody sees it.)  usage: tools/rccl_one_gpu_probe.sh OUTDIR
O=${1:-gpurun_out/rccl_probe}; mkdir -p $O
export EQV_DIST_DEVICE=0 EQV_DIST_COLLECTIVE=rccl MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0 NCCL_DEBUG=WARN
unset EQV_DIST_BACKEND
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29611 \
  tests/_dist_worker.py $O/r.json 8 > $O/probe.log 2>&1
echo "exit code $?" >> $O/probe.log
grep -i "nccl\|rccl\|mv_comm\|error\|warn\|exit code\|Duplicate" $O/probe.log | sort | uniq -c | sort -rn | head -30
[ -f $O/r.json ] && cat $O/r.json
