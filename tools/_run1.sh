cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tail -4
for m in "vit_tiny 8" "vit_base 4"; do timeout 600 python tools/time_train_step.py $m 3 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -1; done
