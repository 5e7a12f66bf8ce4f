cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_session.sh r4prof r04 > gpurun_out/r4prof.log 2>&1
tail -9 gpurun_out/r4prof.log | cut -c1-250
cp gpurun_out/r4prof/traffic.json profiles/traffic.json
timeout 400 python bench.py > gpurun_out/r4prof/bench_default_line.json 2> gpurun_out/r4prof/bench_default.err; cut -c1-400 gpurun_out/r4prof/bench_default_line.json
