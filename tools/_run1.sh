cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_session.sh r4prof r04 > gpurun_out/r4prof.log 2>&1
tail -12 gpurun_out/r4prof.log | cut -c1-300
cat gpurun_out/r4prof/roofline_vs_rocprof.txt
cp gpurun_out/r4prof/traffic.json profiles/traffic.json
timeout 400 python bench.py > gpurun_out/r4prof/bench_default_line.json 2> gpurun_out/r4prof/bench_default.err; cut -c1-1500 gpurun_out/r4prof/bench_default_line.json
