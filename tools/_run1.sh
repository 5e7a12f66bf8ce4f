cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python tools/_fc_gemm_probe.py > $O/probe.log 2>&1
python - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/r3q/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fc_stream_kernel" in r["Kernel_Name"] or "fc_reduce" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, "gemm" if "fc_stream" in r["Kernel_Name"] else "reduce", r["Grid_Size"] if "Grid_Size" in r else ""))
rows.sort()
shapes=[l.strip() for l in open("gpurun_out/r3q/probe.log") if l.startswith("SHAPE")]
# 13 launches of each kernel per shape (3 warm + 10)
per=13
g=[r for r in rows if r[2]=="gemm"]; rd=[r for r in rows if r[2]=="reduce"]
for i,sh in enumerate(shapes):
    gg=[x[1] for x in g[i*per+3:(i+1)*per]]; rr=[x[1] for x in rd[i*per+3:(i+1)*per]]
    import re
    M,K,N=[int(v) for v in re.findall(r"M=(\d+) K=(\d+) N=(\d+)", sh)[0]]
    gm=sum(gg)/len(gg); rm=sum(rr)/len(rr)
    print(f"{sh:50s} gemm {gm:7.1f} us ({2.0*M*N*K/gm/1e6:6.0f} TF/s)  reduce {rm:6.1f} us")
PY
find $O -size +3M -delete
