#!/bin/bash
# Kernel trace of one bench.py command + the launch timeline of the last steps.  Usage: tools/trace_timeline.sh <tag> <bench args...>   (SIGMAENV_LIB is honoured)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python bench.py --cpu-seconds 0 "$@" > "$out/bench.json" 2> "$out/err.txt"
f=$(find "$out" -name "t_kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-160
python - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/t_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "cbf" in r["Kernel_Name"] or "step_wave" in r["Kernel_Name"] or "mlp" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-260:-230]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    nm = r["Kernel_Name"]; nm = nm[nm.find("sigmaenv"):][:34]
    print("%-36s q%-3s start %8.1f us  dur %7.1f us" % (nm, r.get("Queue_Id", ""), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
