#!/bin/bash
# Calibration of the step kernel's VALU figures: the pure-FMA stream of tools/pk_probe/probe.hip (4 wavefronts per SIMD, 16 independent fp32
# accumulators per lane) under the SAME counters the step kernel's issue fraction is built from, plus its kernel trace -- cycles per instruction of a stream that does
# nothing but issue fp32 VALU work -- and the step kernel's instruction rate restated against it.  Usage (GPU box): tools/valu_calibration.sh <outdir>
R=$GRAFT_REPO_ROOT; out=$R/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
P=$R/tools/pk_probe/probe
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAVES --output-format csv -d $out/probe_pmc -o p -- $P > $out/probe_pmc.txt 2> $out/probe_pmc.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/probe_trace -o t -- $P > $out/probe_trace.txt 2> $out/probe_trace.err
$P > $out/probe_plain.txt
cd $R
python - "$out" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "probe_pmc", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "probe_trace", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"].split("(")[0]].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])))
lines, cal = [], {}
for k, cs in sorted(acc.items()):
    # every probe kernel is launched twice: a 16-iteration warm-up and the 20000-iteration measurement -- take the long one
    pick = lambda v: max(v)
    insts, act, busy, wavec = pick(cs["SQ_INSTS_VALU"]), pick(cs["SQ_ACTIVE_INST_VALU"]), pick(cs["SQ_BUSY_CYCLES"]), pick(cs["SQ_WAVE_CYCLES"])
    ns = max(dur.get(k, [0.0]))
    per_simd_inst = insts / 1024.0
    lines.append(f"{k}\n  SQ_INSTS_VALU {insts:.4e}  SQ_ACTIVE_INST_VALU {act:.4e} ({act / insts:.3f} per instruction)  SQ_BUSY_CYCLES {busy:.4e}  SQ_WAVE_CYCLES {wavec:.4e}"
                 f"\n  duration {ns * 1e-3:.1f} us (kernel trace) -> {insts / 1024.0 / (ns * 1e-9) / 1e9 if ns else 0:.4f} G wavefront-instructions / s per SIMD"
                 f" = {ns * 1e-9 * 2.4e9 / per_simd_inst if ns else 0:.3f} cycles of a nominal 2.4 GHz clock per instruction")
    cal[k] = {"insts": insts, "active_inst_valu": act, "busy_cycles": busy, "duration_ns": ns, "inst_per_s_per_simd": insts / 1024.0 / (ns * 1e-9) if ns else None}
open(os.path.join(out, "valu_calibration.txt"), "w").write("\n".join(lines) + "\n")
json.dump(cal, open(os.path.join(out, "valu_calibration_raw.json"), "w"), indent=1)
print("\n".join(lines))
PY
