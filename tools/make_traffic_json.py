#!/usr/bin/env python
"""Builds profiles/traffic_latest.json (HBM bytes per step-kernel launch) from the FETCH_SIZE / WRITE_SIZE passes of
tools/pmc_passes.sh.  Correction per MI355X_MICROARCH.md (HBM / rocprofv3): the counters are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of a wide coalesced read, so the read side is doubled (upper bound for narrower accesses); WRITE_SIZE is taken
as is (uncalibrated).  Usage: tools/make_traffic_json.py <pmc_outdir> <n_agents> <envs_per_launch> <distance> <out.json> [scenario]
Environment: STEPS_PER_LAUNCH (default 1): env steps per launch of the profiled run (the in-kernel step loop); launches with fewer steps (warm-up
remainders) must not be in the pass -- profile with --warmup equal to a multiple of the chunk."""
import csv, glob, json, os, sys

out_dir, n_agents, envs, distance, dst = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
scenario = sys.argv[6] if len(sys.argv) > 6 else "cpm_entire"
vals = {"FETCH_SIZE": [], "WRITE_SIZE": []}
for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "sigmaenv_step_wave_kernel" in row.get("Kernel_Name", "") and row["Counter_Name"] in vals:
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
fetch = sum(vals["FETCH_SIZE"]) / max(1, len(vals["FETCH_SIZE"]))
write = sum(vals["WRITE_SIZE"]) / max(1, len(vals["WRITE_SIZE"]))
STEPS = float(os.environ.get("STEPS_PER_LAUNCH", "1"))
rec = {
    "kernel": "sigmaenv_step_wave_kernel", "steps_per_launch": STEPS, "scenario": scenario, "n_agents": n_agents, "envs_per_launch": envs, "distance": distance,
    "fetch_size_kib_raw": fetch, "write_size_kib_raw": write, "launches_averaged": len(vals["FETCH_SIZE"]),
    "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
    "correction": "read side = 2 x FETCH_SIZE (gfx950 wide-read under-count, upper bound for narrower accesses); WRITE_SIZE as reported",
    "algorithmic_bytes_per_launch": (44 + 251 + 5 * n_agents) * n_agents * envs * STEPS,
}
json.dump(rec, open(dst, "w"), indent=1)
print(json.dumps(rec))
