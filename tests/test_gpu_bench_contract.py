"""bench.py's output contract on a real GPU: one JSON line on stdout, the fields the round prompt names, and the roofline arithmetic a
reader can recompute from the line itself (44 + 251 + 5 N bytes per agent-env-step)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "24", "--warmup", "4", *extra], capture_output=True, text=True, timeout=600,
                         cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # exactly ONE line on stdout
    return json.loads(lines[0])


def test_default_line_follows_the_contract():
    d = _run("--cpu-seconds", "1.5")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 24 and d["warmup"] == 4 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and d["unit"] == "agent-env-steps/s"
    c = d["config"]
    assert "workload" in c and c["n_agents"] == 16 and c["envs_per_gpu"] == 4096 and "model" not in c
    # value = agents x envs x steps / time
    assert abs(d["value"] - 16 * 4096 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["algorithmic_bytes_per_agent_env_step"] == 44 + 251 + 5 * 16
    assert abs(r["achieved"] - 375 * d["value"] / 1e9) <= 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-12
    # 24 timed steps = ONE launch of the in-kernel step loop (sigmaenv_step_autoreset_n)
    assert c["steps_per_launch"] == 24 and r["steps_per_launch"] == 24.0
    assert r["kernel"] == "sigmaenv_step_wave_kernel" and r["kernel_launches"] >= 1 and 0.0 < r["kernel_ms_per_step"] < 1.0
    assert abs(r["kernel_ms_per_step"] * 24 - r["kernel_avg_ms"]) <= 1e-9 and r["algorithmic_bytes_per_launch"] == 375 * 16 * 4096 * 24
    assert r["achieved_incl_record"] > r["achieved"] and 0.0 < r["frac"] < 1.0
    assert r["traffic"] is None or "traffic_source" in r
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert c["per_step_launch"]["ms_per_step"] > 0 and c["per_step_launch"]["env_shards_per_gpu"] == 2
    # `value` is the contract measurement (W warm-up steps on the device as the setup left it); the sustained-clock figure stands beside it, never in its place
    assert d["metric"].endswith("(open loop, 24 steps per launch)")
    assert d["warmup_effective_steps"] == d["warmup"] == 4
    assert d["value_sustained"] > 0 and d["warmup_effective_steps_sustained"] == 2 * 4 + 24 + c["sustained"]["conditioning_steps"] > d["warmup"]
    assert abs(d["value_sustained"] - 16 * 4096 / (d["ms_per_step_sustained"] * 1e-3)) <= 1e-6 * d["value_sustained"]
    assert c["sustained"]["value"] == d["value_sustained"] and "cold_start" not in c
    assert abs(r["frac_sustained"] - 375 * d["value_sustained"] / 1e9 / 8000.0) <= 1e-12
    assert r["frac_measured"] is None or abs(r["frac_measured"] - r["achieved_measured"] / 8000.0) <= 1e-12
    # the form that materialises every per-step output (one launch per step) next to the T-step launch's figure
    assert abs(r["frac_full_outputs"] - 375 * c["per_step_launch"]["value"] / 1e9 / 8000.0) <= 1e-12 and r["frac_full_outputs"] < r["frac_sustained"]
    # the issue-side picture against the MACHINE: VALU wave64 instructions x 2 cycles / (SIMDs x shader clock x time); between 0 and 1, and below the rate relative
    # to the measured FMA stream (which itself does not reach the 2-cycle limit)
    if "valu_issue_frac_arch" in r:
        assert 0.0 < r["valu_lane_cycle_frac_arch"] <= r["valu_issue_frac_arch"] < 1.0
        assert abs(r["valu_issue_frac_arch"] - r["valu_inst_per_s_per_simd"] * 2.0 / (r["shader_clock_hz_measured"] or 2.4e9)) <= 1e-9
    # every other single-GPU BASELINE configuration rides in the same line (timed after the headline's regions)
    lines = {l["name"]: l for l in c["lines"]}
    assert list(lines) == ["config4_on_ramp_32x8192", "config5_cbf_qp", "distance_mtv", "reference_defaults"]
    for name, l in lines.items():
        assert "error" not in l, l
        assert l["value"] > 0 and abs(l["value"] - l["n_agents"] * l["envs_per_gpu"] / (l["ms_per_step"] * 1e-3)) <= 1e-6 * l["value"] and l["wall_s"] < 8.0
        lr = l["roofline"]
        assert lr["bound"] in ("hbm", "valu") and 0.0 < lr["frac"] < 1.0 and lr["kernel"] == l["dominant_kernel"] and lr["kernel_avg_ms"] > 0
    assert lines["config4_on_ramp_32x8192"]["n_agents"] == 32 and lines["config4_on_ramp_32x8192"]["envs_per_gpu"] == 8192
    assert lines["config4_on_ramp_32x8192"]["roofline"]["algorithmic_bytes_per_agent_env_step"] == 44 + 251 + 5 * 32
    assert "cbf_qp" in lines["config5_cbf_qp"]["dominant_kernel"] and lines["config5_cbf_qp"]["env_shards_per_gpu"] == 2
    q = lines["config5_cbf_qp"]["roofline"]
    if q["bound"] == "valu":  # (the committed PMC pass matches this workload) issue rate against one wave64 instruction per 2 cycles at the measured clock
        assert q["unit"] == "G wave64-inst/s/SIMD" and abs(q["frac"] - q["achieved"] / q["peak"]) <= 1e-12 and 0.0 < q["f64_inst_share"] < 1.0 and q["hbm_frac"] < 0.05
    assert lines["distance_mtv"]["value"] < d["value_sustained"] and lines["reference_defaults"]["value"] < lines["distance_mtv"]["value"] * 1.05


def test_without_conditioning_there_is_no_sustained_figure():
    d = _run("--cpu-seconds", "0", "--no-compare", "--condition-ms", "0", "--no-lines")
    assert "value_sustained" not in d and "sustained" not in d["config"] and d["warmup_effective_steps"] == 4 and "lines" not in d["config"]


def test_config4_and_qp_lines_name_their_workload():
    d = _run("--cpu-seconds", "0", "--no-compare", "--scenario", "on_ramp_1", "--agents", "32", "--envs-per-gpu", "1024")
    assert "on_ramp_1" in d["config"]["workload"] and d["config"]["n_agents"] == 32 and "cpu_baseline" not in d
    assert d["roofline"]["algorithmic_bytes_per_agent_env_step"] == 44 + 251 + 5 * 32
    d = _run("--cpu-seconds", "0", "--no-compare", "--cbf-qp", "--envs-per-gpu", "512")
    assert "cbf" in d["config"] and d["value"] > 0 and d["metric"].endswith("(CBF-QP filter launch before every step)")


def _trace_top_kernel(tmp_path, *bench_args):
    """bench.py under `rocprofv3 --kernel-trace --stats`: (the JSON line, {kernel name: total ns} from the trace's kernel stats)."""
    import csv
    import glob

    out_dir = str(tmp_path / "trace")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out_dir, "-o", "t", "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--cpu-seconds", "0", "--no-compare", *bench_args]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd="/tmp", env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][-1])
    totals = {}
    for f in glob.glob(os.path.join(out_dir, "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            totals[row["Name"]] = totals.get(row["Name"], 0.0) + float(row["TotalDurationNs"])
    assert totals, "no kernel stats produced"
    return d, totals


@pytest.mark.parametrize("args,expect", [
    (["--steps", "32", "--warmup", "32"], "sigmaenv_step_wave_kernel"),                                   # the headline: the step kernel
    (["--steps", "16", "--warmup", "4", "--cbf-qp", "--envs-per-gpu", "1024"], "sigmaenv_cbf_qp_kernel"),  # config 5: the QP kernel dominates
])
def test_roofline_names_the_kernel_the_trace_ranks_first(tmp_path, args, expect):
    """`roofline.kernel` is the kernel with the largest share of GPU time -- checked against rocprofv3's own kernel statistics of the same command --
    and its HIP-event average agrees with the trace's average."""
    d, totals = _trace_top_kernel(tmp_path, *args)
    ours = {k: v for k, v in totals.items() if "sigmaenv" in k}
    top = max(ours, key=ours.get)
    r = d["roofline"]
    assert expect in r["kernel"] and expect in top, (r["kernel"], top)
    assert r["kernel"].split("::")[-1] in top
    assert 0.0 < r["frac"] < 1.0 and r["kernel_avg_ms"] > 0
