"""GPU: the driver's command line `python bench.py --gpus 1 --steps K --warmup W` - the LAST line of stdout is the JSON line of the contract, with the roofline block, the
calibrated stream placement and a timed region that the repeats behind it confirm (no cold first region, no outlier)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_gpu_bench_line_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-other-configs",
                        "--no-scaling-configs"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])                                   # the last line, whatever a library wrote before it
    assert d["metric"].startswith("trajectories/sec") and d["unit"] == "trajectories/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f64" and d["vs_baseline"] is None and "workload" in d["config"]
    assert abs(d["value"] - 1024 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-12 and r["kernel_ms_samples"] >= 200
    assert abs(r["achieved"] - 3.5e6 * 1024 / (r["kernel_ms"] * 1e-3) / 1e12) <= 1e-9 * r["achieved"]
    assert 0.3 < r["frac"] < 0.7 and 0.07 < d["ms_per_step"] < 0.13 and 0.08 < r["kernel_ms"] < 0.12
    ld = d["config"]["launch_diagnostics"]
    sp = ld["stream_placement"]
    assert sp["chosen"] in sp["candidates"] and len(sp["ms_per_step"]) == len(sp["candidates"]) and min(sp["ms_per_step"]) > 0.05
    # the timed region is not a cold outlier: within 15 % of the repeats issued right behind it
    reps = ld["timed_region_repeated_ms_per_step"]
    assert d["ms_per_step"] <= 1.15 * max(reps) and min(reps) <= 1.15 * d["ms_per_step"], (d["ms_per_step"], reps)
