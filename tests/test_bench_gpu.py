"""GPU: bench.py end to end on small models -- the JSON contract of the default mode (roofline by time share, expert
roofline, whole-forward fractions, percentiles) and the fp8-arithmetic mode (H scales calibrated on the bench batch, the fused
fp8 expert kernel active at this size)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="4"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]              # ONE JSON line on stdout
    return json.loads(lines[0]), r.stderr


def test_bench_default_contract_small_model():
    d, _ = _run(["--layers", "2", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--latency-iters", "50"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3 and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["value"] > 0 and d["higher_is_better"] is True and d["data"] == "synthetic" and "workload" in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "time_share", "families"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and 0 < rf["frac"] < 1 and 0 < rf["time_share"] <= 1
    assert abs(sum(v["time_share"] for v in rf["families"].values()) - 1.0) < 0.05
    assert d["roofline_expert"]["kernel"].startswith("expert_ffn") and d["roofline_expert"]["bound"] == "hbm"
    lat = d["forward"]["latency_ms"]
    assert lat["n"] >= 50 and lat["min"] <= lat["p50"] <= lat["p99"]
    assert "not a BASELINE.json config" in d["config"]["workload"]         # 2 layers: must not be labelled configs[1]
    # the timed region is repeated; value is the median repeat of exactly `steps` forwards
    reps = d["config"]["ms_per_step_repeats"]
    assert len(reps) >= 5 and min(reps) <= d["ms_per_step"] <= max(reps) and d["config"]["profiler_downgraded"] is False


def test_bench_fp8_arithmetic_small_model():
    d, err = _run(["--layers", "2", "--experts", "64", "--weight-dtype", "fp8", "--fp8-activations", "--batch", "64", "--varlen",
                   "50-500", "--streams", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--latency-iters", "50"])
    assert d["dtype"] == "fp8" and "fp8 arithmetic" in d["config"]["workload"] and d["value"] > 0
    lo, hi = d["config"]["h_scale_min_max"]                # calibrated on the bench batch: 1.25 amax(H) / 448 per layer
    assert 0 < lo <= hi < 1.0
    assert d["roofline_expert"]["kernel"] == "expert_ffn_fused_fp8_kernel", d["roofline_expert"]["kernel"]
    assert d["config"]["packed_rows"] is True
    assert d["forward"]["mfma_peak_tflops"] == 5000.0 and "fp8" in d["forward"]["mfma_peak_is"]     # priced against the chip's fp8 peak
    assert "benchmark batch" in d["config"]["h_scale_calibrated_on"]
