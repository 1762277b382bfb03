"""bench.py end to end on a small synthetic genome: one JSON line with the contract's fields, the roofline and cpu_baseline
objects, and the GPU rows of the CPU sample bit-exact (the CPU leg runs the compiled reference when oracle/_ref travelled,
the restatement otherwise)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--pipeline"]])
def test_bench_line(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--genome-len", "30000000", "--reads", "200000", "--steps", "2",
           "--warmup", "1", "--cpu-seconds", "3"] + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "reads/s" and d["value"] > 0
    assert d["config"]["bit_exact_vs_cpu_sample"] is True
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] and d["roofline"]["kernel_ms"] > 0
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["value"] > 0
    assert abs(d["value"] - 200000 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 0.01


def test_bench_line_with_the_other_workloads():
    """the driver's line at small size: after the headline steps, configs 5 (--adna) and 3 (--pe) and the repeat-family genome, each
    with its own value, roofline, CPU baseline and bit-exact sample; the e2e leg with its own CPU baseline and the sample compared
    down to CIGAR / NM / MD"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--genome-len", "256000000", "--reads", "200000", "--steps", "2", "--warmup", "1",
           "--cpu-seconds", "3", "--extras", "on", "--extra-adna-reads", "20000", "--pairs", "20000", "--extra-repeat-reads", "50000",
           "--extra-cpu-seconds", "3", "--extra-steps", "1", "--e2e-reads", "50000"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["bit_exact_vs_cpu_sample"] is True and d["config"]["instrumented_run_same_rows"] is True
    e = d["e2e"]
    assert e["bit_exact_vs_reference_sample"] is True and e["sample_reads"] == 50000 and "CIGAR" in e["sample_fields"]
    assert e["cpu_baseline"]["kind"] == "reference" and e["cpu_baseline"]["value"] > 0
    w = d["workloads"]
    assert set(w) == {"adna", "pe", "repeats"}
    for k, unit in (("adna", "reads/s"), ("pe", "pairs/s"), ("repeats", "reads/s")):
        x = w[k]
        assert x["unit"] == unit and x["value"] > 0 and x["steps"] == 1, k
        assert x["config"]["bit_exact_vs_cpu_sample"] is True and x["config"]["instrumented_run_same_rows"] is True, (k, x["config"])
        assert 0 < x["roofline"]["frac"] < 1 and x["roofline"]["kernel_ms"] > 0, k
        assert x["cpu_baseline"]["kind"] in ("reference", "port") and x["cpu_baseline"]["value"] > 0, k
    assert w["adna"]["config"]["second_pass_reads"] > 0
