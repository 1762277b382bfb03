"""bench.py's other workloads at small size (the headline run is the driver's): every mode prints one JSON line whose parity
field says the GPU rows are the CPU leg's rows, so that a regression of kernel D / the PE chain shows in -m gpu and not only in
a bench run somebody has to remember."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    env = dict(os.environ)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_adna_small():
    """BASELINE config 5 (-n 0.01 -o 2 -l 16500, 30-70 bp damaged reads) through kernel D, 20 k reads on a 64 Mbp text"""
    r = run_bench("--adna", "--reads", "20000", "--genome-len", "64000000", "--steps", "1", "--warmup", "0", "--cpu-seconds", "5", "--no-e2e")
    assert r["unit"] == "reads/s" and r["value"] > 0
    assert r["config"]["bit_exact_vs_cpu_sample"] is True, r["config"]
    assert r["cpu_baseline"]["kind"] in ("reference", "port") and r["cpu_baseline"]["value"] > 0
    assert r["roofline"]["bound"] == "hbm" and 0 < r["roofline"]["frac"] < 1
    assert r["config"]["second_pass_reads"] > 0                                 # kernel D had work


def test_bench_pe_small():
    """BASELINE config 3 (2 x 150 bp pairs): search of both ends + posn_pair + finish_pair, a sample against the reference chain"""
    r = run_bench("--pe", "--pairs", "20000", "--genome-len", "64000000", "--steps", "1", "--warmup", "0")
    assert r["unit"] == "pairs/s" and r["value"] > 0
    assert r["config"]["bit_exact_vs_cpu_sample"] is True, r["config"]


def test_bench_repeats_small():
    """the repeat-family text (LINE / Alu / tandem families): the branchy reads reach kernel D; rows are the CPU leg's"""
    r = run_bench("--repeats", "--reads", "50000", "--genome-len", "256000000", "--steps", "1", "--warmup", "0", "--cpu-seconds", "5", "--no-e2e")
    assert r["config"]["bit_exact_vs_cpu_sample"] is True, r["config"]
