"""The committed counter passes that bench.py quotes (roofline.traffic) name the kernels bench.py asks for: a kernel renamed without
its summary regenerated would silently turn the field into null."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_counter_summaries_hold_the_kernels_bench_quotes():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for k in ("NABWA_KMER_T", "NABWA_TEXT_MODE", "NABWA_TRIP_BUDGET"):
        assert k not in os.environ
    s = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc.json")))
    assert {"S", "W", "D"} <= set(s)
    d = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_adna.json")))
    assert "D" in d
    assert bench.pmc_traffic() and bench.pmc_traffic() > 1e11          # kernel S on the headline workload: ~172 GB per launch
    assert bench.pmc_traffic(True) and bench.pmc_traffic(True) > 1e11  # kernel D on the ancient-DNA workload
