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
    s = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc.json")))
    assert {"S", "W", "D"} <= set(s) and s["units"] == 10_000_000
    assert bench.pmc_traffic("headline", 10_000_000) > 1e11                 # kernel S on the headline workload: ~172 GB per launch
    for kind, units in (("adna", 6_250_000), ("pe", 1_000_000), ("repeats", 10_000_000)):      # kernel D at the sizes of the driver's line
        d = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_%s.json" % kind)))
        assert "D" in d and d["units"] == units
        assert bench.pmc_traffic(kind, units) > 1e11, kind
        assert bench.pmc_traffic(kind, units + 1) is None                   # a pass taken at another launch size says nothing
