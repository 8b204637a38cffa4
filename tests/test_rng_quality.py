"""The quality gate of the counter-mode generator as a (small) test: tools/rng_battery.py -- the ten SmallCrush tests restated, run on the ADDRESSED
stream the kernels draw -- must pass the shipped generator (pcg4d with the per-path base) and must FAIL deliberately weak ones; a battery that
passes everything proves nothing.  The full-size runs (2^24 and 2^26 words per stream order, all generators) are committed under profiles/r05/."""
import json
import os
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_shipped_generator_passes_and_weak_ones_fail():
    import rng_battery as B
    good = B.run("pcg4d", 20, ["path", "pixel", "row", "seeds"])
    ps = [(o, k, p) for o, d in good.items() for k, p in d["tests"].items()]
    assert len(ps) >= 70
    assert not [x for x in ps if B.verdict(x[2], x[1]) == "FAIL"], [x for x in ps if B.verdict(x[2], x[1]) == "FAIL"]
    for weak in ("pcg4d_half", "lcg"):                                      # one mixing round only; a linear congruential step per word
        res = B.run(weak, 20, ["path", "pixel"])
        fails = [k for d in res.values() for k, p in d["tests"].items() if B.verdict(p, k) == "FAIL"]
        assert len(fails) >= 5, (weak, fails)


def test_the_battery_itself_is_calibrated_on_a_trusted_generator():
    """Every test's p-value on numpy's PCG64 is unremarkable (this caught a real bug in the tool: a tail-merging step that indexed a list while popping it)."""
    import rng_battery as B
    w = np.random.default_rng(7).integers(0, 1 << 32, size=1 << 20, dtype=np.uint64).astype(np.uint32)
    for t in B.TESTS + [B.t_unit_ball]:
        for name, p in t(w):
            assert B.verdict(p, name) != "FAIL", (name, p)


def test_committed_battery_results_cover_the_shipped_generator():
    for f in ("rng_battery_2p24.json", "rng_battery_2p26.json"):
        doc = json.load(open(os.path.join(ROOT, "profiles", "r05", f)))
        g = doc["generators"]["pcg4d"]
        assert g["n_tests"] >= 70 and g["fail"] == [] and g["suspect"] == []
    weak = json.load(open(os.path.join(ROOT, "profiles", "r05", "rng_battery_2p24.json")))["generators"]
    assert len(weak["pcg4d_half"]["fail"]) >= 10 and len(weak["lcg"]["fail"]) >= 50
