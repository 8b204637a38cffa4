"""CPU-only checks of bench.py's bookkeeping: the counters file is only trusted for the kernel sources it was taken on, and the
committed file matches the committed sources (so the driver's bench line says "pmc": "committed (hash-matched)")."""
import importlib
import json
import os
import sys

from conftest import ROOT, pkg

sys.path.insert(0, ROOT)


def test_committed_pmc_counters_belong_to_the_committed_kernel_sources():
    build = pkg("build")
    doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_counters.json")))
    if doc["kernel_hash"] != build.kernel_hash():
        import pytest
        pytest.skip("profiles/pmc_counters.json was taken on other kernel sources: bench.py will print \"pmc\": \"stale\" until tools/pmc_collect.py is re-run on the GPU box")
    bench = importlib.import_module("bench")
    for wl in bench.WORKLOADS:
        if wl == "cornell-box-400x300x16-d4":          # BASELINE configs[0], the CPU plumbing case: a 0.3 ms frame, selectable in bench.py but not a roofline workload
            assert bench.load_pmc(wl, build.kernel_hash())[1] == "absent"
            continue
        rec, state = bench.load_pmc(wl, build.kernel_hash())
        assert state == bench.PMC_OK == "committed (hash-matched)" and rec["valu_wave_insts_per_step"] > 0 and rec["hbm_bytes_per_step"] > 0 and 0 < rec["valu_lane_utilisation"] <= 1
        # every fraction bench.py can print from these counters is physical: issue rate below the peak for any plausible kernel time
        assert rec["kernel"].startswith("k_render_ctr")


def test_stale_or_missing_counters_are_reported_not_used(tmp_path, monkeypatch):
    bench = importlib.import_module("bench")
    p = tmp_path / "pmc.json"
    monkeypatch.setattr(bench, "PMC_FILE", str(p))
    assert bench.load_pmc("cornell-box-800x600x256-d30", "abc") == (None, "absent")
    p.write_text(json.dumps({"kernel_hash": "other", "workloads": {"cornell-box-800x600x256-d30": {"valu_wave_insts_per_step": 1}}}))
    assert bench.load_pmc("cornell-box-800x600x256-d30", "abc") == (None, "stale")
    p.write_text(json.dumps({"kernel_hash": "abc", "workloads": {}}))
    assert bench.load_pmc("cornell-box-800x600x256-d30", "abc") == (None, "absent")
    good = {"kernel": "k_render_ctr_simple", "valu_wave_insts_per_step": 1.0, "hbm_bytes_per_step": 2.0, "valu_lane_utilisation": 0.5}
    p.write_text(json.dumps({"kernel_hash": "abc", "workloads": {"cornell-box-800x600x256-d30": good}}))
    assert bench.load_pmc("cornell-box-800x600x256-d30", "abc", "k_render_ctr_simple") == (good, bench.PMC_OK)
    assert bench.load_pmc("cornell-box-800x600x256-d30", "abc", "k_render_ctr_nomesh") == (None, "other-kernel")   # counters of one variant, time of another
    part = {k: v for k, v in good.items() if k != "hbm_bytes_per_step"}                                              # a counter pass failed
    p.write_text(json.dumps({"kernel_hash": "abc", "workloads": {"cornell-box-800x600x256-d30": part}}))
    assert bench.load_pmc("cornell-box-800x600x256-d30", "abc", "k_render_ctr_simple") == (None, "partial")
    p.write_text("{not json")
    assert bench.load_pmc("cornell-box-800x600x256-d30", "abc") == (None, "unreadable")


def test_cpu_thread_count_respects_an_override(monkeypatch):
    bench = importlib.import_module("bench")
    monkeypatch.setenv("MI355RT_CPU_THREADS", "3")
    assert bench.usable_cores()[0] == 3
    monkeypatch.delenv("MI355RT_CPU_THREADS")
    n, why = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and "affinity" in why


def test_frames_in_flight_fall_back_to_the_queue_classes_that_exist():
    bench = importlib.import_module("bench")
    ok = {"queue_classes_found": 4, "distinct": 4}
    assert bench.fit_frames_to_queues(4, 4, ok) == (4, 4, ok)
    f, s, info = bench.fit_frames_to_queues(4, 4, {"queue_classes_found": 3, "distinct": 3})
    assert (f, s) == (3, 3) and info["reduced_from"] == [4, 4]
    f, s, info = bench.fit_frames_to_queues(4, 2, {"queue_classes_found": 1, "distinct": 1})
    assert (f, s) == (1, 1)
    unprobed = {"queue_classes_found": None}
    assert bench.fit_frames_to_queues(4, 4, unprobed) == (4, 4, unprobed)                      # nothing known: as chosen
    assert bench.fit_frames_to_queues(4, 4, {"queue_classes_found": 2}, auto=False)[:2] == (4, 4)   # typed by the user: as typed
    assert bench.fit_frames_to_queues(1, 1, None) == (1, 1, None)


def test_kernel_hash_covers_the_code_not_the_comments(tmp_path, monkeypatch):
    """profiles/pmc_counters.json is tied to the kernel sources by build.kernel_hash(); the hash is over their CODE (comments and whitespace stripped), so
    that correcting a comment does not orphan the committed counters -- while any change of a token, in a kernel, a header or the host half, does."""
    build = pkg("build")
    assert build._code_only('int a = 1; // c1\n/* c2 */ const char* s = "x // kept"; char q = \'"\'; // "z\n  int   b;') == 'int a = 1; const char* s = "x // kept"; char q = \'"\'; int b;'
    src = tmp_path / "k.hip"; hdr = tmp_path / "h.h"
    src.write_text("__global__ void k(int* p) { *p = 1; }  // one\n"); hdr.write_text("#define X 1\n")
    monkeypatch.setattr(build, "DEVICE_SRCS", [str(src)]); monkeypatch.setattr(build, "DEVICE_HEADERS", [str(hdr)])
    h0 = build.kernel_hash()
    src.write_text("__global__ void k(int* p)   {\n  *p = 1;   /* another comment */ }\n")
    assert build.kernel_hash() == h0
    src.write_text("__global__ void k(int* p) { *p = 2; }\n")
    assert build.kernel_hash() != h0
    hdr.write_text("#define X 2\n"); h1 = build.kernel_hash()
    monkeypatch.setattr(build, "HIPCC_FLAGS", build.HIPCC_FLAGS + ["-O2"])
    assert build.kernel_hash() != h1
