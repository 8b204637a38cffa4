"""The CPU-side producers (scene loader, JSON / OBJ / WO3 / HDR readers, BVH build, PNG / PFM writers) and the oracle's entry
points under AddressSanitizer + UndefinedBehaviorSanitizer: the shipped scenes and ~70 malformed inputs (truncated WO3, OBJ
face indices out of range / negative / zero, deeply nested JSON, Radiance HDR with bad runs or absurd dimensions ...) must
come back as OK or as an error code -- never a sanitizer report, a crash or a surprise.  Recipe: tools/sanitize_host.py."""
import importlib.util
import os
import shutil

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_library_and_oracle_are_clean_under_asan_and_ubsan():
    spec = importlib.util.spec_from_file_location("sanitize_host", os.path.join(ROOT, "tools", "sanitize_host.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rc, out, bad = mod.run()
    assert rc == 0 and not bad, out[-4000:]
    assert "all cases behaved" in out
