import importlib
import os
import sys

import pytest
import torch  # noqa: F401  -- FIRST: torch ships its own HIP runtime; when libmi355rt.so initialises the system's copy before torch is imported, torch.cuda later reports "No HIP GPUs are available" (two runtimes in one process)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = {
    "cornell": os.path.join(ROOT, "data/scenes/tungsten/cornell-box/scene.json"),
    "veach": os.path.join(ROOT, "data/scenes/tungsten/veach-mis/scene.json"),
    "teapot": os.path.join(ROOT, "data/scenes/tungsten/teapot/scene.json"),
    "semesterbild": os.path.join(ROOT, "data/scenes/semesterbild.json"),
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(name=""):
    return importlib.import_module("raytracer-rust_amd" + (("." + name) if name else ""))


@pytest.fixture(scope="session")
def abi():
    return pkg("abi")


@pytest.fixture(scope="session")
def native():
    """Build (if stale) and return the product's native bindings (host, device)."""
    b = pkg("build")
    b.build_host()
    b.build_device()
    return pkg("host"), pkg("device")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


def load_for_both(name, oracle_mod, host, **kw):
    """Scene through the ORACLE-side loader + the PRODUCT BVH builder (independent of the C++ loader)."""
    from oracle import scene_loader
    sc = scene_loader.load_scene(SCENES[name], skip_unknown_primitives=(name == "teapot"), **kw)
    sc._keep = host.attach_bvh(sc)
    return sc
