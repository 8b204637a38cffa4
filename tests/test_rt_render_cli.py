"""The non-Python caller of the C ABI: _build/rt_render, the stand-in for the reference's `main`
(/root/reference/src/main.rs:22-89: load the scene -> render_scene at :57 -> save_image at :58).  Run as a fresh
process, as a user would, and compared byte for byte with what the Python binding gets from the same library."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, SCENES, pkg

W, H, SPP, DEPTH = 64, 48, 8, 6


def _cli(*args, expect=0):
    exe = pkg("build").build_cli()
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)
    assert r.returncode == expect, (r.returncode, r.stdout, r.stderr)
    return r


def test_cli_without_a_gpu_fails_loudly_and_usage_errors_are_usage_errors(native, tmp_path):
    """No CPU fallback behind the CLI either: without a device the render call returns MI355RT_ERR_NO_DEVICE and the process exits 1
    after the scene has loaded (the host half is pure CPU).  Bad command lines exit 2 before anything is loaded."""
    import ctypes as C
    _, device = native
    assert _cli(expect=2).stderr.startswith("usage:")
    assert "unknown option" in _cli(SCENES["cornell"], "--nope", expect=2).stderr
    assert "cannot be combined" in _cli(SCENES["cornell"], "--chunk", "2", "--devices", "0,0", expect=2).stderr
    assert "comma-separated" in _cli(SCENES["cornell"], "--devices", "0;1", expect=2).stderr
    assert "unknown variant `infinite_sphere`" in _cli(SCENES["teapot"], "-o", str(tmp_path / "t.png"), expect=1).stderr    # serde's hard error
    h = C.c_void_p()
    if device.lib().mi355rt_context_create(0, C.byref(h)) == 0:
        device.lib().mi355rt_context_destroy(h)
        pytest.skip("a GPU is visible here")
    r = _cli(SCENES["cornell"], "--width", str(W), "--height", str(H), "--spp", str(SPP), "-o", str(tmp_path / "x.png"), expect=1)
    assert "Scene loaded. Objects: 8" in r.stdout and "no HIP device visible" in r.stderr and not (tmp_path / "x.png").exists()


@pytest.mark.gpu
def test_cli_writes_what_the_library_renders(native, abi, tmp_path):
    """load -> render -> save through the C caller: PNG, PFM and EXR bytes equal the files written from device.render's output;
    --chunk 3 (progressive preview) and --devices 0,0 (the two-GPU strip plan on one device) give the same picture."""
    host, device = native
    sc = host.LoadedScene(SCENES["cornell"], W, H, SPP, DEPTH)
    gp, gl, st = device.render(sc, sc.camera, sc.settings, abi.Options.make())
    want = {}
    for ext, write, data in (("png", host.write_png, gp), ("pfm", host.write_pfm, gl), ("exr", host.write_exr, gl)):
        path = str(tmp_path / f"want.{ext}")
        write(path, data, W, H)
        want[ext] = open(path, "rb").read()
    size = ["--width", str(W), "--height", str(H), "--spp", str(SPP), "--max-depth", str(DEPTH)]

    def run(tag, *extra, files=("png", "pfm", "exr")):
        out = {e: str(tmp_path / f"{tag}.{e}") for e in ("png", "pfm", "exr")}
        r = _cli(SCENES["cornell"], *size, "-o", out["png"], "--pfm", out["pfm"], "--exr", out["exr"], *extra)
        assert f"Image saved as '{out['png']}'" in r.stdout and f"Image: {W}x{H}, Samples: {SPP}, Max Depth: {DEPTH}" in r.stdout
        for e in files:
            assert open(out[e], "rb").read() == want[e], (tag, e)
        return r

    r = run("oneshot")
    assert f"{st.rays / st.samples:.2f} rays/sample" in r.stdout
    r = run("chunked", "--chunk", "3")
    assert [ln.strip() for ln in r.stdout.splitlines() if "samples per pixel" in ln] == ["3 / 8 samples per pixel", "6 / 8 samples per pixel", "8 / 8 samples per pixel"]
    r = run("two_devices", "--devices", "0,0")
    assert "(max over devices)" in r.stdout
    # the reference-stream mode through the CLI equals the library's
    rp, rl, _ = device.render(sc, sc.camera, sc.settings, abi.Options.make(rng_mode=abi.RNG_REF))
    p = str(tmp_path / "ref.png")
    host.write_png(str(tmp_path / "ref_want.png"), rp, W, H)
    _cli(SCENES["cornell"], *size, "--rng", "ref", "-o", p)
    assert open(p, "rb").read() == open(str(tmp_path / "ref_want.png"), "rb").read()
    # a device that does not exist is an error message and exit code 1, not a crash
    r = _cli(SCENES["cornell"], *size, "-o", str(tmp_path / "bad.png"), "--devices", "0,99", expect=1)
    assert "render failed (-1)" in r.stderr and "out of range" in r.stderr
