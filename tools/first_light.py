"""Ad-hoc GPU probe: parity numbers and timing for the headline config. Not a test."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, importlib
import oracle
from conftest import load_for_both, pkg
host, device, abi = pkg("host"), pkg("device"), pkg("abi")
for name, W, H, spp, depth in [("cornell", 80, 60, 8, 4), ("teapot", 64, 48, 4, 16), ("veach", 96, 54, 8, 16), ("semesterbild", 80, 60, 8, 30)]:
    sc = load_for_both(name, oracle, host, width=W, height=H, spp=spp, max_depth=depth)
    for mode in (abi.RNG_CTR, abi.RNG_REF):
        opt = abi.Options.make(rng_mode=mode)
        gp, gl, st = device.render(sc, sc.camera, sc.settings, opt)
        op, ol, cnt = oracle.render(sc, sc.camera, sc.settings, opt)
        d = np.abs(gl - ol)
        print(name, "mode", mode, "bit-identical", np.array_equal(gl.view(np.uint32), ol.view(np.uint32)), "max|d|", d.max(), "px differ", int((d.max(-1) > 0).sum()), "/", W * H,
              "packed equal", float((gp == op).mean()), "rays", st.rays, cnt.rays, "mean", gl.mean(), ol.mean(), flush=True)
for name, W, H, spp, depth in [("cornell", 800, 600, 256, 30), ("semesterbild", 800, 600, 256, 30), ("veach", 1280, 720, 64, 16), ("teapot", 800, 600, 64, 30)]:
    sc = load_for_both(name, oracle, host, width=W, height=H, spp=spp, max_depth=depth)
    for rep in range(2):
        t = time.time(); gp, gl, st = device.render(sc, sc.camera, sc.settings, abi.Options.make()); wall = time.time() - t
        print(name, W, H, spp, "render_ms", st.render_kernel_ms, "resolve_ms", st.resolve_kernel_ms, "wall_s", wall, "Msamples/s", st.samples / st.render_kernel_ms / 1e3,
              "rays/sample", st.rays / st.samples, "grid", st.grid_blocks, "vgprs", st.kernel_vgprs, "bands", st.bands, "mean", float(gl.mean()), flush=True)
