"""Whole-image parity at the BASELINE sizes: HIP path vs oracle, counter-mode RNG, every pixel.
Run on a GPU box (the oracle uses all host cores; cornell 800x600x256 takes ~6 s on 16 cores)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import pkg, load_for_both
import oracle
import torch; torch.zeros(1, device="cuda")
abi, host, device = pkg("abi"), pkg("host"), pkg("device")
oracle.build()
CASES = [("cornell", 800, 600, 256, 30), ("teapot", 800, 600, 256, 64), ("veach", 1280, 720, 256, 16), ("semesterbild", 800, 600, 256, 30)]
for name, W, H, spp, depth in CASES:
    sc = load_for_both(name, oracle, host, width=W, height=H, spp=spp, max_depth=depth)
    opt = abi.Options.make()
    gp, gl, st = device.render(sc, sc.camera, sc.settings, opt)
    t = time.time(); op, ol, cnt = oracle.render(sc, sc.camera, sc.settings, opt); dt = time.time() - t
    same_f32 = (gl.view(np.uint32) == ol.view(np.uint32)).all(-1)
    l2 = np.sqrt(((gl.astype(np.float64) - ol) ** 2).sum(-1))
    print(f"{name} {W}x{H}x{spp} d{depth}: pixels bit-identical (f32 linear) {same_f32.mean() * 100:.4f} %  8-bit identical {(gp == op).mean() * 100:.4f} %  "
          f"L2 <= 1e-3 on {(l2 <= 1e-3).mean() * 100:.4f} %  max per-pixel L2 {l2.max():.3e}  rays gpu {st.rays} oracle {cnt.rays}  (oracle {dt:.1f} s, gpu kernel {st.render_kernel_ms:.1f} ms)", flush=True)
