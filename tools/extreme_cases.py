"""Ad-hoc parity check on unusual shapes (4K-wide windows, 5000 spp, 1x1, tall-thin): HIP path vs oracle.  Run on a GPU box."""
import sys, os, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import pkg, load_for_both
import oracle
import torch; torch.zeros(1, device="cuda")
host, device, abi = pkg("host"), pkg("device"), pkg("abi")
oracle.build()
cases = [("cornell", 3840, 2160, 2, 5, dict(row_begin=1000, row_end=1006)),
         ("cornell", 32, 32, 5000, 4, dict()),
         ("teapot", 4096, 8, 3, 6, dict()),
         ("semesterbild", 7, 2000, 2, 8, dict(row_begin=990, row_end=1010)),
         ("cornell", 1, 1, 1, 1, dict()),
         ("cornell", 2, 3, 1, 0, dict())]
for name, W, H, spp, depth, okw in cases:
    sc = load_for_both(name, oracle, host, width=W, height=H, spp=spp, max_depth=depth)
    opt = abi.Options.make(**okw)
    gp, gl, st = device.render(sc, sc.camera, sc.settings, opt)
    op, ol, cnt = oracle.render(sc, sc.camera, sc.settings, opt)
    exact = np.array_equal(gl.view(np.uint32), ol.view(np.uint32)) and np.array_equal(gp, op)
    close = np.abs(gl - ol).max()
    print(name, W, H, spp, depth, okw, "exact" if exact else f"max|d| {close:.2e}", "rays", st.rays, cnt.rays, flush=True)
