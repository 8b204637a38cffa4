"""Calibration of set_scene's probe threshold (PROBE_RAYS_PER_PATH, rt_api.cpp): veach-mis with max_bounces 1 .. 16 changes how much a path
scatters; at which rays-per-path does the mesh-free wavefront kernel (11) overtake the lockstep kernel without metal / dielectric (9)?"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch; torch.zeros(1, device="cuda")
from conftest import pkg, SCENES
abi, host, device = pkg("abi"), pkg("host"), pkg("device")
W, H, SPP = 1280, 720, 64
out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
for depth in (1, 2, 3, 4, 6, 16):
    sc = host.LoadedScene(SCENES["veach"], W, H, SPP, depth)
    res = {}
    for k in (9, 11, None):
        c = device.Context(0)
        if k is not None: c.set_knob("kernel", k)
        c.set_scene(sc, sc.camera, sc.settings)
        t = []
        for r in range(6):
            s = c.render(out.data_ptr(), None, abi.Options.make(), None, want_stats=True)
            if r: t.append(s.render_kernel_ms)
        res[k] = (c.kernel_variant(), statistics.median(t), s.rays / s.samples)
        c.close()
    print(f"max_bounces {depth:2d}: {res[9][2]:.2f} rays per path | lockstep (9) {res[9][1]:.3f} ms | mesh-free wavefront (11) {res[11][1]:.3f} ms ({100 * (res[11][1] / res[9][1] - 1):+.1f} %) | automatic choice: {res[None][0]}", flush=True)
