"""A/B of kernel variants (diagnostic knob `kernel`) on random MESH-FREE scenes with mixed materials (tests/fuzz_scenes.py) at a size where
the kernels run for milliseconds: is the material-sorting wavefront worth its queues when the list has no mesh?
usage: python tools/ab_fuzz_scene.py [seed ...]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch; torch.zeros(1, device="cuda")
from conftest import pkg
from fuzz_scenes import random_scene
abi, host, device = pkg("abi"), pkg("host"), pkg("device")
CASES = {"all kinds, every exact material": dict(exact_only=True, n_prims=16, only_kinds=[2, 2, 3, 0, 1, 3, 3, 0, 0, 2, 1, 1]),
         "spheres + quads, every material incl. rough": dict(exact_only=False, n_prims=12, only_kinds=[0, 0, 2, 0]),
         "Lambert only": dict(exact_only=True, n_prims=14, only_kinds=[2, 2, 2, 3, 3, 2, 0, 1, 3], lambert_only=True)}
for seed in [int(a) for a in sys.argv[1:]] or [31, 32]:
    for label, kw in CASES.items():
        sc = random_scene(abi, host, seed, **kw)
        st = abi.Settings(640, 480, 64, 12)
        out = torch.zeros(640 * 480, dtype=torch.int32, device="cuda")
        res = {}
        for name, k in (("auto", None), ("lockstep general", 0), ("wavefront mesh-free nospec", 11)):
            c = device.Context(0)
            if k is not None: c.set_knob("kernel", k)
            c.set_scene(sc, sc.camera, st)
            t = []
            for r in range(6):
                s = c.render(out.data_ptr(), None, abi.Options.make(), None, want_stats=True)
                if r: t.append(s.render_kernel_ms)
            res[name] = (c.kernel_variant(), statistics.median(t), int(out.to(torch.int64).sum().item()))
            rps = s.rays / max(s.samples, 1)
            c.close()
        print(f"seed {seed} {label} ({rps:.2f} rays per path): " + "  |  ".join(f"{n}: variant {v} {ms:.3f} ms" for n, (v, ms, _) in res.items()) + ("  checksums equal" if len({x[2] for x in res.values()}) == 1 else "  CHECKSUMS DIFFER"), flush=True)
