"""Long parity soak: many random scenes (every primitive / material kind), HIP path vs oracle, both RNG modes.
usage: python tools/fuzz_soak.py [first_seed] [n_seeds]   (run on a GPU box; prints one line per seed and a summary)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import pkg
import oracle
from fuzz_scenes import random_scene
import torch; torch.zeros(1, device="cuda")
abi, host, device = pkg("abi"), pkg("host"), pkg("device")
oracle.build()
first, n = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0; t0 = time.time()
for seed in range(first, first + n):
    shapes = [dict(), dict(n_prims=5, mesh_tris=400), dict(n_prims=30, mesh_tris=10), dict(n_prims=8, only_kinds=[4, 3, 4, 2], mesh_tris=150),
              dict(n_prims=5 + seed % 19, only_kinds=[[2, 2, 3, 0, 1, 3, 3, 0, 0, 2, 1, 1], [3, 2, 0, 1], [0, 0, 3], [2, 3, 3, 1]][seed // 6 % 4]),   # mesh-free: general lockstep kernel
              dict(n_prims=4 + seed % 13, only_kinds=[[2, 2, 2, 3, 3, 2, 0, 1, 3], [3], [2, 3]][seed // 6 % 3], lambert_only=True),          # Lambert-only lockstep kernel
              dict(n_prims=5 + seed % 11, mesh_tris=20 + seed % 200, identity_meshes=True, no_metal=True)]                                      # untransformed meshes, no metal: k_render_ctr_wf_nometal_ident
    kw = shapes[seed % len(shapes)]
    rough = seed % 5 == 4                                  # every fifth scene has GGX / Beckmann rough conductors too: only the reference-stream
    sc = random_scene(abi, host, seed, exact_only=not rough, **kw)   # mode is bit-exact there (ln / atan / sin / cos rounded once from double on both sides)
    st = abi.Settings(40 + seed % 37, 30 + seed % 23, 3 + seed % 6, 2 + seed % 11)
    line = []
    for mode in ((1,) if rough else (0, 1)):
        opt = abi.Options.make(rng_mode=mode, seed=seed * 7919 if mode == 0 else 0)
        gp, gl, gs = device.render(sc, sc.camera, st, opt)
        op, ol, cnt = oracle.render(sc, sc.camera, st, opt)
        ok = gs.rays == cnt.rays and np.array_equal(gl.view(np.uint32), ol.view(np.uint32)) and np.array_equal(gp, op)
        bad += 0 if ok else 1
        line.append("ok" if ok else f"MISMATCH({int((gl != ol).any(-1).sum())} px, rays {gs.rays} vs {cnt.rays})")
    print(f"seed {seed} {st.width}x{st.height}x{st.samples_per_pixel} d{st.max_depth} {kw}: " + (f"rough conductors, ref {line[0]}" if rough else f"ctr {line[0]}  ref {line[1]}"), flush=True)
print(f"{n} scenes x 2 modes: {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
