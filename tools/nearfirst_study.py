#!/usr/bin/env python3
"""CPU-only study: does a near-child-first BVH walk (explicit stack, equal-t ties broken by pre-order
triangle position) return the hit of the reference's fixed left-then-right recursion (bvh.rs:142-156)?

Renders the mesh scenes with the oracle in counter mode at the BASELINE size and runs EVERY mesh walk in both orders
(oracle/rt_oracle.cpp, bvh_intersect_ordered vs bvh_intersect_reference_id).  Prints one JSON object per scene; the
committed copy is profiles/r02_nearfirst_study.json.  A non-zero `differ_*` means the ordered walk cannot be the default
traversal of a bit-exact drop-in."""
import ctypes as C, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import oracle
oracle.build()
abi = importlib.import_module("raytracer-rust_amd.abi")
host = importlib.import_module("raytracer-rust_amd.host")
SCENES = {"semesterbild-800x600": ("data/scenes/semesterbild.json", 800, 600, 30, False),
          "teapot-800x600": ("data/scenes/tungsten/teapot/scene.json", 800, 600, 64, True)}
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = {}
for name, (path, W, H, depth, skip) in SCENES.items():
    sc = host.LoadedScene(os.path.join(ROOT, path), W, H, spp, depth, skip_unknown_primitives=skip)
    st = np.zeros(8, np.uint64)
    L = oracle.lib()
    L.oracle_walk_study.argtypes = [C.c_void_p]
    L.oracle_walk_study(st.ctypes.data)
    try:
        _, _, cnt = oracle.render(sc, sc.camera, sc.settings, abi.Options.make(), want_linear=False)
    finally:
        L.oracle_walk_study(None)
    keys = ["walks", "differ_tri", "differ_hitmiss", "differ_t_only", "nodes_ref", "nodes_ordered", "tris_ref", "tris_ordered"]
    rec = {k: int(v) for k, v in zip(keys, st)}
    rec.update(spp=spp, samples=int(cnt.samples), rays=int(cnt.rays), seconds=round(cnt.seconds, 1),
               nodes_per_walk_ref=round(rec["nodes_ref"] / max(rec["walks"], 1), 2), nodes_per_walk_ordered=round(rec["nodes_ordered"] / max(rec["walks"], 1), 2),
               tris_per_walk_ref=round(rec["tris_ref"] / max(rec["walks"], 1), 2), tris_per_walk_ordered=round(rec["tris_ordered"] / max(rec["walks"], 1), 2))
    out[name] = rec
    print(json.dumps({name: rec}), flush=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_nearfirst_study.json"), "w"), indent=1)
