#!/usr/bin/env python3
"""Generates tests/golden/oracle_*.npz: small renders of every scene by the CPU oracle, in both RNG
modes, stored as linear f32 + packed u32.  The GPU box has no /root/reference and the fixtures pin the
oracle itself against regressions; the GPU parity tests compare the HIP path to the SAME files.

The reference (Rust) cannot run here, so these vectors come from the oracle, which is itself pinned to
the reference's committed render (tests/test_oracle_golden.py).  Re-run after any intended oracle change:
    python tools/make_golden_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                                   # noqa: E402
from oracle import abi, scene_loader           # noqa: E402

CASES = {   # name: (scene path, W, H, spp, depth, skip_unknown)
    "cornell": ("data/scenes/tungsten/cornell-box/scene.json", 64, 48, 8, 6, False),
    "veach": ("data/scenes/tungsten/veach-mis/scene.json", 64, 36, 8, 16, False),
    "teapot": ("data/scenes/tungsten/teapot/scene.json", 64, 48, 4, 16, True),
    "semesterbild": ("data/scenes/semesterbild.json", 64, 48, 8, 30, False),
}


def main():
    oracle.build()
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, (path, W, H, spp, depth, skip) in CASES.items():
        sc = scene_loader.load_scene(os.path.join(ROOT, path), W, H, spp, depth, skip_unknown_primitives=skip)
        data = {"meta": np.array([W, H, spp, depth], np.uint32)}
        for mode, tag in ((abi.RNG_CTR, "ctr"), (abi.RNG_REF, "ref")):
            packed, linear, cnt = oracle.render(sc, sc.camera, sc.settings, abi.Options.make(rng_mode=mode), threads=0)
            data[f"{tag}_linear"] = linear
            data[f"{tag}_packed"] = packed
            data[f"{tag}_rays"] = np.array([cnt.rays], np.uint64)
        np.savez_compressed(os.path.join(out_dir, f"oracle_{name}.npz"), **data)
        print(name, {k: v.shape for k, v in data.items()})


def main_cfg1():
    """BASELINE.json configs[0] at its own size -- cornell-box 400x300, 16 spp, max_bounces 4: too large to commit as pixels
    (1.4 MB of f32), so the fixture holds SHA-256 digests of the oracle's whole image in both RNG modes plus its work counters
    (2.77 rays per sample).  The product loader (C++) supplies the scene: this is the CPU "plumbing" leg."""
    import hashlib
    import importlib
    import json
    host = importlib.import_module("raytracer-rust_amd.host")
    importlib.import_module("raytracer-rust_amd.build").build_host()
    sc = host.LoadedScene(os.path.join(ROOT, "data/scenes/tungsten/cornell-box/scene.json"), 400, 300, 16, 4)
    doc = {"scene": "data/scenes/tungsten/cornell-box/scene.json", "width": 400, "height": 300, "spp": 16, "max_depth": 4}
    for mode, tag in ((abi.RNG_CTR, "ctr"), (abi.RNG_REF, "ref")):
        packed, linear, cnt = oracle.render(sc, sc.camera, sc.settings, abi.Options.make(rng_mode=mode), threads=0)
        doc[tag] = {"packed_sha256": hashlib.sha256(packed.tobytes()).hexdigest(), "linear_sha256": hashlib.sha256(linear.tobytes()).hexdigest(),
                    "packed_sum": int(packed.astype(np.int64).sum()), "samples": int(cnt.samples), "rays": int(cnt.rays),
                    "depth_exhausted": int(cnt.depth_exhausted), "rng_words": int(cnt.rng_words)}
    json.dump(doc, open(os.path.join(ROOT, "tests", "golden", "oracle_cfg1_cornell_400x300x16_d4.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
    main_cfg1()
