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


if __name__ == "__main__":
    main()
