#!/usr/bin/env python3
"""Quality gate of the counter-mode generator at the IMAGE level, run on the GPU box:

for each of the four scene families at the BASELINE image size, the HIP path in counter mode (whatever generator the library was built with)
against the CPU oracle on the REFERENCE's random stream (ChaCha12 row streams: different numbers, so the comparison is statistical):
  * image-mean relative difference of the linear radiance < 0.2 %,
  * RMSE(gpu ctr, oracle ref) <= 1.25 x RMSE(oracle ref, oracle ref') where ref' is the oracle with another seed -- the pure Monte Carlo noise
    between two independent renders: a generator with visible structure would sit above that floor.
The oracle renders every K-th row (rows are the reference's independent unit) to bound the CPU time; the GPU image is compared on the same rows.

usage: python tools/rng_image_gate.py [--rows-every 6] [--out profiles/r05/rng_image_gate.json]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                     # noqa: E402
import torch                                           # noqa: E402,F401  (first: see tests/conftest.py)
from conftest import pkg                               # noqa: E402
import oracle                                          # noqa: E402  (the checker)

CASES = {   # name: (scene, W, H, spp, depth, skip_unknown)
    "cornell": ("data/scenes/tungsten/cornell-box/scene.json", 800, 600, 256, 30, False),
    "teapot": ("data/scenes/tungsten/teapot/scene.json", 800, 600, 256, 64, True),
    "veach-mis": ("data/scenes/tungsten/veach-mis/scene.json", 1280, 720, 256, 16, False),
    "semesterbild": ("data/scenes/semesterbild.json", 800, 600, 256, 30, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-every", type=int, default=6)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    abi, host, device, build = pkg("abi"), pkg("host"), pkg("device"), pkg("build")
    oracle.build()
    threads = min(os.cpu_count() or 1, 16)
    doc = {"kernel_hash": build.kernel_hash(), "rows_every": args.rows_every, "cases": {}, "limits": {"mean_rel": 0.002, "rmse_ratio": 1.25}}
    ok = True
    for name, (path, W, H, spp, depth, skip) in CASES.items():
        sc = host.LoadedScene(os.path.join(ROOT, path), W, H, spp, depth, skip_unknown_primitives=skip)
        t0 = time.time()
        _, gl, st = device.render(sc, sc.camera, sc.settings, abi.Options.make(rng_mode=abi.RNG_CTR))
        t_gpu = time.time() - t0
        K = args.rows_every
        oa = abi.Options.make(rng_mode=abi.RNG_REF, strip_rows=1, n_parts=K, part=K // 2)
        ob = abi.Options.make(rng_mode=abi.RNG_REF, strip_rows=1, n_parts=K, part=K // 2, seed=100000)
        rows = abi.rows_selected(H, oa)
        t0 = time.time()
        _, a, _ = oracle.render(sc, sc.camera, sc.settings, oa, threads=threads)
        _, b, _ = oracle.render(sc, sc.camera, sc.settings, ob, threads=threads)
        t_cpu = time.time() - t0
        g = gl[rows].astype(np.float64); a = a.astype(np.float64); b = b.astype(np.float64)
        mean_rel = abs(g.mean() - a.mean()) / a.mean()
        mean_rel_floor = abs(b.mean() - a.mean()) / a.mean()                       # what two independent reference-stream renders differ by
        floor = float(np.sqrt(((a - b) ** 2).mean()))
        rmse = float(np.sqrt(((g - a) ** 2).mean()))
        passed = bool(mean_rel < 0.002 and rmse <= 1.25 * floor)
        ok = ok and passed
        doc["cases"][name] = {"size": f"{W}x{H}x{spp} d{depth}", "rows": len(rows), "gpu_mean": float(g.mean()), "ref_mean": float(a.mean()), "mean_rel_diff": float(mean_rel),
                              "mean_rel_diff_of_two_ref_renders": float(mean_rel_floor), "rmse_gpu_vs_ref": rmse, "rmse_ref_vs_ref2": floor,
                              "rmse_ratio": rmse / floor if floor > 0 else None, "pass": passed, "gpu_s": round(t_gpu, 2), "oracle_s": round(t_cpu, 1), "rays_per_sample": st.rays / max(st.samples, 1)}
        print(f"{name:13s} mean rel diff {mean_rel:.5f} (two ref renders: {mean_rel_floor:.5f})  rmse gpu/ref {rmse:.5f} floor {floor:.5f} ratio {rmse / floor if floor else float('nan'):.3f}  "
              f"{'PASS' if passed else 'FAIL'}  [{t_gpu:.1f} s gpu, {t_cpu:.1f} s oracle]", flush=True)
    doc["pass"] = ok
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(doc, open(args.out, "w"), indent=1)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
