"""Every pixel of the reference's committed render docs/semesterbild.png against the oracle's replay of the reference stream
(all 600 rows; tests/test_oracle_golden.py keeps every 8th).  usage: python tools/golden_full_check.py   (CPU, a few minutes)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from PIL import Image
import oracle
from conftest import pkg, SCENES
abi = pkg("abi")
from oracle import scene_loader
oracle.build()
sc = scene_loader.load_scene(SCENES["semesterbild"])
gold = np.array(Image.open(os.path.join(ROOT, "tests/golden/semesterbild_reference_800x600_256spp.png")).convert("RGB")).astype(np.int32)
packed, _, cnt = oracle.render(sc, sc.camera, sc.settings, abi.Options.make(rng_mode=abi.RNG_REF), want_linear=False)
img = np.stack([(packed >> 16) & 255, (packed >> 8) & 255, packed & 255], axis=-1).astype(np.int32)
d = np.abs(img - gold).max(-1)
print(f"{img.shape[1]}x{img.shape[0]}: {(d == 0).sum()} of {d.size} pixels identical, max |d| = {d.max()}, rows identical: {(d.max(1) == 0).sum()} of {d.shape[0]}; "
      f"rays/sample {cnt.rays / cnt.samples:.4f}, BVH nodes/ray {cnt.bvh_nodes / cnt.rays:.3f}")
sys.exit(0 if d.max() == 0 else 1)
