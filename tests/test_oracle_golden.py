"""Pins the oracle to the reference: docs/semesterbild.png is the reference's own committed render of
data/scenes/semesterbild.json at HEAD (800x600, 256 spp, depth 30).  The render is
deterministic (per-row StdRng::seed_from_u64(y), renderer.rs:91) and rows are independent, so every
8th row is rendered and compared against the same rows of the PNG.

SURVEY.md section 4 allowed <= 0.5 % of pixels with |d| > 20: text-mesh triangles whose visibility depends on the tie order of
Rust's sort_unstable_by in the BVH build (App. B-1), which std::stable_sort did not reproduce (0.15 % of the pixels, whole
letter faces).  Since the builders restate Rust's sort itself (rust_sort_unstable.hpp) NO pixel is further than 20/255
from the reference's render, 79.4 % are identical and 90.2 % within +-1 (60.5 % / 80.4 % before; 77.7 % / 89.9 % before the mesh's
rotation quaternion was computed in f32 like glam does).  What is left is a sequential-stream effect: every row is identical up to
the first pixel whose paths meet the glass ball / the text mesh inside it, where one sample of the row takes another branch by an
ulp and the rest of the row draws shifted random numbers -- noise-level differences (max 17/255).  Rows that see only sky, walls
and floor match exactly.
"""
import os

import numpy as np
from PIL import Image

from conftest import ROOT, SCENES


def test_ref_mode_reproduces_the_reference_render(oracle_mod, abi):
    from oracle import scene_loader
    sc = scene_loader.load_scene(SCENES["semesterbild"])
    assert (sc.settings.width, sc.settings.height, sc.settings.samples_per_pixel, sc.settings.max_depth) == (800, 600, 256, 30)
    gold = np.array(Image.open(os.path.join(ROOT, "tests/golden/semesterbild_reference_800x600_256spp.png")).convert("RGB")).astype(np.int32)
    opt = abi.Options.make(rng_mode=abi.RNG_REF, strip_rows=1, n_parts=8, part=3)
    packed, _, cnt = oracle_mod.render(sc, sc.camera, sc.settings, opt, want_linear=False)
    rows = abi.rows_selected(600, opt)
    img = np.stack([(packed >> 16) & 255, (packed >> 8) & 255, packed & 255], axis=-1).astype(np.int32)
    g = gold[rows]
    d = np.abs(img - g)
    assert d.mean() <= 0.35, d.mean()                 # measured 0.28 (0.62 with a stable sort in the BVH build)
    assert abs(img.mean() - g.mean()) <= 0.1
    assert (d.max(-1) > 20).sum() == 0                # the tie order of the reference's BVH build is reproduced: no letter face differs
    assert (d.max(-1) == 0).mean() >= 0.77            # measured 79.4 % (survey probe, stable sort: 61 %)
    assert (d.max(-1) <= 1).mean() >= 0.89            # measured 90.2 % (survey probe: 81 %)
    # sky rows (the top ~30 % of the image): one ray per sample, miss colour GRAY -> 0xB4B4B4, exact
    sky = [i for i, y in enumerate(rows) if y < 100]
    assert sky and np.array_equal(img[sky], g[sky])
    assert np.all(packed[sky] == 0xB4B4B4)
    # work counts of SURVEY.md section 8d, cfg 5
    assert abs(cnt.rays / cnt.samples - 2.99) < 0.1
    assert abs(cnt.bvh_nodes / cnt.rays - 13.07) < 0.3       # the survey's probe (stable sort) counted 12.44: another tree


def test_independent_seed_is_statistically_equal(oracle_mod, abi):
    """A different seed gives a different image with the same mean (noise floor check, small size)."""
    from oracle import scene_loader
    sc = scene_loader.load_scene(SCENES["semesterbild"], width=80, height=60, spp=32)
    a = oracle_mod.render(sc, sc.camera, sc.settings, abi.Options.make(rng_mode=abi.RNG_REF))[1]
    b = oracle_mod.render(sc, sc.camera, sc.settings, abi.Options.make(rng_mode=abi.RNG_REF, seed=100000))[1]
    c = oracle_mod.render(sc, sc.camera, sc.settings, abi.Options.make(rng_mode=abi.RNG_CTR))[1]
    assert not np.array_equal(a, b)
    assert abs(a.mean() - b.mean()) / a.mean() < 0.01
    assert abs(a.mean() - c.mean()) / a.mean() < 0.01      # ctr mode: same estimator, other stream
