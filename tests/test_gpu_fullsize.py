"""Parity at the BASELINE.json sizes through size-independent properties (the oracle cannot render these sizes in seconds):
  * rows are independent units: any subset of rows rendered by the oracle must equal the same rows of the full GPU image;
  * tiling invariance at full size: the image assembled from 8 interleaved strip sets equals the one-shot image;
  * the GPU's counter-mode render of semesterbild at 800x600x256 against the reference's own committed render
    (different random numbers, so statistical -- SURVEY.md section 8c, definition 3):
    image-mean relative difference < 0.5 %, and RMSE(gpu, reference) no larger than what the ORACLE gets against the
    reference when it, too, uses an independent random stream (the golden differs from any render of ours by MC noise
    plus the BVH tie-order holes of SURVEY App. B-1, so the pure noise floor is not reachable: survey 2.62 vs 1.94).
"""
import os

import numpy as np
import pytest
from PIL import Image

from conftest import ROOT, SCENES

pytestmark = pytest.mark.gpu


def _rgb(packed):
    return np.stack([(packed >> 16) & 255, (packed >> 8) & 255, packed & 255], axis=-1).astype(np.float64)


def test_cornell_800x600x256_rows_equal_the_oracle_and_tiling_is_invariant(native, oracle_mod, abi):
    host, device = native
    sc = host.LoadedScene(SCENES["cornell"], 800, 600, 256, 30)
    full, full_lin, st = device.render(sc, sc.camera, sc.settings, abi.Options.make())
    assert st.samples == 800 * 600 * 256
    # 6 rows spread over the image (sky rows, box rows, light rows), oracle in the same counter mode
    opt = abi.Options.make(strip_rows=1, n_parts=100, part=37)
    rows = abi.rows_selected(600, opt)
    op, ol, cnt = oracle_mod.render(sc, sc.camera, sc.settings, opt)
    assert np.array_equal(full_lin[rows].view(np.uint32), ol.view(np.uint32)) and np.array_equal(full[rows], op)
    # 8 interleaved parts (the 8-GPU decomposition), assembled
    out = np.zeros_like(full)
    rays = 0
    for part in range(8):
        o = abi.Options.make(strip_rows=3, n_parts=8, part=part)
        p, _, s = device.render(sc, sc.camera, sc.settings, o, want_linear=False)
        out[abi.rows_selected(600, o)] = p
        rays += s.rays
    assert np.array_equal(out, full) and rays == st.rays


def test_semesterbild_800x600x256_statistics_against_the_reference_render(native, oracle_mod, abi):
    host, device = native
    sc = host.LoadedScene(SCENES["semesterbild"])                      # 800x600, 256 spp, depth 30 as shipped
    gp, gl, st = device.render(sc, sc.camera, sc.settings, abi.Options.make())
    gold = np.array(Image.open(os.path.join(ROOT, "tests/golden/semesterbild_reference_800x600_256spp.png")).convert("RGB")).astype(np.float64)
    g = _rgb(gp)
    assert abs(g.mean() - gold.mean()) / gold.mean() < 0.005
    # noise floor from two independent oracle renders (reference RNG stream) of every 10th row
    opt_a = abi.Options.make(rng_mode=abi.RNG_REF, strip_rows=1, n_parts=10, part=4)
    opt_b = abi.Options.make(rng_mode=abi.RNG_REF, strip_rows=1, n_parts=10, part=4, seed=100000)
    rows = abi.rows_selected(600, opt_a)
    a = _rgb(oracle_mod.render(sc, sc.camera, sc.settings, opt_a, want_linear=False)[0])
    b = _rgb(oracle_mod.render(sc, sc.camera, sc.settings, opt_b, want_linear=False)[0])
    floor = np.sqrt(((a - b) ** 2).mean())                              # pure MC noise between two independent renders
    rmse_gpu = np.sqrt(((g[rows] - gold[rows]) ** 2).mean())
    rmse_same = np.sqrt(((a - gold[rows]) ** 2).mean())                 # oracle on the reference's own stream
    rmse_indep = np.sqrt(((b - gold[rows]) ** 2).mean())                # oracle on an independent stream
    assert rmse_same < rmse_indep                                       # following the reference stream is measurably closer
    assert rmse_gpu <= 1.1 * rmse_indep and rmse_gpu <= 1.5 * floor, (rmse_gpu, rmse_indep, floor)
    sky = [y for y in range(600) if y < 100]
    assert np.array_equal(g[sky], gold[sky])                            # miss colour rows are exact whatever the stream


def test_gpu_reference_stream_replay_reproduces_the_reference_render(native, abi):
    """MI355RT_RNG_REF on the GPU: every row consumes StdRng::seed_from_u64(y) exactly like renderer.rs:91-101, so the
    image must reproduce docs/semesterbild.png the way the CPU oracle does (SURVEY.md section 4 thresholds): identical
    pixels until a row's first ulp-level divergence, identical sky rows, BVH tie-order holes in < 0.5 % of pixels."""
    host, device = native
    sc = host.LoadedScene(SCENES["semesterbild"])
    gp, _, st = device.render(sc, sc.camera, sc.settings, abi.Options.make(rng_mode=abi.RNG_REF), want_linear=False)
    gold = np.array(Image.open(os.path.join(ROOT, "tests/golden/semesterbild_reference_800x600_256spp.png")).convert("RGB")).astype(np.float64)
    d = np.abs(_rgb(gp) - gold)
    assert d.mean() <= 1.0 and abs(_rgb(gp).mean() - gold.mean()) <= 0.1
    assert (d.max(-1) > 20).mean() <= 0.005
    assert (d.max(-1) == 0).mean() >= 0.5 and (d.max(-1) <= 1).mean() >= 0.7
    assert np.array_equal(gp[:100], np.full((100, 800), 0xB4B4B4, np.uint32))
