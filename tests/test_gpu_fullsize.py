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


def _l2(a, b):
    return np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum(-1))


def test_teapot_800x600x256_d64_rows_equal_the_oracle(native, oracle_mod, abi):
    """BASELINE config 3 at full size (derived fixture: infinite_sphere dropped, WO3 read with the reference's stride).
    Plastic + checker use only + - * / sqrt, so the rows must be bit-identical."""
    host, device = native
    sc = host.LoadedScene(SCENES["teapot"], 800, 600, 256, 64, skip_unknown_primitives=True)
    full, full_lin, st = device.render(sc, sc.camera, sc.settings, abi.Options.make())
    assert st.samples == 800 * 600 * 256
    opt = abi.Options.make(strip_rows=1, n_parts=75, part=41)               # 8 rows: sky, spout, body, checker floor
    rows = abi.rows_selected(600, opt)
    op, ol, cnt = oracle_mod.render(sc, sc.camera, sc.settings, opt)
    assert np.array_equal(full_lin[rows].view(np.uint32), ol.view(np.uint32)) and np.array_equal(full[rows], op)
    # the same rows rendered alone must trace exactly the oracle's rays
    _, _, st_rows = device.render(sc, sc.camera, sc.settings, opt, want_linear=False)
    assert st_rows.rays == cnt.rays


def test_veach_mis_1280x720x1024_d16_rows_match_the_oracle(native, oracle_mod, abi):
    """BASELINE config 4 at full size.  RoughConductor evaluates logf / atanf / sincosf, where the device libm and glibc differ
    by ulps: per-pixel linear-RGB L2 <= 1e-3 on >= 99.5 % of the pixels, >= 99 % of the 8-bit pixels identical."""
    host, device = native
    sc = host.LoadedScene(SCENES["veach"], 1280, 720, 1024, 16)
    full, full_lin, st = device.render(sc, sc.camera, sc.settings, abi.Options.make())
    assert st.samples == 1280 * 720 * 1024 and st.bands == 1
    opt = abi.Options.make(strip_rows=1, n_parts=120, part=77)              # 6 rows through lights, plates and floor
    rows = abi.rows_selected(720, opt)
    op, ol, cnt = oracle_mod.render(sc, sc.camera, sc.settings, opt)
    d = _l2(full_lin[rows], ol)
    assert (d <= 1e-3).mean() >= 0.995, (d.max(), (d <= 1e-3).mean())
    assert (full[rows] == op).mean() >= 0.99
    _, _, st_rows = device.render(sc, sc.camera, sc.settings, opt, want_linear=False)
    assert abs(int(st_rows.rays) - int(cnt.rays)) <= 1e-6 * cnt.rays        # an ulp may flip a branch for isolated samples


def test_semesterbild_1920x1080x4096_d30_bands_rows_and_tiling(native, oracle_mod, abi):
    """BASELINE config 5 on ONE GPU: 8.49 G samples > the 2^31-sample band limit, so the radiance workspace is cycled through
    4 bands.  Rows against the oracle (GGX floor: same tolerance as veach-mis), and the 8-GPU strip decomposition assembled on
    one GPU must reproduce the banded one-shot image bit-for-bit."""
    host, device = native
    sc = host.LoadedScene(SCENES["semesterbild"], 1920, 1080, 4096, 30)
    full, full_lin, st = device.render(sc, sc.camera, sc.settings, abi.Options.make())
    assert st.samples == 1920 * 1080 * 4096 and st.bands == 4
    opt = abi.Options.make(strip_rows=1, n_parts=360, part=181)             # rows 181 (wall), 541 (text mesh + sphere), 901 (floor)
    rows = abi.rows_selected(1080, opt)
    op, ol, cnt = oracle_mod.render(sc, sc.camera, sc.settings, opt)
    d = _l2(full_lin[rows], ol)
    assert (d <= 1e-3).mean() >= 0.995, (d.max(), (d <= 1e-3).mean())
    assert (full[rows] == op).mean() >= 0.99
    out = np.zeros_like(full)
    rays = 0
    for part in range(8):
        o = abi.Options.make(strip_rows=3, n_parts=8, part=part)
        p, _, s = device.render(sc, sc.camera, sc.settings, o, want_linear=False)
        assert s.bands == 1
        out[abi.rows_selected(1080, o)] = p
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
    """MI355RT_RNG_REF on the GPU: every row consumes StdRng::seed_from_u64(y) exactly like renderer.rs:91-101, and the HIP path
    -- through the C ABI, on the product loader's scene and BVH -- reproduces the reference's own committed render
    docs/semesterbild.png EXACTLY: all 480 000 pixels.  (What that took: Rust's sort_unstable_by restated in the BVH builder,
    glam's quaternion in f32, the camera's tan correctly rounded, and in this mode ln / atan / sin / cos rounded once from double;
    tests/test_oracle_golden.py has the story.  With the device's native float functions 594 of the 600 rows are exact.)"""
    host, device = native
    sc = host.LoadedScene(SCENES["semesterbild"])
    gp, _, st = device.render(sc, sc.camera, sc.settings, abi.Options.make(rng_mode=abi.RNG_REF), want_linear=False)
    gold = np.array(Image.open(os.path.join(ROOT, "tests/golden/semesterbild_reference_800x600_256spp.png")).convert("RGB")).astype(np.float64)
    d = np.abs(_rgb(gp) - gold)
    assert d.max() == 0, f"{(d.max(-1) != 0).sum()} of 480000 pixels differ from the reference's render (max {d.max()})"
    assert np.array_equal(gp[:100], np.full((100, 800), 0xB4B4B4, np.uint32))
