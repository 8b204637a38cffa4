"""BASELINE.json configs[0] at its own size: cornell-box scene.json, 400x300, 16 spp, max_bounces 4
(the reference's render loop, /root/reference/src/renderer.rs:67-123, with the overrides applied where parser.rs:260-285 parses them).

CPU ("plumbing", no GPU): product loader (C++) -> oracle in the reference-stream mode over ALL rows -> PNG writer -> read back;
the whole image is pinned by the committed digests of tests/golden/oracle_cfg1_cornell_400x300x16_d4.json
(tools/make_golden_fixtures.py), the work counters by SURVEY.md 8d (2.77 rays per sample, 0.393 depth-exhausted paths per sample),
and the result must not depend on how many threads share the rows.

GPU: the whole image through host.LoadedScene + mi355rt_render, counter mode and reference-stream mode, bit-identical in linear
f32 and packed pixels, with equal ray counts, against the oracle run in the same process and against the committed digests.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import ROOT, SCENES

W, H, SPP, DEPTH = 400, 300, 16, 4


def _fixture():
    doc = json.load(open(os.path.join(ROOT, "tests/golden/oracle_cfg1_cornell_400x300x16_d4.json")))
    assert (doc["width"], doc["height"], doc["spp"], doc["max_depth"]) == (W, H, SPP, DEPTH)
    return doc


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_cfg1_cpu_plumbing_whole_image(native, oracle_mod, abi, tmp_path):
    host, _ = native
    fx = _fixture()
    sc = host.LoadedScene(SCENES["cornell"], W, H, SPP, DEPTH)
    assert (sc.settings.width, sc.settings.height, sc.settings.samples_per_pixel, sc.settings.max_depth) == (W, H, SPP, DEPTH)
    opt = abi.Options.make(rng_mode=abi.RNG_REF)
    packed, linear, cnt = oracle_mod.render(sc, sc.camera, sc.settings, opt, threads=0)
    assert packed.shape == (H, W) and cnt.samples == W * H * SPP == fx["ref"]["samples"]
    assert _sha(packed) == fx["ref"]["packed_sha256"] and _sha(linear) == fx["ref"]["linear_sha256"]
    assert cnt.rays == fx["ref"]["rays"] and cnt.depth_exhausted == fx["ref"]["depth_exhausted"]
    # SURVEY.md 8d, cfg 1 row: 2.77 rays / sample, 14.4 RNG words / sample, 0.393 depth-exhausted paths / sample
    assert cnt.rays / cnt.samples == pytest.approx(2.77, abs=0.01)
    assert cnt.rng_words / cnt.samples == pytest.approx(14.4, abs=0.1)
    assert cnt.depth_exhausted / cnt.samples == pytest.approx(0.393, abs=0.002)
    # rows are the unit of parallelism (renderer.rs:87-91): one thread or many, the same image
    p1, l1, c1 = oracle_mod.render(sc, sc.camera, sc.settings, opt, threads=1)
    assert np.array_equal(p1, packed) and np.array_equal(l1.view(np.uint32), linear.view(np.uint32)) and c1.rays == cnt.rays
    # output stage: 0x00RRGGBB -> 8-bit RGB PNG (renderer.rs:125-143), read back by PIL
    from PIL import Image
    path = str(tmp_path / "cfg1.png")
    host.write_png(path, packed, W, H)
    got = np.array(Image.open(path).convert("RGB")).astype(np.uint32)
    assert np.array_equal((got[..., 0] << 16) | (got[..., 1] << 8) | got[..., 2], packed)
    # the picture is the cornell box seen from outside: Color::GRAY sky (0xB4B4B4, renderer.rs:61) left and right of it, a red wall,
    # a green wall, the light in the ceiling (16 spp is noisy: region means)
    rgb = got.astype(np.float64)
    assert (packed[:, :48] == 0xB4B4B4).all() and (packed[:, -48:] == 0xB4B4B4).all()
    left, right = rgb[40:260, 53:60].mean((0, 1)), rgb[40:260, 338:346].mean((0, 1))
    assert left[0] > 2 * left[1] and right[1] > 1.4 * right[0]
    assert rgb[18:24, 170:210].mean() > 250


def test_cfg1_oracle_counter_mode_digest(native, oracle_mod, abi):
    host, _ = native
    fx = _fixture()
    sc = host.LoadedScene(SCENES["cornell"], W, H, SPP, DEPTH)
    packed, linear, cnt = oracle_mod.render(sc, sc.camera, sc.settings, abi.Options.make(rng_mode=abi.RNG_CTR))
    assert _sha(packed) == fx["ctr"]["packed_sha256"] and _sha(linear) == fx["ctr"]["linear_sha256"] and cnt.rays == fx["ctr"]["rays"]
    assert cnt.rays / cnt.samples == pytest.approx(2.77, abs=0.01)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["ctr", "ref"])
def test_cfg1_hip_path_whole_image_equals_the_oracle(tag, native, oracle_mod, abi):
    host, device = native
    fx = _fixture()
    sc = host.LoadedScene(SCENES["cornell"], W, H, SPP, DEPTH)                      # product loader end to end
    opt = abi.Options.make(rng_mode=abi.RNG_CTR if tag == "ctr" else abi.RNG_REF)
    gp, gl, st = device.render(sc, sc.camera, sc.settings, opt)                     # mi355rt_render: what src/main.rs:57 would call
    op, ol, cnt = oracle_mod.render(sc, sc.camera, sc.settings, opt)
    assert st.samples == W * H * SPP == cnt.samples and st.rows_rendered == H
    assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32)), f"{(np.abs(gl - ol).max(-1) > 0).sum()} pixels differ"
    assert np.array_equal(gp, op)
    assert st.rays == cnt.rays == fx[tag]["rays"]
    assert st.rays / st.samples == pytest.approx(2.77, abs=0.01)
    assert _sha(gp) == fx[tag]["packed_sha256"] and _sha(gl) == fx[tag]["linear_sha256"]
