"""HDR skybox on miss (renderer.rs:40-54) and the `sky.texture` loader path (parser.rs:497-509)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import SCENES, load_for_both


def _write_hdr(path, img_rgbe, rle):
    """img_rgbe: uint8 [H, W, 4].  Writes a Radiance file, flat or new-style RLE (runs and literals mixed)."""
    H, W, _ = img_rgbe.shape
    out = bytearray(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n" + f"-Y {H} +X {W}\n".encode())
    for y in range(H):
        if not rle:
            out += img_rgbe[y].tobytes(); continue
        out += bytes([2, 2, W >> 8, W & 255])
        for c in range(4):
            row = img_rgbe[y, :, c]; x = 0
            while x < W:
                run = 1
                while x + run < W and run < 127 and row[x + run] == row[x]:
                    run += 1
                if run >= 3:
                    out += bytes([128 + run, int(row[x])]); x += run
                else:
                    n = min(W - x, 5)
                    out += bytes([n]) + row[x:x + n].tobytes(); x += n
    open(path, "wb").write(bytes(out))


def _attach_sky(sc, abi, sky):
    sc._sky = np.ascontiguousarray(sky, np.float32)
    sc.c.sky_height, sc.c.sky_width = sc._sky.shape[:2]
    sc.c.sky_rgb = sc._sky.ctypes.data_as(C.POINTER(C.c_float))


def _empty_scene(abi):
    s = abi.Scene(); s.miss_color[:] = [0.5, 0.5, 0.5]
    cam = abi.Camera(); cam.position[:] = [0, 0, 0]; cam.forward[:] = [0, 0, -1]; cam.right[:] = [1, 0, 0]; cam.true_up[:] = [0, 1, 0]
    cam.half_width, cam.half_height = 1.0, 0.75
    class Box: pass
    b = Box(); b.c = s
    return b, cam


@pytest.mark.parametrize("mode", [0, 1])
def test_oracle_skybox_lookup_known_answers(mode, oracle_mod, abi):
    b, cam = _empty_scene(abi)
    # uniform sky: every pixel is the sky colour, whatever the direction
    _attach_sky(b, abi, np.tile(np.array([0.25, 1.5, 0.04], np.float32), (4, 8, 1)))
    st = abi.Settings(16, 12, 3, 4)
    packed, lin, cnt = oracle_mod.render(b, cam, st, abi.Options.make(rng_mode=mode))
    assert np.allclose(lin, [0.25, 1.5, 0.04], rtol=1e-6) and cnt.rays == cnt.samples
    # v = acos(dir.y) / pi, y_pixel = (v * 3) truncated.  Along the centre column dir.y runs from +0.6 (top, v*3 = 0.885 -> row 0)
    # through 0 (v*3 = 1.5 -> row 1) to -0.6 (bottom, v*3 = 2.11 -> row 2); row 3 is never reached.
    sky = np.zeros((4, 2, 3), np.float32)
    for r in range(4):
        sky[r] = 10.0 ** r
    _attach_sky(b, abi, sky)
    _, lin, _ = oracle_mod.render(b, cam, st, abi.Options.make(rng_mode=mode))
    assert np.allclose(lin[0, 7:9], 1.0) and np.allclose(lin[5:7, 7:9], 10.0) and np.allclose(lin[-1, 7:9], 100.0)
    assert lin.max() <= 100.0 and lin.min() >= 1.0
    # u = (atan2(dir.z, dir.x) + pi) / 2pi: forward = -z -> atan2 in (-pi, 0) -> u in (0, 0.5): left half of the map only
    sky = np.zeros((1, 4, 3), np.float32); sky[0, :, 0] = [1, 2, 3, 4]
    _attach_sky(b, abi, sky)
    _, lin, _ = oracle_mod.render(b, cam, st, abi.Options.make(rng_mode=mode))
    assert lin[..., 0].min() >= 1.0 and lin[..., 0].max() <= 2.0 and lin[:, 0, 0].max() == 1.0 and lin[:, -1, 0].min() == 2.0   # x = u * 3 truncated: texels 0 and 1 only


def test_hdr_decoder_and_loader_paths(native, tmp_path, abi):
    host, _ = native
    from oracle import scene_loader
    rng = np.random.default_rng(7)
    H, W = 6, 20
    img = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
    img[:, :, 3] = rng.integers(120, 136, size=(H, W))
    img[2, 3:12] = img[2, 3]; img[4, :, :] = img[4, 0]            # runs for the RLE coder
    img[0, 0, 3] = 0                                               # e == 0 -> black
    base = json.load(open(SCENES["cornell"]))
    for rle in (False, True):
        name = f"sky_{int(rle)}.hdr"
        _write_hdr(str(tmp_path / name), img, rle)
        got = scene_loader.load_radiance_hdr(str(tmp_path / name))
        want = (np.exp2(img[..., 3].astype(np.float32) - 136.0)[..., None] * img[..., :3].astype(np.float32)) * (img[..., 3:4] != 0)
        assert got.shape == (H, W, 3) and np.array_equal(got, want.astype(np.float32))
        base["sky"] = {"texture": name}
        p = tmp_path / f"scene_{int(rle)}.json"
        p.write_text(json.dumps(base))
        a = host.LoadedScene(str(p)); b = scene_loader.load_scene(str(p))
        assert (a.c.sky_width, a.c.sky_height) == (W, H) == (b.c.sky_width, b.c.sky_height)
        ga = np.ctypeslib.as_array(a.c.sky_rgb, shape=(H * W * 3,)); gb = np.ctypeslib.as_array(b.c.sky_rgb, shape=(H * W * 3,))
        assert np.array_equal(ga, gb) and np.array_equal(ga.reshape(H, W, 3), got)
    # LDR sky textures are loaded by the reference but never sampled; a missing file keeps the default background
    for tex in ("sky.png", "missing.hdr"):
        base["sky"] = {"texture": tex}
        p = tmp_path / "scene_x.json"; p.write_text(json.dumps(base))
        a = host.LoadedScene(str(p)); b = scene_loader.load_scene(str(p))
        assert a.c.sky_width == 0 and not a.c.sky_rgb and b.c.sky_width == 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_hip_skybox_matches_oracle(mode, native, oracle_mod, abi):
    """acosf/atan2f on the device differ from glibc by ulps, so a texel boundary can flip for isolated samples."""
    host, device = native
    rng = np.random.default_rng(3)
    b, cam = _empty_scene(abi)
    _attach_sky(b, abi, np.tile(np.array([0.25, 1.5, 0.04], np.float32), (4, 8, 1)))
    st = abi.Settings(16, 12, 3, 4)
    gp, gl, _ = device.render(b, cam, st, abi.Options.make(rng_mode=mode))
    op, ol, _ = oracle_mod.render(b, cam, st, abi.Options.make(rng_mode=mode))
    assert np.array_equal(gl, ol) and np.array_equal(gp, op)
    sc = load_for_both("cornell", oracle_mod, host, width=64, height=48, spp=8, max_depth=6)
    _attach_sky(sc, abi, rng.uniform(0.0, 2.0, size=(32, 64, 3)).astype(np.float32))
    gp, gl, gs = device.render(sc, sc.camera, sc.settings, abi.Options.make(rng_mode=mode))
    op, ol, cnt = oracle_mod.render(sc, sc.camera, sc.settings, abi.Options.make(rng_mode=mode))
    assert gs.rays == cnt.rays
    l2 = np.sqrt(((gl.astype(np.float64) - ol) ** 2).sum(-1))
    assert (l2 <= 1e-3).mean() >= 0.99 and abs(gl.mean() - ol.mean()) < 1e-3 * ol.mean()
