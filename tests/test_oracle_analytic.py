"""Known answers derivable from the reference's code alone, on the oracle (CPU) and,
marked gpu, on the HIP path."""
import ctypes as C

import numpy as np
import pytest


def _scene(abi, prims=(), mats=()):
    s = abi.Scene()
    P = (abi.Primitive * max(len(prims), 1))(*prims)
    M = (abi.Material * max(len(mats), 1))(*mats)
    s.primitives, s.n_primitives = P, len(prims)
    s.materials, s.n_materials = M, len(mats)
    s.miss_color[:] = [0.5, 0.5, 0.5]
    s._keep = (P, M)
    return s


def _camera(abi):
    c = abi.Camera()
    c.position[:] = [0, 0, 0]; c.forward[:] = [0, 0, -1]; c.right[:] = [1, 0, 0]; c.true_up[:] = [0, 1, 0]
    c.half_width, c.half_height = 0.4, 0.3
    return c


def _mat(abi, kind, albedo=(0, 0, 0), p0=0.0):
    m = abi.Material(); m.kind = kind; m.albedo[:] = albedo; m.p0 = p0
    return m


def _quad_facing_camera(abi, z, half, material):
    p = abi.Primitive(); p.kind, p.material = abi.PRIM_QUAD, material
    base, e0, e1, n = (-half, -half, z), (2 * half, 0, 0), (0, 2 * half, 0), (0, 0, 1)
    inv = 1.0 / (4 * half * half)
    p.data[0:15] = [*base, *e0, *e1, *n, z, inv, inv]
    return p


def _sphere(abi, c, r, material):
    p = abi.Primitive(); p.kind, p.material = abi.PRIM_SPHERE, material
    p.data[0:4] = [*c, r]
    return p


def _renderers(request, oracle_mod):
    yield "oracle", lambda sc, cam, st, opt: oracle_mod.render(sc, cam, st, opt)[:2]


CASES = ["empty", "emissive", "exhaust", "absorb"]


def _build(abi, case):
    cam = _camera(abi)
    if case == "empty":            # no objects: every path misses -> GRAY -> sqrt(0.5)*255 = 180.3 -> 0xB4
        return _scene(abi), cam, abi.Settings(16, 12, 4, 5), 0xB4B4B4, (0.5, 0.5, 0.5)
    if case == "emissive":         # emissive quad filling the view: pixel = e, packed = clamp(sqrt(e))
        sc = _scene(abi, [_quad_facing_camera(abi, -1.0, 10.0, 0)], [_mat(abi, abi.MAT_EMISSIVE, (0.25, 4.0, 0.0625))])
        return sc, cam, abi.Settings(16, 12, 4, 5), (127 << 16) | (255 << 8) | 63, (0.25, 4.0, 0.0625)
    if case == "exhaust":          # camera inside a white Lambert sphere: paths never end -> depth exhaustion -> black
        sc = _scene(abi, [_sphere(abi, (0, 0, 0), 5.0, 0)], [_mat(abi, abi.MAT_LAMBERT_SOLID, (1, 1, 1))])
        return sc, cam, abi.Settings(16, 12, 4, 7), 0, (0.0, 0.0, 0.0)
    if case == "absorb":           # NullMaterial: scatter -> None, emitted black
        sc = _scene(abi, [_quad_facing_camera(abi, -1.0, 10.0, 0)], [_mat(abi, abi.MAT_NULL)])
        return sc, cam, abi.Settings(16, 12, 4, 5), 0, (0.0, 0.0, 0.0)
    raise KeyError(case)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", [0, 1])
def test_oracle_known_answers(case, mode, oracle_mod, abi):
    sc, cam, st, want_packed, want_lin = _build(abi, case)
    packed, lin, cnt = oracle_mod.render(sc, cam, st, abi.Options.make(rng_mode=mode))
    assert np.all(packed == want_packed), hex(int(packed[0, 0]))
    assert np.allclose(lin, np.array(want_lin, np.float32), rtol=1e-6, atol=0)
    if case == "exhaust":
        assert cnt.rays == cnt.samples * 7 and cnt.depth_exhausted == cnt.samples


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mode", [0, 1])
def test_hip_known_answers(case, mode, native, abi):
    _, device = native
    sc, cam, st, want_packed, want_lin = _build(abi, case)
    packed, lin, stats = device.render(sc, cam, st, abi.Options.make(rng_mode=mode))
    assert np.all(packed == want_packed)
    assert np.allclose(lin, np.array(want_lin, np.float32), rtol=1e-6, atol=0)
    if case == "exhaust":
        assert stats.rays == stats.samples * 7


def test_pack_truncates_and_nan_is_black(oracle_mod):
    L = oracle_mod.lib()
    assert L.oracle_color_to_u32(1.0, 0.25, 0.0) == (255 << 16) | (127 << 8)       # sqrt(0.25)*255 = 127.5 -> 127 (truncation)
    assert L.oracle_color_to_u32(float("nan"), -1.0, 9.0) == 255                      # NaN -> 0, negative -> NaN -> 0, >1 clamps
