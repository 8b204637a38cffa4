"""Known-answer tests for the BSDFs and primitives the reference-held golden image does not exercise:
Plastic, checker, Metal (fuzz 0 and > 0), Dielectric (both faces, TIR), Beckmann / GGX, sphere (outside / inside /
tangent) and quad (edges, back face, parallel ray).  The expected values come from tests/kat_f32.py -- numpy float32
written from the Rust formulas -- and must equal the oracle's per-function hooks bit for bit (transcendental BSDFs: a
few ulps).  The gpu twin runs the same cases through the device code (k_debug_scatter / k_debug_hit behind the C ABI
library's diagnostic hooks).
"""
import ctypes as C

import numpy as np
import pytest

import kat_f32 as K
from conftest import pkg

f32 = np.float32


def _bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def material(abi, kind, albedo=(0, 0, 0), aux=(0, 0, 0), p0=0.0, eta=(0, 0, 0), k=(0, 0, 0)):
    m = abi.Material()
    m.kind = kind
    m.albedo[:] = albedo; m.aux[:] = aux; m.p0 = p0; m.eta[:] = eta; m.k[:] = k
    return m


def unit(v):
    return K.normalized(K.V(*v))


def scatter_cases(abi):
    """(name, abi.Material, expected-fn(rd, p, n, front, draws), transcendental?)"""
    cu_eta, cu_k = (0.200, 1.090, 1.420), (3.910, 2.570, 2.300)          # MetalType::Cu, tungsten/materials.rs:118-121
    al = lambda c: K.V(*c)
    return [
        ("lambert", material(abi, abi.MAT_LAMBERT_SOLID, (0.7, 0.6, 0.5)),
         lambda rd, p, n, ff, d: K.scatter_lambert(al((0.7, 0.6, 0.5)), rd, p, n, d), False),
        ("checker", material(abi, abi.MAT_LAMBERT_CHECKER, (0.9, 0.8, 0.1), (0.1, 0.2, 0.3), 1.0 / 0.37),
         lambda rd, p, n, ff, d: K.scatter_lambert(K.checker_value(al((0.9, 0.8, 0.1)), al((0.1, 0.2, 0.3)), f32(1.0 / 0.37), p), rd, p, n, d), False),
        ("metal_mirror", material(abi, abi.MAT_METAL, (0.8, 0.8, 0.9), p0=0.0),
         lambda rd, p, n, ff, d: K.scatter_metal(al((0.8, 0.8, 0.9)), f32(0.0), rd, p, n, d), False),
        ("metal_fuzz", material(abi, abi.MAT_METAL, (0.8, 0.6, 0.2), p0=0.35),
         lambda rd, p, n, ff, d: K.scatter_metal(al((0.8, 0.6, 0.2)), f32(0.35), rd, p, n, d), False),
        ("glass", material(abi, abi.MAT_DIELECTRIC, p0=1.5),
         lambda rd, p, n, ff, d: K.scatter_dielectric(f32(1.5), ff, rd, p, n, d), False),
        ("plastic", material(abi, abi.MAT_PLASTIC, (0.2, 0.5, 0.9), p0=1.9),
         lambda rd, p, n, ff, d: K.scatter_plastic(al((0.2, 0.5, 0.9)), f32(1.9), rd, p, n, d), False),
        ("beckmann_cu", material(abi, abi.MAT_ROUGH_BECKMANN, (1, 1, 1), p0=0.1, eta=cu_eta, k=cu_k),
         lambda rd, p, n, ff, d: K.scatter_rough(al((1, 1, 1)), f32(0.1), al(cu_eta), al(cu_k), False, rd, p, n, d), True),
        ("ggx_cu", material(abi, abi.MAT_ROUGH_GGX, (0.9, 0.95, 1.0), p0=0.3, eta=cu_eta, k=cu_k),
         lambda rd, p, n, ff, d: K.scatter_rough(al((0.9, 0.95, 1.0)), f32(0.3), al(cu_eta), al(cu_k), True, rd, p, n, d), True),
    ]


def scatter_inputs():
    """(rd, p, n, front_face, counters): incidence from normal to grazing, both faces, negative checker cells, axis-aligned and
    oblique normals (to_world's two `up` choices)."""
    rng = np.random.default_rng(7)
    out = []
    normals = [unit((0, 1, 0)), unit((0, 0, 1)), unit((0.3, -0.8, 0.52)), unit((-1, 0.02, 0.01))]
    for i in range(40):
        n = normals[i % len(normals)]
        # a direction against the normal (the hit record's normal always faces the ray), from steep to grazing
        t = K.normalized(K.cross(n, K.V(0.37, 0.61, -0.7)))
        graze = f32([0.02, 0.3, 0.7, 0.95, 0.999][i % 5])
        rd = K.normalized(t * graze - n * f32(np.sqrt(max(0.0, 1.0 - float(graze) ** 2))))
        p = K.V(*(rng.uniform(-3, 3, 3)))
        ctr = tuple(int(v) for v in rng.integers(0, 2 ** 32, 2)) + (int(rng.integers(0, 800)), int(rng.integers(0, 256)), int(rng.integers(1, 30)))
        out.append((rd, p, n, (i % 3) != 0, ctr))
    return out


def hit_scene_for(abi, prims):
    sc = abi.Scene()
    arr = (abi.Primitive * len(prims))(*prims)
    mats = (abi.Material * 1)(material(abi, abi.MAT_LAMBERT_SOLID, (0.5, 0.5, 0.5)))
    sc.primitives, sc.n_primitives, sc.materials, sc.n_materials = arr, len(prims), mats, 1
    sc.miss_color[:] = (0.5, 0.5, 0.5)
    sc._keep = (arr, mats)
    return sc


def sphere_prim(abi, c, r):
    p = abi.Primitive(); p.kind = abi.PRIM_SPHERE; p.material = 0
    p.data[0:4] = [c[0], c[1], c[2], r]
    return p


def quad_prim(abi, q):
    p = abi.Primitive(); p.kind = abi.PRIM_QUAD; p.material = 0
    p.data[0:15] = [*q["base"], *q["e0"], *q["e1"], *q["n"], q["d"], q["inv0"], q["inv1"]]
    return p


def hit_cases(abi):
    """(name, abi.Scene, rays[(o, d)], expected-fn(o, d_normalised))"""
    c, r = K.V(0.25, -0.5, 2.0), f32(1.3)
    q = K.quad_from_corners(K.V(-1.0, 0.2, 3.0), K.V(2.0, 0.1, 0.3), K.V(-0.2, 1.5, 0.4))
    rays_s = [(K.V(0, 0, -5), (0.05, -0.1, 1)), (K.V(0, 0, -5), (0.25, -0.5, 7.0)), (K.V(0.25, -0.5, 2.0), (0.3, 0.2, -1)),      # outside, centre, from inside
              (K.V(0.3, -0.4, 2.2), (-1, 0.5, 0.2)), (K.V(1.55, -0.5, -4), (0, 0, 1)), (K.V(1.5501, -0.5, -4), (0, 0, 1)),           # inside, tangent, just missing
              (K.V(0, 0, 5), (0, 0, 1)), (K.V(5, 5, 5), (-1, -1.1, -0.6))]                                                           # behind, oblique
    rays_q = [(K.V(0, 1, -2), (0, 0, 1)), (K.V(0, 1, 8), (0, 0, -1)),                                  # front and back face
              (K.V(-1.0, 0.2, -2), (0, 0, 1)), (K.V(-1.00005, 0.2, -2), (0, 0, 1)), (K.V(-1.001, 0.2, -2), (0, 0, 1)),   # on the corner, inside the EPS rim, outside it
              (K.V(0, 1, -2), (2.0, 0.1, 0.3)), (K.V(0.9, 1.7, -2), (0.01, -0.02, 1)), (K.V(0, 1, 3.5), (0, 0, 1))]       # parallel to the plane, near the far edge, behind
    return [
        ("sphere", hit_scene_for(abi, [sphere_prim(abi, c, r)]), rays_s, lambda o, d: K.hit_sphere(c, r, o, d, K.EPS, f32(np.inf))),
        ("quad", hit_scene_for(abi, [quad_prim(abi, q)]), rays_q, lambda o, d: K.hit_quad(q, o, d, K.EPS, f32(np.inf))),
    ]


def texture_fixture(abi):
    """An 8 x 4 RGBA8 image with a distinct colour per texel + the material that samples it (h_offset 0.3)."""
    rgba = np.zeros((4, 8, 4), np.uint8)
    for y in range(4):
        for x in range(8):
            rgba[y, x] = (10 + 30 * x, 20 + 50 * y, 255 - 25 * x - 7 * y, 255)
    tex = abi.Texture(rgba.ctypes.data_as(C.POINTER(C.c_uint8)), 8, 4)
    textures = (abi.Texture * 1)(tex)
    mat = material(abi, abi.MAT_TEXTURE, (0.9, 0.8, 0.7), p0=0.3)
    mat.texture = 0
    return rgba, textures, mat


def texture_normals():
    """Unit normals whose equirect coordinates stay at least 2 % of a texel away from every texel border (acos / atan2 ulps
    cannot change the texel), over all octants and both poles."""
    rgba = np.zeros((4, 8, 4), np.uint8)
    rng = np.random.default_rng(11)
    out = []
    while len(out) < 48:
        n = K.normalized(K.V(*rng.normal(size=3)))
        _, (fx, fy) = K.texture_value(rgba, f32(0.3), n)
        if min(fx % 1, 1 - fx % 1, fy % 1, 1 - fy % 1) > 0.02:
            out.append(n)
    return out + [K.V(0, 1, 0), K.V(0, -1, 0)]


def check_scatter(name, want, got, transcendental):
    ok, o, d, a = want
    assert bool(got[0]) == ok, name
    if not ok:
        return
    if transcendental:                       # ln / atan / sin / cos differ by ulps between libms
        assert np.allclose(got[1:4], o, rtol=2e-6, atol=1e-6) and np.allclose(got[4:7], d, rtol=1e-5, atol=2e-6) and np.allclose(got[7:10], a, rtol=2e-4, atol=1e-6), name
    else:
        assert np.array_equal(_bits(got[1:4]), _bits(o)) and np.array_equal(_bits(got[4:7]), _bits(d)) and np.array_equal(_bits(got[7:10]), _bits(a)), \
            (name, got[1:10], o, d, a)


def check_hit(name, want, hit, out9):
    assert hit == (want is not None), name
    if want is None:
        return
    t, p, n, front = want
    assert np.array_equal(_bits(out9[0:3]), _bits(p)) and np.array_equal(_bits(out9[3:6]), _bits(n)) and _bits([out9[6]])[0] == _bits([t])[0], (name, out9, want)
    assert bool(out9[8]) == bool(front), name


def test_philox_and_float_conversions_of_the_independent_restatement(oracle_mod):
    for args in [(0, 0, 0, 0, 0, 0), (0xa4093822, 0x299f31d0, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (1, 2, 3, 4, 5, 6)]:
        assert K.philox(*args) == [int(v) for v in oracle_mod.philox(*args)]
        assert K.pcg4d(*args[:4]) == [int(v) for v in oracle_mod.pcg4d(*args[:4])]
        assert K.ctr_block(*args) == [int(v) for v in oracle_mod.ctr_block(*args, gen=2)] == [int(v) for v in oracle_mod.ctr_block(*args)]   # (2 = the generator in force)
    for w in [0, 1, 0xFF, 0x100, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0x12345678]:
        assert _bits([K.u01(w)])[0] == _bits([oracle_mod.lib().oracle_u32_to_f01(w)])[0]
        assert _bits([K.range11(w)])[0] == _bits([oracle_mod.lib().oracle_u32_to_range11(w)])[0]


def test_oracle_scatter_equals_the_numpy_known_answers(oracle_mod, abi):
    L = oracle_mod.lib()
    n_checked = 0
    for name, mat, expect, transcendental in scatter_cases(abi):
        for rd, p, n, ff, ctr in scatter_inputs():
            want = expect(rd, p, n, ff, K.CtrDraws(*ctr))
            out = np.zeros(10, np.float32)
            o = np.zeros(3, np.float32)
            rc = L.oracle_scatter_ctr(C.byref(mat), o.ctypes.data, rd.ctypes.data, p.ctypes.data, n.ctypes.data, int(ff), *ctr, out.ctypes.data)
            assert rc == 0
            check_scatter(name, want, out, transcendental)
            n_checked += 1
    assert n_checked == 8 * 40


def test_oracle_texture_material_equals_the_numpy_known_answers(oracle_mod, abi):
    """TextureMaterial (parser.rs:199-243): Lambert bounce, attenuation = albedo * the texel the hit NORMAL selects."""
    rgba, textures, mat = texture_fixture(abi)
    L = oracle_mod.lib()
    L.oracle_set_textures.argtypes = [C.POINTER(abi.Texture), C.c_uint32]
    L.oracle_set_textures(textures, 1)
    try:
        seen = set()
        for i, n in enumerate(texture_normals()):
            rd, p, ctr = K.normalized(-n + K.V(0.1, 0.05, -0.02)), K.V(0.5, -1.0, 2.0), (7, 9, i, 3, 2)
            want = K.scatter_texture(K.V(0.9, 0.8, 0.7), rgba, f32(0.3), rd, p, n, K.CtrDraws(*ctr))
            out = np.zeros(10, np.float32)
            assert L.oracle_scatter_ctr(C.byref(mat), np.zeros(3, np.float32).ctypes.data, rd.ctypes.data, p.ctypes.data, n.ctypes.data, 1, *ctr, out.ctypes.data) == 0
            check_scatter("texture", want, out, False)
            seen.add(tuple(out[7:10]))
        assert len(seen) >= 20                                  # many different texels were reached
    finally:
        L.oracle_set_textures(None, 0)


def test_dielectric_total_internal_reflection_takes_no_draw(oracle_mod, abi):
    """Inside glass at a grazing angle: cannot_refract short-circuits the random draw (material.rs:145) and the ray reflects."""
    n = unit((0, 1, 0))
    rd = K.normalized(K.V(0.9, -0.2, 0.1))
    p = K.V(0.1, 0.2, 0.3)
    want = K.scatter_dielectric(f32(1.5), False, rd, p, n, K.CtrDraws(1, 2, 3, 4, 5))
    assert np.array_equal(_bits(want[2]), _bits(K.normalized(K.normalized(K.reflect(K.normalized(rd), n)))))
    out = np.zeros(10, np.float32)
    mat = material(abi, abi.MAT_DIELECTRIC, p0=1.5)
    oracle_mod.lib().oracle_scatter_ctr(C.byref(mat), np.zeros(3, np.float32).ctypes.data, rd.ctypes.data, p.ctypes.data, n.ctypes.data, 0, 1, 2, 3, 4, 5, out.ctypes.data)
    check_scatter("tir", want, out, False)


def test_oracle_primitive_hits_equal_the_numpy_known_answers(oracle_mod, abi):
    for name, sc, rays, expect in hit_cases(abi):
        for o, d in rays:
            dn = K.normalized(K.V(*d))                           # Ray::new
            hit, out9 = oracle_mod.scene_hit(sc, o, np.asarray(d, np.float32))
            check_hit(name, expect(o, dn), hit, out9)


@pytest.mark.gpu
def test_gpu_scatter_and_hits_equal_the_numpy_known_answers(native, abi):
    """The same known answers through the DEVICE code: scatter_pre / diffuse_finish / hit_scene as the render kernels call them."""
    host, device = native
    cases = scatter_cases(abi)
    mats = (abi.Material * len(cases))(*[c[1] for c in cases])
    inputs = scatter_inputs()
    recs, wants = [], []
    for mi, (name, _, expect, transcendental) in enumerate(cases):
        for rd, p, n, ff, ctr in inputs:
            recs.append((mi, ff, rd, p, n, ctr))
            wants.append((name, expect(rd, p, n, ff, K.CtrDraws(*ctr)), transcendental))
    got = device.debug_scatter(mats, recs)
    for (name, want, transcendental), g in zip(wants, got):
        check_scatter(name, want, g, transcendental)
    rgba, textures, tmat = texture_fixture(abi)
    trecs, twants = [], []
    for i, n in enumerate(texture_normals()):
        rd, p, ctr = K.normalized(-n + K.V(0.1, 0.05, -0.02)), K.V(0.5, -1.0, 2.0), (7, 9, i, 3, 2)
        trecs.append((0, True, rd, p, n, ctr))
        twants.append(K.scatter_texture(K.V(0.9, 0.8, 0.7), rgba, f32(0.3), rd, p, n, K.CtrDraws(*ctr)))
    for want, g in zip(twants, device.debug_scatter((abi.Material * 1)(tmat), trecs, textures=textures)):
        check_scatter("texture", want, g, False)
    for name, sc, rays, expect in hit_cases(abi):
        out = device.debug_hit(sc, [(o, K.V(*d)) for o, d in rays])
        for (o, d), g in zip(rays, out):
            check_hit(name, expect(o, K.normalized(K.V(*d))), bool(g[9]), g[:9])
