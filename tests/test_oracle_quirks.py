"""Hand-derived known answers for the oracle's intersection routines and for the reference's load-bearing quirks
(SURVEY.md Appendix B), through oracle_scene_hit (HittableList::hit with t_min = 1e-4, t_max = inf)."""
import ctypes as C

import numpy as np
import pytest

EPS = np.float32(1e-4)
IDENT = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]


def _scene(abi, prims, tris=None, host=None):
    s = abi.Scene()
    P = (abi.Primitive * max(len(prims), 1))(*prims)
    M = (abi.Material * 4)()
    s.primitives, s.n_primitives, s.materials, s.n_materials = P, len(prims), M, 4
    s.miss_color[:] = [0.5] * 3
    keep = [P, M]
    if tris is not None:
        T = (abi.Triangle * len(tris))()
        C.memmove(T, np.ascontiguousarray(tris, np.float32).ctypes.data, len(tris) * 48)
        mesh = (abi.Mesh * 1)(); mesh[0].first_triangle, mesh[0].triangle_count = 0, len(tris)
        s.triangles, s.n_triangles, s.meshes, s.n_meshes = T, len(tris), mesh, 1
        keep += [T, mesh]
    class Box: pass
    b = Box(); b.c = s; b._keep = keep
    return b


def _prim(abi, kind, data, material=0, mesh=0):
    p = abi.Primitive(); p.kind, p.material, p.mesh = kind, material, mesh
    p.data[0:len(data)] = [float(v) for v in data]
    return p


def _quad_xz(abi, y, half=1.0, material=0):          # parallelogram in the plane Y = y, normal +Y (edge0 x edge1 with edge0 = +X*.., edge1 = +Z..)
    base, e0, e1 = (-half, y, -half), (2 * half, 0, 0), (0, 0, 2 * half)
    n = (0, -1, 0)                                    # (2h,0,0) x (0,0,2h) = (0*2h-0*0, 0*0-2h*2h, 0) -> -Y
    inv = 1.0 / (4 * half * half)
    return _prim(abi, abi.PRIM_QUAD, [*base, *e0, *e1, *n, -y, inv, inv], material)


def _tri(v0, v1, v2):
    v0, v1, v2 = (np.array(v, np.float32) for v in (v0, v1, v2))
    n = np.cross(v1 - v0, v2 - v0); n = n / np.linalg.norm(n)
    return np.concatenate([v0, v1, v2, n]).astype(np.float32)


def test_sphere_outside_inside_and_grazing(oracle_mod, abi):
    sc = _scene(abi, [_prim(abi, abi.PRIM_SPHERE, [0, 0, 0, 1])])
    hit, r = oracle_mod.scene_hit(sc, (0, 0, 5), (0, 0, -2))                 # direction is normalised by Ray::new
    assert hit and r[6] == pytest.approx(4.0, abs=1e-6) and list(r[3:6]) == [0, 0, 1] and r[8] == 1.0
    hit, r = oracle_mod.scene_hit(sc, (0, 0, 0), (1, 0, 0))                  # from the centre: far root, normal flipped, back face
    assert hit and r[6] == pytest.approx(1.0, abs=1e-6) and list(r[3:6]) == [-1, 0, 0] and r[8] == 0.0
    assert not oracle_mod.scene_hit(sc, (0, 2, 5), (0, 0, -1))[0]
    hit, r = oracle_mod.scene_hit(sc, (0, 0, 1), (0, 0, 1))                  # on the surface heading out: both roots <= t_min
    assert not hit


def _tiny_sphere_scene(abi, radius):
    sc = _scene(abi, [_prim(abi, abi.PRIM_SPHERE, [0, 0, 0, 1]), _prim(abi, abi.PRIM_SPHERE, [0, 2, 0, radius])])
    cam = abi.Camera(); cam.position[:] = [0, 0, 8]; cam.forward[:] = [0, 0, -1]; cam.right[:] = [1, 0, 0]; cam.true_up[:] = [0, 1, 0]
    cam.half_width, cam.half_height = 0.4, 0.3
    return sc, cam, abi.Settings(8, 6, 1, 2)


@pytest.mark.parametrize("radius", [0.0, 9.9e-5, -5e-5])
def test_a_sphere_the_reference_panics_on_is_refused_by_the_oracle(radius, oracle_mod, abi):
    """sphere.rs:38 computes the normal as `(position - center) / radius` with Vec3's `/ f32`, which panics for |radius| < 1e-4
    (/root/reference/src/vec3.rs:120-122): the reference cannot render such a scene, so the oracle does not either."""
    sc, cam, st = _tiny_sphere_scene(abi, radius)
    with pytest.raises(RuntimeError, match="oracle_render failed: -1"):
        oracle_mod.render(sc, cam, st, abi.Options.make())
    ok, cam, st = _tiny_sphere_scene(abi, 1e-4)                              # exactly EPSILON: `abs() < EPSILON` is false, it renders
    oracle_mod.render(ok, cam, st, abi.Options.make())
    neg, cam, st = _tiny_sphere_scene(abi, -0.5)                             # a negative radius of ordinary size is the reference's business: allowed
    oracle_mod.render(neg, cam, st, abi.Options.make())


@pytest.mark.gpu
def test_hip_path_refuses_the_sphere_the_reference_panics_on(native, oracle_mod, abi):
    _, device = native
    for radius in (0.0, 9.9e-5, -5e-5):
        sc, cam, st = _tiny_sphere_scene(abi, radius)
        with pytest.raises(device.RenderError, match="sphere radius") as e:
            device.render(sc, cam, st, abi.Options.make())
        assert e.value.rc == abi.ERR_INVALID
    for radius in (1e-4, -0.5):                                             # what the reference renders, both sides render alike
        sc, cam, st = _tiny_sphere_scene(abi, radius)
        gp, gl, gs = device.render(sc, cam, st, abi.Options.make())
        op, ol, cnt = oracle_mod.render(sc, cam, st, abi.Options.make())
        assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32)) and np.array_equal(gp, op) and gs.rays == cnt.rays


def test_quad_edges_use_epsilon_inclusive_bounds(oracle_mod, abi):
    sc = _scene(abi, [_quad_xz(abi, 0.0)])
    assert oracle_mod.scene_hit(sc, (0.25, 1, 0.25), (0, -1, 0))[0]
    # l0 = (x + 1) / 2 must lie in [-1e-4, 1 + 1e-4]  (quad.rs:103-110)
    assert oracle_mod.scene_hit(sc, (1.0 + 1.5e-4, 1, 0), (0, -1, 0))[0]       # l0 = 1 + 0.75e-4: inside
    assert not oracle_mod.scene_hit(sc, (1.0 + 3e-4, 1, 0), (0, -1, 0))[0]     # l0 = 1 + 1.5e-4: outside
    assert not oracle_mod.scene_hit(sc, (0, 1, 0), (1, 0, 0))[0]               # parallel: |denom| < 1e-4
    hit, r = oracle_mod.scene_hit(sc, (0, -1, 0), (0, 1, 0))                   # from below: normal (0,-1,0) faces the ray? dot(d, n) = -1 < 0 -> front
    assert hit and list(r[3:6]) == [0, -1, 0] and r[8] == 1.0


def test_first_primitive_wins_exact_ties(oracle_mod, abi):
    a, b = _quad_xz(abi, 0.0, material=1), _quad_xz(abi, 0.0, material=2)
    assert oracle_mod.scene_hit(_scene(abi, [a, b]), (0, 1, 0), (0, -1, 0))[1][7] == 1      # quad: t >= t_max rejects the later one
    assert oracle_mod.scene_hit(_scene(abi, [b, a]), (0, 1, 0), (0, -1, 0))[1][7] == 2
    cube = _prim(abi, abi.PRIM_CUBE, [2, 0, 0, 0, 0, 1, 0, 0, 0, 0, 2, 0, 0, -0.5, 0, 1,          # o2w: scale (2,1,2), top face at y = 0
                                      0.5, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0.5, 0, 0, 0.5, 0, 1], material=3)
    assert oracle_mod.scene_hit(_scene(abi, [a, cube]), (0, 1, 0), (0, -1, 0))[1][7] == 1   # cube.rs:101 t_hit >= t_max
    assert oracle_mod.scene_hit(_scene(abi, [cube, a]), (0, 1, 0), (0, -1, 0))[1][7] == 3


def test_cube_face_normal_and_world_t(oracle_mod, abi):
    # unit cube scaled to 2 x 4 x 6 and moved to (10, 0, 0)
    o2w = [2, 0, 0, 0, 0, 4, 0, 0, 0, 0, 6, 0, 10, 0, 0, 1]
    w2o = [0.5, 0, 0, 0, 0, 0.25, 0, 0, 0, 0, 1 / 6, 0, -5, 0, 0, 1]
    sc = _scene(abi, [_prim(abi, abi.PRIM_CUBE, o2w + w2o)])
    hit, r = oracle_mod.scene_hit(sc, (0, 0.5, 1), (1, 0, 0))
    assert hit and r[6] == pytest.approx(9.0, rel=1e-6) and list(r[0:3]) == pytest.approx([9, 0.5, 1], rel=1e-6)
    assert list(r[3:6]) == [-1, 0, 0] and r[8] == 1.0
    hit, r = oracle_mod.scene_hit(sc, (10, 0, 0), (0, 0, 1))                   # from inside: exit face +z at 3, back face
    assert hit and r[6] == pytest.approx(3.0, rel=1e-6) and list(r[3:6]) == [0, 0, -1] and r[8] == 0.0


def test_zero_thickness_boxes_never_hit(native, oracle_mod, abi):
    """aabb.rs:40 rejects t_max <= t_min: a leaf whose triangles lie in an axis plane has min == max on that axis and is
    invisible; tilt it slightly and it is hit."""
    flat = [_tri((-1, -1, 0), (1, -1, 0), (0, 1, 0))]
    tilted = [_tri((-1, -1, 0), (1, -1, 0), (0, 1, 0.01))]
    mesh = lambda: _prim(abi, abi.PRIM_MESH, IDENT + IDENT)
    assert not oracle_mod.scene_hit(_scene(abi, [mesh()], flat), (0, 0, 5), (0, 0, -1))[0]
    hit, r = oracle_mod.scene_hit(_scene(abi, [mesh()], tilted), (0, 0, 5), (0, 0, -1))
    assert hit and r[6] == pytest.approx(4.995, abs=1e-3)


def test_mesh_t_world_is_multiplied_by_the_scale(native, oracle_mod, abi):
    """mesh_object.rs:312-314: t_world = t_obj * |d_obj| / |d_world| (it should divide).  With a uniform scale of 2 the
    reported t is a quarter of the true distance, so a mesh BEHIND a quad still wins the closest-hit test."""
    o2w = [2, 0, 0, 0, 0, 2, 0, 0, 0, 0, 2, 0, 0, 0, 0, 1]
    w2o = [0.5, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 1]
    tris = [_tri((-1, -1, -0.05), (1, -1, 0.05), (0, 1, 0.0))]                     # object space, around z = 0 -> world z ~ 0
    mesh = _prim(abi, abi.PRIM_MESH, o2w + w2o, material=2)
    sc = _scene(abi, [mesh], tris)
    hit, r = oracle_mod.scene_hit(sc, (0, 0, 8), (0, 0, -1))
    assert hit and r[2] == pytest.approx(0.0, abs=0.11)                             # the hit POSITION is right (world z ~ 0, 8 away)
    assert r[6] == pytest.approx(8.0 / 4.0, rel=0.02)                               # ...but t is 8 * 0.5 * 0.5
    quad_in_front = _prim(abi, abi.PRIM_QUAD, [-5, -5, 4, 10, 0, 0, 0, 10, 0, 0, 0, 1, 4, 0.01, 0.01], material=1)   # plane z = 4, 4 away
    # mesh listed first: it reports t = 2, so the quad at t = 4 -- which is in FRONT of it -- is rejected (t >= closest)
    hit, r = oracle_mod.scene_hit(_scene(abi, [mesh, quad_in_front], tris), (0, 0, 8), (0, 0, -1))
    assert hit and r[7] == 2 and r[6] == pytest.approx(2.0, rel=0.02)
    # quad listed first: closest = 4 is handed to the BVH as t_max of the OBJECT-space ray (mesh_object.rs:291), where the
    # triangle sits at t_obj = 4: `t < t_max` fails and the mesh is culled -- the other half of the same quirk
    hit, r = oracle_mod.scene_hit(_scene(abi, [quad_in_front, mesh], tris), (0, 0, 8), (0, 0, -1))
    assert hit and r[7] == 1 and r[6] == pytest.approx(4.0, rel=1e-6)


@pytest.mark.gpu
def test_hip_path_reproduces_the_quirks(native, oracle_mod, abi):
    """Same scenes through the C ABI: flat leaf invisible, scaled-mesh t quirk in both list orders -- bit-identical images."""
    host, device = native
    cam = abi.Camera(); cam.position[:] = [0, 0, 8]; cam.forward[:] = [0, 0, -1]; cam.right[:] = [1, 0, 0]; cam.true_up[:] = [0, 1, 0]
    cam.half_width, cam.half_height = 0.4, 0.3
    st = abi.Settings(32, 24, 4, 4)
    o2w = [2, 0, 0, 0, 0, 2, 0, 0, 0, 0, 2, 0, 0, 0, 0, 1]; w2o = [0.5, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 1]
    tri_scaled = [_tri((-1, -1, -0.05), (1, -1, 0.05), (0, 1, 0.0))]
    quad = lambda: _prim(abi, abi.PRIM_QUAD, [-5, -5, 4, 10, 0, 0, 0, 10, 0, 0, 0, 1, 4, 0.01, 0.01], material=1)
    mesh2 = lambda: _prim(abi, abi.PRIM_MESH, o2w + w2o, material=2)
    cases = {
        "flat leaf": ([_prim(abi, abi.PRIM_MESH, IDENT + IDENT)], [_tri((-1, -1, 0), (1, -1, 0), (0, 1, 0))]),
        "tilted leaf": ([_prim(abi, abi.PRIM_MESH, IDENT + IDENT)], [_tri((-1, -1, 0), (1, -1, 0), (0, 1, 0.01))]),
        "mesh then quad": ([mesh2(), quad()], tri_scaled),
        "quad then mesh": ([quad(), mesh2()], tri_scaled),
    }
    means = {}
    for name, (prims, tris) in cases.items():
        sc = _scene(abi, prims, tris)
        for i, albedo in enumerate([(0.8, 0.8, 0.8), (0.9, 0.1, 0.1), (0.1, 0.9, 0.1), (0.5, 0.5, 0.5)]):
            sc.c.materials[i].kind = abi.MAT_LAMBERT_SOLID; sc.c.materials[i].albedo[:] = albedo
        sc._bvh = host.attach_bvh(sc)
        for mode in (0, 1):
            opt = abi.Options.make(rng_mode=mode)
            gp, gl, gs = device.render(sc, cam, st, opt)
            op, ol, cnt = oracle_mod.render(sc, cam, st, opt)
            assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32)) and np.array_equal(gp, op) and gs.rays == cnt.rays, name
        means[name] = gl.mean(axis=(0, 1))
    assert np.allclose(means["flat leaf"], 0.5)                        # nothing is ever hit: pure miss colour
    assert not np.allclose(means["tilted leaf"], 0.5)
    # the green mesh shows in front of the (nearer!) red quad only when it is listed first; listed second it is culled
    assert means["mesh then quad"][1] > means["quad then mesh"][1] + 0.01
    assert means["quad then mesh"][0] > means["mesh then quad"][0] + 0.01


def _flat_leaf_case(abi, host):
    sc = _scene(abi, [_prim(abi, abi.PRIM_MESH, IDENT + IDENT)], [_tri((-1, -1, 0), (1, -1, 0), (0, 1, 0))])
    sc.c.materials[0].kind = abi.MAT_EMISSIVE; sc.c.materials[0].albedo[:] = (0.9, 0.1, 0.1)     # the triangle glows red
    sc._bvh = host.attach_bvh(sc)
    cam = abi.Camera(); cam.position[:] = [0, 0, 8]; cam.forward[:] = [0, 0, -1]; cam.right[:] = [1, 0, 0]; cam.true_up[:] = [0, 1, 0]
    cam.half_width, cam.half_height = 0.4, 0.3
    return sc, cam, abi.Settings(32, 24, 2, 4)


def test_fixed_aabb_flag_makes_flat_leaves_visible_in_the_oracle(native, oracle_mod, abi):
    """MI355RT_FLAG_FIXED_AABB (opt-in, not the reference): the slab test misses only on t_max < t_min, so the flat
    triangle of test_zero_thickness_boxes_never_hit is seen.  Without the flag the image is the miss colour everywhere."""
    host, _ = native
    sc, cam, st = _flat_leaf_case(abi, host)
    ref_p, ref_l, _ = oracle_mod.render(sc, cam, st, abi.Options.make())
    fix_p, fix_l, _ = oracle_mod.render(sc, cam, st, abi.Options.make(flags=abi.FLAG_FIXED_AABB))
    assert (ref_p == 0xB4B4B4).all()                                   # sqrt(0.5) * 255 = 180 = 0xB4: only the miss colour
    red = (fix_l[..., 0] > 0.8) & (fix_l[..., 1] < 0.2)
    assert 0.02 < red.mean() < 0.5 and (fix_p[~red & (fix_l[..., 1] > 0.45)] == 0xB4B4B4).all()
    again, _, _ = oracle_mod.render(sc, cam, st, abi.Options.make())    # the flag does not leak into the next call
    assert np.array_equal(again, ref_p)


@pytest.mark.gpu
def test_hip_fixed_aabb_matches_the_oracle(native, oracle_mod, abi):
    from conftest import load_for_both
    host, device = native
    sc, cam, st = _flat_leaf_case(abi, host)
    flag = abi.Options.make(flags=abi.FLAG_FIXED_AABB)
    gp, gl, gs = device.render(sc, cam, st, flag)
    op, ol, cnt = oracle_mod.render(sc, cam, st, flag)
    assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32)) and np.array_equal(gp, op) and gs.rays == cnt.rays
    assert not np.array_equal(gp, device.render(sc, cam, st, abi.Options.make())[0])
    for name in ("teapot", "semesterbild"):                              # real meshes: more triangles become visible, parity holds
        s2 = load_for_both(name, oracle_mod, host, width=72, height=54, spp=4, max_depth=10)
        gp, gl, gs = device.render(s2, s2.camera, s2.settings, flag)
        op, ol, cnt = oracle_mod.render(s2, s2.camera, s2.settings, flag)
        if name == "teapot":
            assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32)) and gs.rays == cnt.rays
        else:
            assert (np.sqrt(((gl.astype(np.float64) - ol) ** 2).sum(-1)) <= 1e-3).mean() >= 0.995
        assert not np.array_equal(gp, device.render(s2, s2.camera, s2.settings, abi.Options.make())[0]) or name == "teapot"
    with pytest.raises(RuntimeError, match="MI355RT_RNG_CTR"):
        device.render(sc, cam, st, abi.Options.make(rng_mode=abi.RNG_REF, flags=abi.FLAG_FIXED_AABB))
    with pytest.raises(RuntimeError, match="unknown bits"):
        device.render(sc, cam, st, abi.Options.make(flags=0x10))
