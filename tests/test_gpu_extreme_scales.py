"""The short reciprocal / division of rt_math.h are exact only on stated argument ranges; outside them a ballot sends the whole wave to the
compiler's division (recip3: a non-zero magnitude below 2^-126 or one of 2^126 or more; hit_quad: a numerator of 2^100 or more).  Ordinary
scenes never get there, so these scenes do: geometry at 2^100 .. 2^105 next to ordinary geometry, through every kernel family, bit for
bit against the oracle (plain IEEE arithmetic on the CPU).  Reference arithmetic: quad.rs:83-132, cube.rs:59-158."""
import numpy as np
import pytest


def _scene(abi, host, huge_cube, huge_quad, with_mesh):
    from oracle import scene_loader as L
    F = np.float32
    mats = []

    def mat(kind, albedo=(0, 0, 0), p0=0.0):
        m = abi.Material(); m.kind = kind; m.albedo[:] = [float(F(v)) for v in albedo]; m.p0 = float(F(p0)); mats.append(m); return len(mats) - 1

    grey, red, light = mat(abi.MAT_LAMBERT_SOLID, (0.7, 0.7, 0.7)), mat(abi.MAT_LAMBERT_SOLID, (0.8, 0.2, 0.1)), mat(abi.MAT_EMISSIVE, (4, 4, 4))
    prims = []

    def quad(scale, translate, material, euler=(0.0, 0.0, 0.0)):
        m = L.mat4_from_scale_rotation_translation([F(v) for v in scale], L.quat_from_euler_yxz_deg(*[F(v) for v in euler]), [F(v) for v in translate])
        p = abi.Primitive(); p.kind = abi.PRIM_QUAD; p.material = material; p.data[0:15] = [float(v) for v in L.quad_from_matrix(m)]; prims.append(p)

    def cube(scale, translate, material, euler=(20.0, 30.0, 10.0)):
        m = L.mat4_from_scale_rotation_translation([F(v) for v in scale], L.quat_from_euler_yxz_deg(*[F(v) for v in euler]), [F(v) for v in translate])
        p = abi.Primitive(); p.kind = abi.PRIM_CUBE; p.material = material
        p.data[0:16] = [float(v) for v in m]; p.data[16:32] = [float(v) for v in L.mat4_inverse(m)]; prims.append(p)

    quad((8, 1, 8), (0, -1, 0), grey)                                   # an ordinary floor ...
    cube((1.5, 1.5, 1.5), (0.5, 0.0, -1.0), red)                        # ... an ordinary cube ...
    quad((3, 1, 3), (0, 4, 0), light, euler=(0.0, 180.0, 0.0))          # ... and a light above them
    if huge_quad:
        quad((2.0 ** 40, 1, 2.0 ** 40), (0, -(2.0 ** 105), 0), grey)     # plane constant 2^105: |numerator| >= 2^100 for every ray (its parallelogram test overflows: a miss, on both sides)
    if huge_cube:
        cube((2.0 ** 115, 40.0, 40.0), (0, 0, 0), grey, euler=(0.0, 0.0, 0.0))   # contains everything; w2o = diag(2^-115, 1/40, 1/40): the object-space
                                                                        # direction's x is a denormal for every ray with |d.x| < 2^-11 (a few dozen of the 85 000)
    meshes, tris = [], np.zeros((0, 12), F)
    if with_mesh:
        p = abi.Primitive(); p.kind = abi.PRIM_MESH; p.material = grey
        m = L.mat4_from_scale_rotation_translation([F(1), F(1), F(1)], L.quat_from_euler_yxz_deg(F(0), F(0), F(0)), [F(-2), F(0), F(0)])
        p.data[0:16] = [float(v) for v in m]; p.data[16:32] = [float(v) for v in L.mat4_inverse(m)]
        v = np.random.default_rng(5).uniform(-1, 1, size=(40, 3, 3)).astype(F)
        tris = L._triangles_from_indexed(v.reshape(-1, 3), np.arange(120).reshape(40, 3))
        mesh = abi.Mesh(); mesh.first_triangle, mesh.triangle_count = 0, len(tris); meshes.append(mesh); p.mesh = 0; prims.append(p)
    sc = L.LoadedScene(); sc.materials, sc.primitives, sc.meshes, sc.triangles = mats, prims, meshes, tris
    sc.finalize(); sc.c.miss_color[:] = [0.5, 0.6, 0.7]
    sc._keep = host.attach_bvh(sc)
    sc.camera = L.camera_new((0.0, 1.0, 6.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), F(55.0), F(4.0 / 3.0))
    return sc


@pytest.mark.gpu
@pytest.mark.parametrize("huge_cube,huge_quad,with_mesh", [(True, True, False), (True, False, False), (False, True, False), (True, True, True)])
def test_geometry_at_two_to_the_hundred(huge_cube, huge_quad, with_mesh, native, oracle_mod, abi):
    host, device = native
    sc = _scene(abi, host, huge_cube, huge_quad, with_mesh)
    st = abi.Settings(96, 72, 8, 6)
    for mode, seed in ((abi.RNG_CTR, 11), (abi.RNG_CTR, 12345), (abi.RNG_REF, 0)):
        opt = abi.Options.make(rng_mode=mode, seed=seed)
        gp, gl, gs = device.render(sc, sc.camera, st, opt)
        op, ol, cnt = oracle_mod.render(sc, sc.camera, st, opt)
        assert gs.rays == cnt.rays
        assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32)) and np.array_equal(gp, op)
    if not with_mesh:                                                     # the general mesh-free kernels see the same scenes (knob `kernel`: 0 general lockstep, 11 mesh-free wavefront)
        op, ol, cnt = oracle_mod.render(sc, sc.camera, st, abi.Options.make(rng_mode=abi.RNG_CTR, seed=11))
        for k in (0, 9, 11):
            device.set_knob("kernel", k)
            try:
                gp, gl, gs = device.render(sc, sc.camera, st, abi.Options.make(rng_mode=abi.RNG_CTR, seed=11))
            finally:
                device.clear_knobs()
            assert gs.rays == cnt.rays and np.array_equal(gl.view(np.uint32), ol.view(np.uint32)), k
