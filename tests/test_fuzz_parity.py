"""Random scenes covering every primitive and material kind: the oracle must agree with itself across fold orders
(CPU), and the HIP path must agree with the oracle (GPU) -- bit-for-bit when no transcendental is involved."""
import numpy as np
import pytest

from fuzz_scenes import random_scene


def _kinds_present(sc):
    return {sc.c.primitives[i].kind for i in range(sc.c.n_primitives)}, {sc.c.materials[i].kind for i in range(sc.c.n_materials)}


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_scene_covers_everything_and_oracle_is_consistent(seed, native, oracle_mod, abi):
    host, _ = native
    sc = random_scene(abi, host, seed, exact_only=False)
    pk, mk = _kinds_present(sc)
    assert pk == {0, 1, 2, 3, 4} and mk == set(range(9))
    st = abi.Settings(48, 36, 4, 6)
    a = oracle_mod.render(sc, sc.camera, st, abi.Options.make(rng_mode=abi.RNG_CTR), fold=0)[1]     # tail-first fold
    b = oracle_mod.render(sc, sc.camera, st, abi.Options.make(rng_mode=abi.RNG_CTR), fold=1)[1]     # forward fold
    assert np.allclose(a, b, rtol=2e-6, atol=1e-7)                 # same paths, products associated differently
    assert np.isfinite(a).all() and a.max() > 0.05 and (a == 0).mean() < 0.9


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("mode", [0, 1])
def test_hip_matches_oracle_on_exact_fuzz_scenes(seed, mode, native, oracle_mod, abi):
    host, device = native
    sc = random_scene(abi, host, seed, exact_only=True)
    st = abi.Settings(64, 48, 6, 8)
    opt = abi.Options.make(rng_mode=mode)
    gp, gl, stats = device.render(sc, sc.camera, st, opt)
    op, ol, cnt = oracle_mod.render(sc, sc.camera, st, opt)
    assert stats.rays == cnt.rays
    assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32)), f"{(gl != ol).any(-1).sum()} px differ, max {np.abs(gl - ol).max()}"
    assert np.array_equal(gp, op)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_hip_matches_oracle_on_fuzz_scenes_with_rough_conductors(seed, native, oracle_mod, abi):
    host, device = native
    sc = random_scene(abi, host, seed, exact_only=False)
    st = abi.Settings(64, 48, 6, 8)
    opt = abi.Options.make()
    gp, gl, stats = device.render(sc, sc.camera, st, opt)
    op, ol, cnt = oracle_mod.render(sc, sc.camera, st, opt)
    l2 = np.sqrt(((gl.astype(np.float64) - ol) ** 2).sum(-1))
    assert (l2 <= 1e-3).mean() >= 0.995 and (gp == op).mean() >= 0.99
    assert abs(stats.rays - cnt.rays) <= 0.001 * cnt.rays


@pytest.mark.gpu
@pytest.mark.parametrize("label,kw", [
    ("meshes only", dict(n_prims=6, only_kinds=[4], mesh_tris=200)),
    ("tiny meshes", dict(n_prims=9, only_kinds=[4, 2, 4], mesh_tris=0)),
    ("mesh first and last", dict(n_prims=7, only_kinds=[4, 0, 3, 2, 1, 3, 4], mesh_tris=40)),
    ("long list", dict(n_prims=70, mesh_tris=25)),
])
def test_state_machine_corner_shapes(label, kw, native, oracle_mod, abi):
    host, device = native
    sc = random_scene(abi, host, 21, exact_only=True, **kw)
    st = abi.Settings(48, 36, 5, 7)
    for mode in (0, 1):
        opt = abi.Options.make(rng_mode=mode)
        gp, gl, stats = device.render(sc, sc.camera, st, opt)
        op, ol, cnt = oracle_mod.render(sc, sc.camera, st, opt)
        assert stats.rays == cnt.rays, label
        assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32)) and np.array_equal(gp, op), label


@pytest.mark.gpu
@pytest.mark.parametrize("label,kw,kernel", [
    ("every mesh-free kind, every exact material", dict(n_prims=16, only_kinds=[2, 2, 3, 0, 1, 3, 3, 0, 0, 2, 1, 1]), 0),
    ("alternating kinds (runs of one)", dict(n_prims=12, only_kinds=[3, 2, 0, 1]), 0),
    ("one long run", dict(n_prims=40, only_kinds=[2]), 0),
    ("Lambert-only list", dict(n_prims=14, only_kinds=[2, 2, 2, 3, 3, 2, 0, 1, 3], lambert_only=True), 3),
    ("Lambert-only cubes", dict(n_prims=9, only_kinds=[3], lambert_only=True), 14),                         # quads and cubes only: the instantiation pruned to those two kinds
    ("Lambert-only quads and cubes", dict(n_prims=13, only_kinds=[2, 3, 3, 2, 2], lambert_only=True), 14),
])
@pytest.mark.parametrize("seed", [31, 32, 33])
def test_lockstep_kernels_on_mesh_free_fuzz_scenes(label, kw, kernel, seed, native, oracle_mod, abi):
    """Mesh-free lists go to the lockstep kernels, whose list walk loops over RUNS of equal kinds (DevPrim.run_end) and, in the
    Lambert-only kernel, carries the cube's object-space hit point in the candidate: random lists with runs of every length and
    order, both RNG modes, must match the oracle bit for bit -- and must really have run on the kernel the case is meant for."""
    host, device = native
    sc = random_scene(abi, host, seed, exact_only=True, **kw)
    st = abi.Settings(56, 40, 5, 9)
    ctx = device.Context(0)
    try:
        ctx.set_scene(sc, sc.camera, st)
        assert ctx.kernel_variant() == kernel, label
    finally:
        ctx.close()
    for mode in (0, 1):
        opt = abi.Options.make(rng_mode=mode, seed=seed * 104729 if mode == 0 else 0)
        gp, gl, stats = device.render(sc, sc.camera, st, opt)
        op, ol, cnt = oracle_mod.render(sc, sc.camera, st, opt)
        assert stats.rays == cnt.rays, label
        assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32)) and np.array_equal(gp, op), label


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(40, 52)))
def test_untransformed_mesh_kernel_on_fuzz_scenes(seed, native, oracle_mod, abi):
    """k_render_ctr_wf_nometal_ident (variant 12) is picked for lists whose meshes are all untransformed and that hold no metal.  Random lists of
    that shape -- meshes next to every other primitive kind, every other material -- with the fuzz scenes' camera ON the plane x = 0 (so that every
    primary ray has a zero component and takes the general form of mesh_setup, the bounced rays the short one) must match the oracle bit for bit and
    must really have run on that kernel; the general instantiation (10) forced on the same scene gives the same image (mesh_object.rs:264-291)."""
    host, device = native
    shapes = [dict(n_prims=9, mesh_tris=80), dict(n_prims=5, only_kinds=[4, 4, 2], mesh_tris=300), dict(n_prims=14, only_kinds=[4, 3, 0, 4, 2, 1], mesh_tris=30)]
    sc = random_scene(abi, host, seed, exact_only=True, identity_meshes=True, no_metal=True, **shapes[seed % 3])
    st = abi.Settings(56 + seed % 9, 40 + seed % 7, 4 + seed % 3, 3 + seed % 9)
    opt = abi.Options.make()
    op, ol, cnt = oracle_mod.render(sc, sc.camera, st, opt)
    for forced in (None, 10):
        ctx = device.Context(0)
        try:
            if forced is not None:
                ctx.set_knob("kernel", forced)
            ctx.set_scene(sc, sc.camera, st)
            assert ctx.kernel_variant() == (12 if forced is None else forced)
        finally:
            ctx.close()
        if forced is not None:
            device.set_knob("kernel", forced)
        try:
            gp, gl, stats = device.render(sc, sc.camera, st, opt)
        finally:
            device.clear_knobs()
        assert stats.rays == cnt.rays, (seed, forced)
        assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32)) and np.array_equal(gp, op), (seed, forced)
