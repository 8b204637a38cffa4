"""The C ABI must reject malformed input with an error code -- never launch a kernel on it (a faulting kernel
can reset the GPU).  These run on the GPU box because validation lives behind context creation."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_for_both

pytestmark = pytest.mark.gpu


def _render_rc(device, abi, sc, cam, st, opt=None):
    packed = np.zeros((st.height, st.width), np.uint32)
    rc = device.lib().mi355rt_render(C.byref(getattr(sc, "c", sc)), C.byref(cam), C.byref(st), C.byref(opt) if opt is not None else None,
                                     packed.ctypes.data, None, None)
    return rc, device.lib().mi355rt_last_error().decode()


def test_bad_indices_and_kinds_are_rejected(native, oracle_mod, abi):
    host, device = native
    sc = load_for_both("semesterbild", oracle_mod, host, width=16, height=12, spp=1, max_depth=3)
    st, cam = sc.settings, sc.camera
    assert _render_rc(device, abi, sc, cam, st)[0] == 0

    def restore(obj, field, value):
        setattr(obj, field, value)

    p0 = sc.c.primitives[0]
    old = p0.material; p0.material = 999
    rc, msg = _render_rc(device, abi, sc, cam, st); restore(p0, "material", old)
    assert rc == abi.ERR_INVALID and "material" in msg
    old = p0.kind; p0.kind = 17
    rc, msg = _render_rc(device, abi, sc, cam, st); restore(p0, "kind", old)
    assert rc == abi.ERR_INVALID and "kind" in msg
    m0 = sc.c.materials[0]
    old = m0.kind; m0.kind = 99
    rc, msg = _render_rc(device, abi, sc, cam, st); restore(m0, "kind", old)
    assert rc == abi.ERR_INVALID
    mesh = sc.c.meshes[0]
    old = mesh.node_count; mesh.node_count = sc.c.n_nodes + 5
    rc, msg = _render_rc(device, abi, sc, cam, st); restore(mesh, "node_count", old)
    assert rc == abi.ERR_INVALID and "node range" in msg
    # BVH with a cycle: root's left child points back at the root
    root = sc.c.nodes[0]
    old = root.left; root.left = 0
    rc, msg = _render_rc(device, abi, sc, cam, st); restore(root, "left", old)
    assert rc == abi.ERR_INVALID and ("cycle" in msg or "malformed" in msg)
    # a leaf that references a triangle outside the mesh
    leaf_idx = next(i for i in range(sc.c.n_nodes) if sc.c.nodes[i].index_count > 0)
    k = sc.c.nodes[leaf_idx].first_index
    old = sc.c.tri_indices[k]; sc.c.tri_indices[k] = 10 ** 6
    rc, msg = _render_rc(device, abi, sc, cam, st); sc.c.tri_indices[k] = old
    assert rc == abi.ERR_INVALID and "triangle id" in msg
    # still renders after all that
    assert _render_rc(device, abi, sc, cam, st)[0] == 0


def test_settings_and_options_are_validated(native, oracle_mod, abi):
    host, device = native
    sc = load_for_both("cornell", oracle_mod, host, width=8, height=6, spp=2, max_depth=3)
    cam = sc.camera
    for bad in (abi.Settings(0, 6, 2, 3), abi.Settings(8, 0, 2, 3), abi.Settings(8, 6, 0, 3), abi.Settings(1 << 24, 6, 2, 3)):
        assert _render_rc(device, abi, sc, cam, bad)[0] == abi.ERR_INVALID
    st = sc.settings
    o = abi.Options.make(); o.abi_version = 7
    assert _render_rc(device, abi, sc, cam, st, o)[0] == abi.ERR_INVALID
    o = abi.Options.make(rng_mode=5)
    assert _render_rc(device, abi, sc, cam, st, o)[0] == abi.ERR_INVALID
    o = abi.Options.make(row_begin=4, row_end=2)
    assert _render_rc(device, abi, sc, cam, st, o)[0] == abi.ERR_INVALID
    o = abi.Options.make(workspace_bytes=8)              # less than one pixel's spp * 12 bytes
    rc, msg = _render_rc(device, abi, sc, cam, st, o)
    assert rc == abi.ERR_INVALID and "workspace" in msg
    sky = abi.Scene(); C.memmove(C.byref(sky), C.byref(sc.c), C.sizeof(abi.Scene)); sky.sky_width = 4; sky.sky_height = 2
    rc, msg = _render_rc(device, abi, sky, cam, st)              # dimensions without pixels
    assert rc == abi.ERR_INVALID and "sky" in msg
    # max_depth 0: every path is BLACK without tracing (renderer.rs:20-22)
    st0 = abi.Settings(8, 6, 2, 0)
    packed, lin, stats = device.render(sc, cam, st0, abi.Options.make())
    assert np.all(packed == 0) and stats.rays == 0 and stats.samples == 8 * 6 * 2
    # an options selection with no rows is a no-op
    o = abi.Options.make(row_begin=3, row_end=3)
    assert _render_rc(device, abi, sc, cam, st, o)[0] == 0


def test_context_reuse_and_scene_switch(native, oracle_mod, abi):
    """One context, two scenes, many renders: results do not depend on what ran before."""
    import ctypes
    host, device = native
    a = load_for_both("cornell", oracle_mod, host, width=32, height=24, spp=3, max_depth=5)
    b = load_for_both("teapot", oracle_mod, host, width=32, height=24, spp=3, max_depth=5)
    want_a = device.render(a, a.camera, a.settings, abi.Options.make())[0]
    want_b = device.render(b, b.camera, b.settings, abi.Options.make())[0]
    hip = ctypes.CDLL("libamdhip64.so")
    n = 32 * 24 * 4
    d = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(d), n) == 0
    ctx = device.Context(0)
    out = np.zeros((24, 32), np.uint32)
    for sc, want in ((a, want_a), (b, want_b), (a, want_a)):
        ctx.set_scene(sc, sc.camera, sc.settings)
        for _ in range(2):
            ctx.render(d.value, None, abi.Options.make(), None, want_stats=True)
            assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), d, n, 2) == 0
            assert np.array_equal(out, want)
    ctx.close(); hip.hipFree(d)


def test_a_scaled_quad_normal_is_refused(native, oracle_mod, abi):
    """The quad test divides by dot(normal, direction) with a division proven equal to `/` for divisors up to 2^25 (rt_math.h div_bounded); the
    reference's constructor stores a UNIT normal (tungsten/objects/quad.rs:26-79).  A caller that hands the C ABI a scaled or non-finite normal
    is outside both: refused at upload, never rendered with arithmetic that could differ from the reference's."""
    host, device = native
    sc = load_for_both("cornell", oracle_mod, host, width=16, height=12, spp=1, max_depth=3)
    quad = next(sc.c.primitives[i] for i in range(sc.c.n_primitives) if sc.c.primitives[i].kind == abi.PRIM_QUAD)
    old = quad.data[10]
    for bad in (3.0e7, float("inf"), float("nan")):
        quad.data[10] = bad
        rc, msg = _render_rc(device, abi, sc, sc.camera, sc.settings)
        assert rc == abi.ERR_INVALID and "quad normal" in msg, (bad, rc, msg)
    quad.data[10] = old
    assert _render_rc(device, abi, sc, sc.camera, sc.settings)[0] == 0


@pytest.mark.gpu
def test_degenerate_renders_go_to_the_plain_loop(native, oracle_mod, abi):
    """The mesh-free lockstep kernels are compiled for lists that hold something and for paths that may take a step (rt_kernels.hip render_ctr_lockstep:
    __builtin_assume); rt_api.cpp sends the two degenerate renders -- an empty list, max_depth == 0 -- to the plain per-lane loop.  Every sample is the
    miss colour / BLACK, with the oracle's ray counts, whichever kernel the scene would otherwise pick."""
    from fuzz_scenes import random_scene
    host, device = native
    empty = random_scene(abi, host, 7, True, n_prims=0, only_kinds=[abi.PRIM_QUAD])
    st = abi.Settings(24, 18, 5, 6)
    for mode in (0, 1):
        opt = abi.Options.make(rng_mode=mode, seed=11 if mode == 0 else 0)
        gp, gl, gs = device.render(empty, empty.camera, st, opt)
        op, ol, cnt = oracle_mod.render(empty, empty.camera, st, opt)
        assert np.array_equal(gp, op) and np.array_equal(gl.view(np.uint32), ol.view(np.uint32)) and gs.rays == cnt.rays == 24 * 18 * 5
    for kw in (dict(n_prims=9, only_kinds=[2, 3], lambert_only=True), dict(n_prims=9, only_kinds=[2, 3, 0, 1])):   # k_render_ctr_simple; the general mesh-free kernels
        sc = random_scene(abi, host, 8, True, **kw)
        st0 = abi.Settings(24, 18, 5, 0)
        gp, gl, gs = device.render(sc, sc.camera, st0, abi.Options.make())
        op, ol, cnt = oracle_mod.render(sc, sc.camera, st0, abi.Options.make())
        assert np.all(gp == 0) and np.array_equal(gp, op) and gs.rays == cnt.rays == 0 and gs.samples == 24 * 18 * 5
        st1 = abi.Settings(24, 18, 5, 1)                                                                          # ... and one step: the assuming kernels
        gp, gl, gs = device.render(sc, sc.camera, st1, abi.Options.make())
        op, ol, cnt = oracle_mod.render(sc, sc.camera, st1, abi.Options.make())
        assert np.array_equal(gl.view(np.uint32), ol.view(np.uint32)) and gs.rays == cnt.rays
