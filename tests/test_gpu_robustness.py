"""Robustness of the resident-scene API on a real GPU (ADVICE rt_api.cpp:347):
  * a radiance workspace that does not fit is not an error: the band is halved until hipMalloc succeeds, and the banded image
    is bit-identical to the one-band image;
  * one context driven from two HIP streams: the second render waits for the first (the workspaces belong to one render at a
    time), so both images are exactly what each call gives on its own.
"""
import numpy as np
import pytest
import torch

from conftest import SCENES

pytestmark = pytest.mark.gpu


def test_out_of_memory_for_the_workspace_falls_back_to_smaller_bands(native, abi):
    host, device = native
    W, H, spp = 800, 600, 1024                                       # 491.5 M samples -> a 5.9 GB one-band workspace (12 B per sample)
    sc = host.LoadedScene(SCENES["cornell"], W, H, spp, 8)
    n = W * H
    ref = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx = device.Context(0)
    ctx.set_scene(sc, sc.camera, sc.settings)
    st0 = ctx.render(ref.data_ptr(), None, abi.Options.make(), None, want_stats=True)
    ctx.close()
    assert st0.bands == 1
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free, _total = torch.cuda.mem_get_info()
    hog = torch.empty(max(free - (2 << 30), 0), dtype=torch.uint8, device="cuda")      # leave about 2 GB
    try:
        out = torch.zeros(n, dtype=torch.int32, device="cuda")
        ctx = device.Context(0)
        ctx.set_scene(sc, sc.camera, sc.settings)
        st = ctx.render(out.data_ptr(), None, abi.Options.make(), None, want_stats=True)
        ctx.close()
        assert st.bands >= 4 and st.samples == st0.samples and st.rays == st0.rays       # 5.9 -> 2.95 -> 1.47 GB
        assert torch.equal(out, ref)
    finally:
        del hog
        torch.cuda.empty_cache()


def test_one_context_on_two_streams_is_serialised(native, abi):
    host, device = native
    sc = host.LoadedScene(SCENES["cornell"], 400, 300, 64, 8)
    n = 400 * 300
    a_opt, b_opt = abi.Options.make(strip_rows=3, n_parts=2, part=0), abi.Options.make(strip_rows=3, n_parts=2, part=1)
    ctx = device.Context(0)
    ctx.set_scene(sc, sc.camera, sc.settings)
    want_a = torch.zeros(n // 2, dtype=torch.int32, device="cuda"); want_b = torch.zeros(n // 2, dtype=torch.int32, device="cuda")
    ctx.render(want_a.data_ptr(), None, a_opt, None); torch.cuda.synchronize()
    ctx.render(want_b.data_ptr(), None, b_opt, None); torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(4):                                               # back to back on alternating streams, no host sync in between
        got_a = torch.zeros(n // 2, dtype=torch.int32, device="cuda"); got_b = torch.zeros(n // 2, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.render(got_a.data_ptr(), None, a_opt, s1.cuda_stream)
        ctx.render(got_b.data_ptr(), None, b_opt, s2.cuda_stream)
        ctx.render(got_a.data_ptr(), None, a_opt, s2.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(got_a, want_a) and torch.equal(got_b, want_b)
    ctx.close()


def test_a_wave_that_gives_up_is_reported_on_the_asynchronous_path(native, abi):
    """render_scene is infallible (/root/reference/src/renderer.rs:67): the boundary must never hand back a partial image as OK.
    The wavefront kernel bounds every wait; the diagnostic knob `spin_idle` = 1 (through RenderParams, same product library)
    makes a wave that finds its queues empty twice in a row give up, which leaves paths unfinished.  The render was enqueued with
    stats == NULL, so the call itself returned OK -- the failure must surface, exactly once, in mi355rt_context_check, in
    mi355rt_context_read_timing, at the next render on the context, and in the synchronous (stats) form; afterwards the
    context works again, and nothing hangs or aborts."""
    host, device = native
    sc = host.LoadedScene(SCENES["semesterbild"], 160, 120, 16, 30)
    n = 160 * 120
    out = torch.zeros(n, dtype=torch.int32, device="cuda")

    def fresh(spin):
        c = device.Context(0)
        if spin:
            c.set_knob("spin_idle", spin)
        c.set_scene(sc, sc.camera, sc.settings)
        assert c.kernel_variant() in (7, 10, 12, 13)                    # a wavefront kernel (10 / 12 / 13: the instantiations without the metal branch)
        return c

    good = fresh(0)
    want = torch.zeros(n, dtype=torch.int32, device="cuda")
    st = good.render(want.data_ptr(), None, abi.Options.make(), None, want_stats=True)
    good.check()                                                       # a healthy context: nothing to report
    good.close()

    # (1) asynchronous render, then the explicit check
    ctx = fresh(1)
    ctx.render(out.data_ptr(), None, abi.Options.make(), None)         # stats == NULL: enqueue only, returns OK
    torch.cuda.synchronize()
    with pytest.raises(device.RenderError, match="watchdog") as e:
        ctx.check()
    assert e.value.rc == abi.ERR_HIP
    # the message explains itself: which kernel, which of its bounded waits gave up, under which limits, and that it was an EARLIER render
    msg = str(e.value)
    assert "in an earlier render on this context" in msg and "k_render_ctr_wf" in msg and f"(variant {ctx.kernel_variant()})" in msg
    assert "idle: no progress in the workgroup" in msg and "limits: 1 idle polls" in msg
    ctx.check()                                                        # reported once
    # (2) ... the next render on the context reports the previous one
    ctx.render(out.data_ptr(), None, abi.Options.make(), None)
    torch.cuda.synchronize()
    with pytest.raises(device.RenderError, match="watchdog"):
        ctx.render(out.data_ptr(), None, abi.Options.make(), None)
    # (3) ... read_timing after a timed asynchronous render
    ctx.set_timing(True)
    ctx.render(out.data_ptr(), None, abi.Options.make(), None)
    torch.cuda.synchronize()
    with pytest.raises(device.RenderError, match="watchdog"):
        ctx.read_timing()
    ctx.set_timing(False)
    # (4) ... and the synchronous form reports its own render
    with pytest.raises(device.RenderError, match="watchdog") as e4:
        ctx.render(out.data_ptr(), None, abi.Options.make(), None, want_stats=True)
    assert "in this render" in str(e4.value) and "earlier" not in str(e4.value)
    ctx.close()
    # the one-shot call (what src/main.rs:57 would bind) is synchronous: same error code, no image handed over as OK
    device.set_knob("spin_idle", 1)
    try:
        with pytest.raises(device.RenderError, match="watchdog"):
            device.render(sc, sc.camera, sc.settings, abi.Options.make())
    finally:
        device.clear_knobs()
    # a context with the product's limits renders the same image as before: the device is fine
    again = fresh(0)
    st2 = again.render(out.data_ptr(), None, abi.Options.make(), None, want_stats=True)
    again.check()
    again.close()
    assert torch.equal(out, want) and st2.rays == st.rays
