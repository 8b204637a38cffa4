"""Processing order (rt_api.cpp row_tables()): set_scene measures the cost of the image rows with a small probe render and the
persistent kernels then process the dearest rows first, the cheapest (sky) last, so that a launch does not end on its longest paths.  The image
must not change by a single bit -- draws are keyed by absolute row / x / sample and k_resolve writes every pixel where it belongs -- whatever
the selection, the banding or the chunking.  The mechanism is opt-in (diagnostic knob "row_order" = 1): measured in round 4, it moves full
frames by +-0.5 % (profiles/r04/ab_processing_order.txt), so the product processes rows in image order.  Reference semantics: rows are independent (src/renderer.rs:87-91)."""
import numpy as np
import pytest
import torch

from conftest import load_for_both

pytestmark = pytest.mark.gpu


def _render(device, sc, opt, order):
    device.set_knob("row_order", order)
    try:
        return device.render(sc, sc.camera, sc.settings, opt)
    finally:
        device.clear_knobs()


@pytest.mark.parametrize("name,W,H,spp,depth,opt_kw", [
    ("cornell", 64, 48, 6, 8, {}),
    ("semesterbild", 96, 64, 5, 12, {}),
    ("teapot", 64, 48, 4, 16, {"strip_rows": 3, "n_parts": 2, "part": 1}),                 # what one of two GPUs renders
    ("veach", 80, 45, 8, 8, {"row_begin": 5, "row_end": 39}),
    ("semesterbild", 64, 40, 6, 10, {"workspace_bytes": 64 * 7 * 6 * 12}),                  # 7 rows per band: six bands, each with its own shards
    ("semesterbild", 33, 35, 3, 6, {"strip_rows": 2, "n_parts": 3, "part": 0, "workspace_bytes": 33 * 4 * 3 * 12}),
])
def test_processing_order_does_not_change_the_image(name, W, H, spp, depth, opt_kw, native, oracle_mod, abi):
    host, device = native
    sc = load_for_both(name, oracle_mod, host, width=W, height=H, spp=spp, max_depth=depth)
    opt = abi.Options.make(**opt_kw)
    p0, l0, s0 = _render(device, sc, opt, 0)
    p1, l1, s1 = _render(device, sc, opt, 1)
    assert np.array_equal(p0, p1) and np.array_equal(l0.view(np.uint32), l1.view(np.uint32))
    assert (s0.samples, s0.rays, s0.rows_rendered) == (s1.samples, s1.rays, s1.rows_rendered)
    op, ol, cnt = oracle_mod.render(sc, sc.camera, sc.settings, opt)                          # and both are the oracle's image
    if name in ("cornell", "teapot"):                                                        # (scenes with a rough conductor agree with the oracle to 1e-3 only: libm)
        assert np.array_equal(l1.view(np.uint32), ol.view(np.uint32)) and np.array_equal(p1, op)


def test_dear_rows_first_sky_rows_last_in_every_group(native, oracle_mod, abi):
    host, device = native
    sc = load_for_both("semesterbild", oracle_mod, host, width=96, height=64, spp=4, max_depth=12)
    ctx = device.Context(0)
    try:
        ctx.set_knob("row_order", 1)
        ctx.set_scene(sc, sc.camera, sc.settings)
        out = torch.zeros((64, 96), dtype=torch.int32, device="cuda")
        ctx.render(out.data_ptr(), None, abi.Options.make(), None, want_stats=True)
        natural, processing, out_row, cost = ctx.row_tables()
        assert list(natural) == list(range(64)) and sorted(processing) == list(range(64)) and sorted(out_row) == list(range(64))
        assert all(natural[out_row[j]] == processing[j] for j in range(64))
        assert cost.shape == (64,) and cost.min() == 1.0 and cost.max() > 1.5              # sky rows: exactly one ray per path
        groups = 8                                                                           # one band: the kernel's 8 work shards
        per = [processing[g * 8:(g + 1) * 8] for g in range(groups)]
        for rows in per:
            c = cost[rows]
            assert all(c[i] >= c[i + 1] for i in range(len(c) - 1))                          # dearest first inside a group
        assert abs(sum(cost[per[0]]) - sum(cost[per[-1]])) <= cost.max()                     # and the groups cost about the same
        # the last row of every group is among the cheapest eighth of the image
        assert all(cost[rows[-1]] <= np.sort(cost)[8] for rows in per)
        # the same context, another selection: new tables, same picture
        opt = abi.Options.make(strip_rows=4, n_parts=2, part=1)
        st = ctx.render(out.data_ptr(), None, opt, None, want_stats=True)
        nat2, proc2, out2, _ = ctx.row_tables()
        assert list(nat2) == abi.rows_selected(64, opt) and sorted(proc2) == sorted(nat2) and st.rows_rendered == 32
    finally:
        ctx.close()
    device.set_knob("row_order", 0)
    try:
        want = device.render(sc, sc.camera, sc.settings, abi.Options.make(strip_rows=4, n_parts=2, part=1))[0]
    finally:
        device.clear_knobs()
    assert np.array_equal(out[:32].cpu().numpy().astype(np.uint32), want)


def test_progressive_chunks_and_the_reference_stream_under_the_order(native, oracle_mod, abi):
    host, device = native
    sc = load_for_both("semesterbild", oracle_mod, host, width=48, height=40, spp=8, max_depth=8)
    want_p, want_l, _ = _render(device, sc, abi.Options.make(), 0)
    ctx = device.Context(0)
    try:
        ctx.set_knob("row_order", 1)
        ctx.set_scene(sc, sc.camera, sc.settings)
        packed = torch.zeros((40, 48), dtype=torch.int32, device="cuda")
        linear = torch.zeros((40, 48, 3), dtype=torch.float32, device="cuda")
        accum = torch.zeros((40, 48, 4), dtype=torch.float32, device="cuda")
        for s0, s1 in ((0, 3), (3, 4), (4, 8)):                                              # the sample loop in chunks (renderer.rs:93-101)
            ctx.render_progressive(s0, s1, accum.data_ptr(), packed.data_ptr(), linear.data_ptr(), abi.Options.make())
        ctx.check()
        assert np.array_equal(packed.cpu().numpy().astype(np.uint32), want_p)
        assert np.array_equal(linear.cpu().numpy().view(np.uint32), want_l.view(np.uint32))
        # the replay of the reference's row streams ignores the order (a row's stream is sequential over its pixels)
        ref = abi.Options.make(rng_mode=abi.RNG_REF)
        ctx.render(packed.data_ptr(), linear.data_ptr(), ref, None, want_stats=True)
        op, ol, _ = oracle_mod.render(sc, sc.camera, sc.settings, ref)
        assert np.array_equal(packed.cpu().numpy().astype(np.uint32), op)
    finally:
        ctx.close()
