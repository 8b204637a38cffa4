"""Progressive rendering (src/renderer.rs:93-101 cut into sample chunks): any chunking of 0..N must reproduce the
one-shot N-spp image bit-for-bit, and every intermediate image must equal the one-shot image of that many samples."""
import numpy as np
import pytest

from conftest import load_for_both

pytestmark = pytest.mark.gpu


def _one_shot(device, abi, sc, spp, opt):
    st = abi.Settings(sc.settings.width, sc.settings.height, spp, sc.settings.max_depth)
    return device.render(sc, sc.camera, st, opt)


@pytest.mark.parametrize("name,chunks,opt_kw", [
    ("cornell", (3, 5, 8), {}),
    ("cornell", (1, 1, 30), {"strip_rows": 3, "n_parts": 2, "part": 1}),
    ("teapot", (4, 2, 6), {"workspace_bytes": 64 * 48 * 12 * 3}),          # several pixel bands per chunk
    ("veach", (7, 9), {"row_begin": 5, "row_end": 29}),
])
def test_chunked_samples_equal_one_shot(name, chunks, opt_kw, native, oracle_mod, abi):
    import torch
    host, device = native
    sc = load_for_both(name, oracle_mod, host, width=64, height=48, spp=sum(chunks), max_depth=8)
    opt = abi.Options.make(**opt_kw)
    rows = len(abi.rows_selected(48, opt))
    ctx = device.Context(0)
    ctx.set_scene(sc, sc.camera, sc.settings)
    accum = torch.full((rows * 64, 4), float("nan"), dtype=torch.float32, device="cuda")      # never read before the first chunk wrote it
    packed = torch.zeros(rows * 64, dtype=torch.int32, device="cuda")
    linear = torch.zeros(rows * 64 * 3, dtype=torch.float32, device="cuda")
    s = 0
    for c in chunks:
        st = ctx.render_progressive(s, s + c, accum.data_ptr(), packed.data_ptr(), linear.data_ptr(), opt, want_stats=True)
        s += c
        assert st.samples == rows * 64 * c
        gp, gl, _ = _one_shot(device, abi, sc, s, opt)
        assert np.array_equal(packed.cpu().numpy().view(np.uint32).reshape(gp.shape), gp), f"packed image after {s} samples"
        assert np.array_equal(linear.cpu().numpy().view(np.uint32), gl.reshape(-1).view(np.uint32)), f"linear image after {s} samples"
    op, ol, _ = oracle_mod.render(sc, sc.camera, sc.settings, opt)
    if name != "veach":                                                    # transcendental-free scenes: exact against the oracle too
        assert np.array_equal(linear.cpu().numpy().view(np.uint32), ol.reshape(-1).view(np.uint32))
    ctx.close()


def test_progressive_argument_checks(native, oracle_mod, abi):
    import torch
    host, device = native
    sc = load_for_both("cornell", oracle_mod, host, width=16, height=8, spp=4, max_depth=3)
    ctx = device.Context(0)
    ctx.set_scene(sc, sc.camera, sc.settings)
    accum = torch.zeros((16 * 8, 4), dtype=torch.float32, device="cuda")
    packed = torch.zeros(16 * 8, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="sample_end"):
        ctx.render_progressive(4, 4, accum.data_ptr(), packed.data_ptr())
    with pytest.raises(RuntimeError, match="d_accum"):
        ctx.render_progressive(0, 4, 0, packed.data_ptr())
    with pytest.raises(RuntimeError, match="MI355RT_RNG_CTR"):
        ctx.render_progressive(0, 2, accum.data_ptr(), packed.data_ptr(), options=abi.Options.make(rng_mode=abi.RNG_REF))
    ctx.close()
