"""mi355rt_render_multi: ONE host process drives several devices (row strips dealt round-robin, device-to-host copies
into the caller's image, no collective).  On a one-GPU box the same device is listed several times: every part still
has its own context, buffers and host thread, so the plumbing is the real one."""
import numpy as np
import pytest

from conftest import load_for_both

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,H,devices,opt_kw", [
    ("cornell", 48, [0, 0], {"strip_rows": 4}),
    ("cornell", 50, [0, 0, 0], {"strip_rows": 0}),                       # 0 -> strips of 4; 50 rows do not divide evenly
    ("teapot", 37, [0, 0, 0, 0, 0, 0, 0, 0], {"strip_rows": 1}),         # 8 parts like a full node
    ("cornell", 48, [0, 0], {"strip_rows": 5, "row_begin": 7, "row_end": 41}),
    ("cornell", 3, [0, 0, 0, 0], {"strip_rows": 2}),                     # more devices than strips: some parts are empty
])
def test_one_process_many_devices_equals_one_device(name, H, devices, opt_kw, native, oracle_mod, abi):
    host, device = native
    sc = load_for_both(name, oracle_mod, host, width=64, height=H, spp=5, max_depth=8)
    opt = abi.Options.make(**opt_kw)
    mp, ml, mst = device.render_multi(sc, sc.camera, sc.settings, devices, opt)
    window = abi.Options.make(row_begin=opt.row_begin, row_end=opt.row_end)
    gp, gl, st = device.render(sc, sc.camera, sc.settings, window)
    assert mp.shape == gp.shape and np.array_equal(mp, gp) and np.array_equal(ml.view(np.uint32), gl.view(np.uint32))
    assert (mst.samples, mst.rays, mst.rows_rendered) == (st.samples, st.rays, st.rows_rendered)


def test_multi_argument_checks(native, oracle_mod, abi):
    host, device = native
    sc = load_for_both("cornell", oracle_mod, host, width=16, height=8, spp=2, max_depth=3)
    with pytest.raises(RuntimeError, match="out of range"):
        device.render_multi(sc, sc.camera, sc.settings, [0, 99])
    with pytest.raises(RuntimeError, match="deals the strips itself"):
        device.render_multi(sc, sc.camera, sc.settings, [0, 0], abi.Options.make(n_parts=2, part=1))
    with pytest.raises(RuntimeError, match="empty"):
        device.render_multi(sc, sc.camera, sc.settings, [])
