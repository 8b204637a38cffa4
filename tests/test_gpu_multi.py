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


def test_bench_runs_its_rccl_branch_with_one_rank(tmp_path, native, abi):
    """bench.py's multi-GPU branch (RCCL process group, barrier, all_gather_into_tensor on the launch stream, two frames in flight,
    all_reduce of the step time) cannot run with two ranks on a one-GPU box -- RCCL refuses two ranks per device -- but it can run
    with ONE: MI355RT_BENCH_FORCE_DIST=1 under torch.distributed.run.  The image must be the 1-GPU image (checksum)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MI355RT_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["forced_dist"] and line["n_gpus"] == 1 and line["config"]["frames_in_flight"] == 2
    assert "RCCL all-gather over xGMI" in line["config"]["parallelism"]
    host, device = native                                                    # the same frame through the plain one-GPU path
    sc = host.LoadedScene(os.path.join(root, "data/scenes/tungsten/cornell-box/scene.json"), 800, 600, 256, 30)
    packed = device.render(sc, sc.camera, sc.settings, abi.Options.make())[0]
    assert line["image_checksum"] == int(packed.astype(np.int64).sum())
