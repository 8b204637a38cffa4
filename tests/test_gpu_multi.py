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


def _run_bench(args, **env_extra):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MI355RT_BENCH_FORCE_DIST", "MI355RT_BENCH_REHEARSE", "MI355RT_BENCH_LAUNCHER")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, (json.loads(lines[-1]) if lines else None)


def _one_gpu_checksum(native, abi, W, H, spp, depth):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host, device = native                                                    # the same frame through the plain one-GPU path
    sc = host.LoadedScene(os.path.join(root, "data/scenes/tungsten/cornell-box/scene.json"), W, H, spp, depth)
    return int(device.render(sc, sc.camera, sc.settings, abi.Options.make())[0].astype(np.int64).sum())


def test_bench_runs_its_rccl_branch_with_one_rank(native, abi):
    """bench.py's multi-GPU branch (RCCL process group, barrier, all_gather_into_tensor on the launch stream, two frames in flight,
    all_reduce of the step time) cannot run with two ranks on a one-GPU box -- RCCL refuses two ranks per device -- but it can run
    with ONE: MI355RT_BENCH_FORCE_DIST=1.  The command is the one the driver types, `python bench.py --gpus 1 ...`: bench.py starts
    torch.distributed.run itself (self-launch) and relays the rank's line.  The image must be the 1-GPU image (checksum)."""
    out, line = _run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0"], MI355RT_BENCH_FORCE_DIST="1")
    assert out.returncode == 0 and line, out.stderr[-2000:]
    assert line["forced_dist"] and line["n_gpus"] == 1
    cfg = line["config"]                                                     # 4 frames in flight on 1/4 of the device each -- or as many as hardware-queue classes were found
    found = cfg["streams"]["queue_classes_found"]
    assert (cfg["frames_in_flight"], cfg["share_of_device_per_frame"]) == ((4, 4) if found is None or found >= 4 else (found, found)), cfg
    assert "RCCL all-gather over xGMI" in line["config"]["parallelism"]
    assert line["launch"].startswith("self: bench.py --gpus 1 started torch.distributed.run")
    d = line["distributed"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and len(d["ranks"]) == 1 and d["ranks"][0]["device_index"] == 0
    assert d["ranks"][0]["kernel_ms_per_step"] > 0 and d["ranks"][0]["rows"] == 600
    assert line["image_checksum"] == _one_gpu_checksum(native, abi, 800, 600, 256, 30)


def test_bench_gpus_2_as_typed_starts_two_ranks(native, abi):
    """`python bench.py --gpus 2` with no launcher around it: two fresh rank processes (rehearsal: both on cuda:0, the gather over gloo --
    RCCL refuses two ranks on one device), one result line, the one-GPU image."""
    out, line = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--workload", "cornell-box-400x300x16-d4"], MI355RT_BENCH_REHEARSE="1")
    assert out.returncode == 0 and line, out.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["rehearsal"] and line["launch"].startswith("self: bench.py --gpus 2")
    d = line["distributed"]
    assert d["world_size"] == 2 and [r["rank"] for r in d["ranks"]] == [0, 1] and d["ranks"][0]["pid"] != d["ranks"][1]["pid"]
    assert all(r["rows"] == 150 and r["kernel_ms_per_step"] > 0 for r in d["ranks"])
    assert line["image_checksum"] == _one_gpu_checksum(native, abi, 400, 300, 16, 4)


def test_bench_single_process_mode_and_the_fallback_into_it(tmp_path, native, abi):
    """--single-process: one process, a context + stream per device, device-to-device copies into device 0 (rehearsal: every part on cuda:0).
    Reached directly, and as the fallback of `bench.py --gpus 2` when the rank launch fails (a stub launcher that exits 9)."""
    import sys
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--workload", "cornell-box-400x300x16-d4"]
    want = _one_gpu_checksum(native, abi, 400, 300, 16, 4)
    out, line = _run_bench(args + ["--single-process"], MI355RT_BENCH_REHEARSE="1")
    assert out.returncode == 0 and line, out.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["launch"] == "direct: --single-process" and line["image_checksum"] == want
    assert line["distributed"]["world_size"] == 2 and [r["part"] for r in line["distributed"]["ranks"]] == [0, 1]
    stub = tmp_path / "failing_launcher.py"
    stub.write_text("import sys; sys.exit(9)\n")
    out, line = _run_bench(args, MI355RT_BENCH_REHEARSE="1", MI355RT_BENCH_LAUNCHER=f"{sys.executable} {stub}")
    assert out.returncode == 0 and line, out.stderr[-2000:]
    assert line["launch"].startswith("fallback: one process drives 2 devices") and "ended with code 9" in line["launch"]
    assert line["image_checksum"] == want and "device-to-device copies" in line["config"]["parallelism"]
