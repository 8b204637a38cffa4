"""`python bench.py --gpus N` as typed must start its ranks itself: the parent process relays a launcher it starts as a
fresh child and never touches the GPU.  Here, without a GPU: a stub stands in for torch.distributed.run (MI355RT_BENCH_LAUNCHER) and writes
down what it was given; the parent's exit code, its relayed output and its fallback to --single-process are checked."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")

STUB = textwrap.dedent("""
    import json, os, sys
    json.dump({"argv": sys.argv[1:], "launch": os.environ.get("MI355RT_BENCH_LAUNCH"), "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
               "rank_in_env": "RANK" in os.environ}, open(os.environ["STUB_OUT"], "w"))
    if os.environ.get("STUB_LINE") == "1":
        print(json.dumps({"metric": "stub", "value": 1.0}), flush=True)
    sys.exit(int(os.environ.get("STUB_RC", "0")))
""")

# runs bench.py as __main__ and records, at exit, whether the PARENT ever imported torch or the product
RUNNER = textwrap.dedent("""
    import atexit, json, os, runpy, sys
    atexit.register(lambda: json.dump({"torch": "torch" in sys.modules, "product": any(m.startswith("raytracer-rust_amd") for m in sys.modules)},
                                      open(os.environ["PARENT_OUT"], "w")))
    sys.argv = [sys.argv[1]] + sys.argv[2:]
    runpy.run_path(sys.argv[0], run_name="__main__")
""")


def run_parent(tmp_path, args, **env_extra):
    stub = tmp_path / "stub_launcher.py"
    stub.write_text(STUB)
    runner = tmp_path / "runner.py"
    runner.write_text(RUNNER)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MI355RT_BENCH_FORCE_DIST", "MI355RT_BENCH_REHEARSE")}
    env.update(MI355RT_BENCH_LAUNCHER=f"{sys.executable} {stub}", STUB_OUT=str(tmp_path / "stub.json"), PARENT_OUT=str(tmp_path / "parent.json"), **env_extra)
    out = subprocess.run([sys.executable, str(runner), BENCH, *args], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    seen = json.load(open(tmp_path / "stub.json")) if (tmp_path / "stub.json").exists() else None
    parent = json.load(open(tmp_path / "parent.json"))
    return out, seen, parent


def test_gpus_2_starts_the_launcher_itself_and_relays_it(tmp_path):
    out, seen, parent = run_parent(tmp_path, ["--gpus", "2", "--steps", "3", "--warmup", "1"], STUB_LINE="1")
    assert out.returncode == 0, out.stderr
    a = seen["argv"]
    assert a[0] == "--nnodes=1" and a[1] == "--nproc-per-node=2"
    assert a[2:4] == ["--master-addr", "127.0.0.1"] and a[4] == "--master-port" and 1024 <= int(a[5]) < 65536
    assert os.path.samefile(a[6], BENCH) and a[7:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert seen["ipc"] == "0" and not seen["rank_in_env"] and seen["launch"].startswith("self: bench.py --gpus 2 started torch.distributed.run")
    assert json.loads(out.stdout.strip().splitlines()[-1]) == {"metric": "stub", "value": 1.0}          # the child's line, passed through
    assert parent == {"torch": False, "product": False}                                                   # the parent stayed off torch, the product and the GPU


def test_failed_launch_relays_the_code_or_falls_back_to_one_process(tmp_path):
    # no fallback wanted: the launcher's exit code is the parent's
    out, seen, parent = run_parent(tmp_path, ["--gpus", "4"], STUB_RC="3", MI355RT_BENCH_NO_FALLBACK="1")
    assert out.returncode == 3 and seen["argv"][1] == "--nproc-per-node=4" and parent["torch"] is False
    # a launch that ends cleanly but without a result line is a failure too
    out, _, _ = run_parent(tmp_path, ["--gpus", "4"], MI355RT_BENCH_NO_FALLBACK="1")
    assert out.returncode == 1
    # default: a second fresh child in --single-process mode.  Without a GPU that child stops at once ("no GPU visible"), and says so.
    out, _, parent = run_parent(tmp_path, ["--gpus", "2", "--cpu-seconds", "0"], STUB_RC="5")
    assert out.returncode != 0 and parent["torch"] is False
    assert "starting ONE fresh process that drives all 2 devices (--single-process)" in out.stderr
    assert "no GPU visible" in out.stderr


def test_a_launcher_that_cannot_be_started_is_a_failed_launch(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MI355RT_BENCH_FORCE_DIST", "MI355RT_BENCH_REHEARSE")}
    env.update(MI355RT_BENCH_LAUNCHER=str(tmp_path / "no_such_launcher"), MI355RT_BENCH_NO_FALLBACK="1")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 127 and "could not start the launcher" in out.stderr and "Traceback" not in out.stderr


def test_under_a_launcher_nothing_is_launched(tmp_path):
    # RANK in the environment = a launcher (the driver's torch.distributed.run) is already around this process: it is a rank, not a parent.
    # Without a GPU the rank stops at "no GPU visible"; the stub must not have been started.
    stub_out = tmp_path / "stub.json"
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29599",
               MI355RT_BENCH_LAUNCHER=f"{sys.executable} -c pass", STUB_OUT=str(stub_out))
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--cpu-seconds", "0"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "no GPU visible" in out.stderr and not stub_out.exists()


def test_launcher_command_line():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    old = os.environ.pop("MI355RT_BENCH_LAUNCHER", None)
    try:
        cmd = bench.launcher_command(8, ["--gpus", "8"])
    finally:
        if old is not None:
            os.environ["MI355RT_BENCH_LAUNCHER"] = old
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and cmd[-2:] == ["--gpus", "8"]
