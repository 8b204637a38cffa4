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


HANG_STUB = textwrap.dedent("""
    import json, os, subprocess, sys, time
    # a launcher that never gets anywhere -- with a child of its own, as torch.distributed.run has its ranks
    kid = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])
    json.dump({"pid": os.getpid(), "kid": kid.pid}, open(os.environ["STUB_OUT"], "w"))
    if os.environ.get("STUB_LINE") == "1":
        print(json.dumps({"metric": "stub", "value": 2.0}), flush=True)      # the measurement is out ... and then teardown hangs
    time.sleep(600)
""")
FALLBACK_STUB = textwrap.dedent("""
    import json, os, sys
    json.dump({"argv": sys.argv[1:], "launch": os.environ.get("MI355RT_BENCH_LAUNCH"), "info": json.loads(os.environ.get("MI355RT_BENCH_FALLBACK_INFO", "null"))},
              open(os.environ["FALLBACK_OUT"], "w"))
    print(json.dumps({"metric": "fallback-stub", "launch": os.environ.get("MI355RT_BENCH_LAUNCH"), **json.loads(os.environ.get("MI355RT_BENCH_FALLBACK_INFO", "{}"))}), flush=True)
""")


def gone(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return True
    try:                                                      # a zombie that nobody reaped yet is gone for our purposes
        return open(f"/proc/{pid}/stat").read().split(") ")[1][0] == "Z"
    except OSError:
        return True


def run_with(tmp_path, stub_text, args, **env_extra):
    stub = tmp_path / "hang_launcher.py"
    stub.write_text(stub_text)
    fb = tmp_path / "fallback_stub.py"
    fb.write_text(FALLBACK_STUB)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MI355RT_BENCH_FORCE_DIST", "MI355RT_BENCH_REHEARSE")}
    env.update(MI355RT_BENCH_LAUNCHER=f"{sys.executable} {stub}", MI355RT_BENCH_FALLBACK_CMD=f"{sys.executable} {fb}",
               STUB_OUT=str(tmp_path / "stub.json"), FALLBACK_OUT=str(tmp_path / "fallback.json"), **env_extra)
    import time
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, BENCH, *args], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    return out, time.monotonic() - t0


def test_a_launch_that_hangs_is_killed_at_the_deadline_and_the_fallback_runs(tmp_path):
    """a rendezvous or RCCL-init hang must not end as the DRIVER's kill with no line.  The launcher stub sleeps forever (so does
    its own child); after --launch-timeout the parent ends the whole process group and starts the --single-process child, whose line says why."""
    out, took = run_with(tmp_path, HANG_STUB, ["--gpus", "8", "--launch-timeout", "3"])
    assert took < 30, took                                                       # 3 s deadline + 5 s SIGTERM wait at most + interpreter starts
    ids = json.load(open(tmp_path / "stub.json"))
    assert gone(ids["pid"]) and gone(ids["kid"])                                 # the launcher AND what it had started
    assert "no result line within 3 s" in out.stderr and "starting ONE fresh process that drives all 8 devices" in out.stderr
    fb = json.load(open(tmp_path / "fallback.json"))
    assert fb["argv"][-1] == "--single-process" and fb["argv"][:2] == ["--gpus", "8"]
    assert "timed out" in fb["launch"] and fb["launch"].startswith("fallback: one process drives 8 devices")
    assert fb["info"] == {"fallback": True, "rank_launch_rc": 124, "rank_launch_timed_out": True, "rank_launch_seconds": fb["info"]["rank_launch_seconds"]}
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert out.returncode == 0 and line["fallback"] is True and line["rank_launch_timed_out"] is True and "timed out" in line["launch"]
    # without the fallback the exit code says "timed out" the way timeout(1) does
    out, _ = run_with(tmp_path, HANG_STUB, ["--gpus", "8", "--launch-timeout", "2"], MI355RT_BENCH_NO_FALLBACK="1")
    assert out.returncode == 124 and "timed out" in out.stderr and out.stdout.strip() == ""


def test_a_launch_that_hangs_in_teardown_keeps_its_line(tmp_path):
    """The ranks printed their result line and then hang (destroy_process_group, the launcher's own exit): the line is relayed, the group is
    ended after --teardown-grace, the exit code is 0 and NO second measurement is made."""
    out, took = run_with(tmp_path, HANG_STUB, ["--gpus", "4", "--launch-timeout", "30", "--teardown-grace", "2"], STUB_LINE="1")
    assert out.returncode == 0 and took < 30, (out.returncode, took, out.stderr)
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert [json.loads(l) for l in lines] == [{"metric": "stub", "value": 2.0}]          # one line: the launch's
    assert not (tmp_path / "fallback.json").exists() and "teardown" in out.stderr
    ids = json.load(open(tmp_path / "stub.json"))
    assert gone(ids["pid"]) and gone(ids["kid"])


def test_a_launch_that_printed_its_line_and_then_failed_is_not_repeated(tmp_path):
    """ADVICE r4 (bench.py:426): ranks that die in teardown AFTER their line must not trigger the fallback -- a second line under the same n_gpus
    would be a different kind of measurement.  The launcher's code is passed on."""
    out, seen, _ = run_parent(tmp_path, ["--gpus", "2"], STUB_LINE="1", STUB_RC="7", MI355RT_BENCH_FALLBACK_CMD=f"{sys.executable} -c raise_SystemExit(99)")
    assert out.returncode == 7 and "no fallback" in out.stderr
    assert [json.loads(l) for l in out.stdout.strip().splitlines()] == [{"metric": "stub", "value": 1.0}]


def test_the_rank_path_bounds_its_rendezvous_and_announces_itself():
    src = open(BENCH).read()
    assert 'init_process_group(backend="nccl", device_id=dev, timeout=limit)' in src and "--rendezvous-timeout" in src
    assert src.index("on device {local_rank}") < src.index('init_process_group(backend="gloo"')      # the per-rank line comes BEFORE the first collective
    bench = __import__("importlib").import_module("bench")
    # frames in flight per workload, each slot told its share of the device (mi355rt_context_set_share): never more than HIP's 4 hardware queues
    for wl in bench.WORKLOADS:
        frames, share = bench.frames_in_flight(wl)
        assert 1 <= frames <= 4 and 1 <= share <= frames
    assert bench.frames_in_flight("cornell-box-800x600x256-d30") == (4, 4) and bench.frames_in_flight("teapot-800x600x256-d64") == (4, 2)


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
