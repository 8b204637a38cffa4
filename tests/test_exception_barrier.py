"""mi355rt.h: "nothing aborts, nothing throws across the ABI".  Every extern "C" entry of libmi355rt.so and libmi355rt_host.so runs inside
an exception barrier; here host allocations are made to fail (RLIMIT_AS lowered after the libraries are loaded, in a subprocess) and the
calls must come back with MI355RT_ERR_OOM and a message -- not std::terminate.  No GPU and no compute: the calls fail before any HIP work.
Reference contract: the call the library replaces is infallible-or-panic inside one Rust process (src/renderer.rs:67)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import ctypes as C, importlib, json, os, resource, sys
    sys.path.insert(0, sys.argv[1])
    abi = importlib.import_module("raytracer-rust_amd.abi")
    build = importlib.import_module("raytracer-rust_amd.build")
    dev = C.CDLL(build.DEVICE_SO); host = C.CDLL(build.HOST_SO)
    dev.mi355rt_last_error.restype = C.c_char_p; host.mi355rt_host_last_error.restype = C.c_char_p
    dev.mi355rt_rows_selected.argtypes = [C.POINTER(abi.Settings), C.POINTER(abi.Options), C.POINTER(C.c_uint32)]
    host.mi355rt_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
    host.mi355rt_write_exr.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
    out = {}
    st = abi.Settings(1, (1 << 24) - 1, 1, 1)                       # 16 M rows: the row list alone is 64 MB
    n = C.c_uint32()
    out["rows_ok"] = [dev.mi355rt_rows_selected(C.byref(st), None, C.byref(n)), n.value]
    W = H = 8192                                                     # write_png's raw scanlines: 192 MB
    img = (C.c_uint32 * 16)()                                        # never read: the allocation fails first
    vm = int([l for l in open("/proc/self/status") if l.startswith("VmSize")][0].split()[1]) * 1024
    resource.setrlimit(resource.RLIMIT_AS, (vm + (24 << 20), vm + (24 << 20)))
    out["rows"] = [dev.mi355rt_rows_selected(C.byref(st), None, C.byref(n)), dev.mi355rt_last_error().decode()]
    out["png"] = [host.mi355rt_write_png(os.path.join(sys.argv[2], "x.png").encode(), img, W, H), host.mi355rt_host_last_error().decode()]
    out["exr"] = [host.mi355rt_write_exr(os.path.join(sys.argv[2], "x.exr").encode(), img, 1 << 24, 1), host.mi355rt_host_last_error().decode()]
    out["png_file"] = os.path.exists(os.path.join(sys.argv[2], "x.png"))
    resource.setrlimit(resource.RLIMIT_AS, (resource.RLIM_INFINITY, resource.RLIM_INFINITY)) if False else None
    print(json.dumps(out))
""")


def test_allocation_failures_come_back_as_error_codes(tmp_path):
    import json
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    out = subprocess.run([sys.executable, str(script), ROOT, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stderr[-2000:])          # -6 here would be SIGABRT: std::terminate
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["rows_ok"] == [0, (1 << 24) - 1]                                   # the same call succeeds while memory is to be had
    assert r["rows"][0] == -4 and "allocation failed" in r["rows"][1]           # MI355RT_ERR_OOM
    assert r["png"][0] == -4 and "write_png" in r["png"][1] and not r["png_file"]
    assert r["exr"][0] == -4 and "write_exr" in r["exr"][1]


def test_every_entry_point_sits_behind_the_barrier():
    """Source check: each `int mi355rt_*(...)` defined in rt_api.cpp opens with the guard, and the host writers with theirs."""
    import re
    api = open(os.path.join(ROOT, "raytracer-rust_amd/csrc/device/rt_api.cpp")).read()
    body = api[api.index('extern "C" {'):]
    defs = re.findall(r"\nint (mi355rt_\w+)\((?:[^()]|\([^()]*\))*\) \{\n(.*)\n", body)
    assert len(defs) >= 17
    bare = [n for n, first in defs if "return guard([&]() -> int {" not in first and n not in ("mi355rt_debug_has_variant",)]
    assert not bare, bare
    assert api.count("catch") >= 5
    png = open(os.path.join(ROOT, "raytracer-rust_amd/csrc/host/png_write.cpp")).read()
    assert png.count("return mi355rt_host::guard(") == 3
    # ... and EVERY `int mi355rt_*(...)` the host library defines, whichever file it lives in (bvh_build* and scene_load_json had
    # ad-hoc handlers that could throw again while building their message and had no catch (...))
    host_dir = os.path.join(ROOT, "raytracer-rust_amd/csrc/host")
    seen = []
    for name in sorted(os.listdir(host_dir)):
        if not name.endswith(".cpp"):
            continue
        src = open(os.path.join(host_dir, name)).read()
        for m in re.finditer(r"\n(?:extern \"C\" )?int (mi355rt_\w+)\((?:[^()]|\([^()]*\))*\) \{\n((?:.*\n){1,6})", src):
            seen.append(m.group(1))
            assert "mi355rt_host::guard(" in m.group(2), f"{name}: {m.group(1)} does not open with the barrier"
    assert {"mi355rt_bvh_build", "mi355rt_bvh_build_threads", "mi355rt_scene_load_json", "mi355rt_write_png", "mi355rt_write_pfm", "mi355rt_write_exr"} <= set(seen), seen
