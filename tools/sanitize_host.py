#!/usr/bin/env python3
"""Build the CPU-side code with AddressSanitizer + UndefinedBehaviorSanitizer and run it over the shipped scenes and a set
of malformed inputs.  CPU only -- GPU sanitizers are not available on this pool.

  python tools/sanitize_host.py [--keep]

Builds tools/sanitize/driver.cpp + csrc/host/*.cpp + oracle/rt_oracle.cpp into one instrumented executable under
tools/sanitize/_build/, runs it in a scratch directory and fails on any sanitizer report or unexpected result."""
import glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "sanitize", "_build")
FLAGS = ["-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", "-fno-fast-math", "-pthread",
         "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]


def build():
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, "sanitize_driver")
    srcs = [os.path.join(ROOT, "tools", "sanitize", "driver.cpp"), os.path.join(ROOT, "oracle", "rt_oracle.cpp")] + \
        sorted(glob.glob(os.path.join(ROOT, "raytracer-rust_amd", "csrc", "host", "*.cpp")))
    deps = srcs + glob.glob(os.path.join(ROOT, "raytracer-rust_amd", "csrc", "host", "*.hpp")) + [os.path.join(ROOT, "include", "mi355rt.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call(["g++", *FLAGS, "-o", exe, *srcs, "-lz"])
    return exe


def run(keep=False):
    exe = build()
    tmp = tempfile.mkdtemp(prefix="mi355rt_san_")
    try:
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
        p = subprocess.run([exe, ROOT, tmp], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
        out = p.stdout
        bad = [l for l in out.splitlines() if "ERROR: AddressSanitizer" in l or "runtime error:" in l or "ERROR: LeakSanitizer" in l or l.startswith("UNEXPECTED")]
        return p.returncode, out, bad
    finally:
        if not keep:
            shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    rc, out, bad = run("--keep" in sys.argv)
    print(out[-6000:])
    print(f"exit {rc}; {len(bad)} finding(s)")
    sys.exit(1 if rc or bad else 0)
