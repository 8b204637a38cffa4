"""Build recipes for the native libraries (in-tree, gfx950 only).

  libmi355rt.so       hipcc: HIP kernels + the device half of the C ABI (csrc/device)
  libmi355rt_host.so  g++:   CPU-side producers -- scene loader, mesh readers, BVH build, PNG (csrc/host)
  rt_render           g++:   CLI that stands in for the Rust `main` (csrc/tools)

Outputs go to raytracer-rust_amd/_build/ (git-ignored, but they travel to the GPU box with gpurun).
hipcc cross-compiles gfx950 without a GPU, so this also runs in the CPU-only container.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "_build")
CSRC = os.path.join(HERE, "csrc")

DEVICE_SO = os.path.join(OUT, "libmi355rt.so")
HOST_SO = os.path.join(OUT, "libmi355rt_host.so")
CLI = os.path.join(OUT, "rt_render")

# -ffp-contract=off: no FMA contraction -- the reference (rustc) never fuses a*b+c, and parity with the
# CPU oracle is bit-level.  Correctly rounded f32 divide/sqrt are HIP's default; stated explicitly.
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent f32 ops into v_pk_mul/add_f32, which issue slower than
# the two scalar ops they replace on gfx950 (measured: -6 % kernel time on cornell, identical results).
# -amdgpu-atomic-optimizer-strategy=None: the kernels' atomics (the work cursor, the wavefront kernels' queue tickets) are issued by ONE lane already; the
# optimizer pass wraps each of them in a second single-lane election (mbcnt, bcnt, a multiply, readfirstlane: ~10 instructions per atomic) that can never
# merge anything.  Round 5: mesh scenes and veach-mis -0.6 ... -0.8 %, cornell +-0 (profiles/r05/ab_scalar_diet.txt).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-command-line-argument"]
CXX_FLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-Wall", "-pthread"]

DEVICE_SRCS = [os.path.join(CSRC, "device", "rt_kernels.hip"), os.path.join(CSRC, "device", "rt_api.cpp")]
DEVICE_HEADERS = sorted(os.path.join(CSRC, "device", f) for f in os.listdir(os.path.join(CSRC, "device")) if f.endswith(".h"))
DEVICE_DEPS = DEVICE_SRCS + DEVICE_HEADERS + [os.path.join(ROOT, "include", "mi355rt.h")]


def _host_srcs():
    d = os.path.join(CSRC, "host")
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".cpp")) if os.path.isdir(d) else []


def _host_deps():
    d = os.path.join(CSRC, "host")
    hs = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hpp", ".h"))) if os.path.isdir(d) else []
    return _host_srcs() + hs + [os.path.join(ROOT, "include", "mi355rt.h")]


def _code_only(text):
    """C++ source without its comments, whitespace runs collapsed (string and character literals are kept as they are)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":                                      # a literal: copy to its closing quote
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1]); i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i); i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2); i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c); i += 1
    return " ".join("".join(out).split())


def kernel_hash():
    """sha256 over everything that decides the device code AND how it is launched: the kernel source, every header of csrc/device, the
    host half (rt_api.cpp: grid size, shard size, guided_div, the choice of the kernel variant) and the hipcc flags -- the CODE of those
    files: comments and whitespace are stripped first, so that correcting a comment does not orphan the committed counters.
    profiles/pmc_counters.json records it, and bench.py refuses counters taken on another library."""
    import hashlib
    h = hashlib.sha256()
    for f in DEVICE_SRCS + DEVICE_HEADERS:                  # rt_kernels.hip, rt_api.cpp and every header they include (rt_device.h, rt_math.h, ...)
        h.update(_code_only(open(f, encoding="utf-8").read()).encode())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def hipcc_path():
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: the HIP path cannot be built (there is no CPU fallback)")
    return p


def build_device(force=False, extra_flags=(), verbose=False):
    os.makedirs(OUT, exist_ok=True)
    if force or _stale(DEVICE_SO, DEVICE_DEPS):
        cmd = [hipcc_path(), *HIPCC_FLAGS, *extra_flags, "-o", DEVICE_SO, *DEVICE_SRCS]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return DEVICE_SO


def build_device_variant(name, defines=(), force=False, verbose=False, flags=()):
    """Diagnostic / A-B builds (e.g. cycle stamps) into their own library; never loaded by the product path."""
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, f"libmi355rt_{name}.so")
    if force or _stale(so, DEVICE_DEPS):
        cmd = [hipcc_path(), *HIPCC_FLAGS, *flags, *[f"-D{d}" for d in defines], "-o", so, *DEVICE_SRCS]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return so


def build_host(force=False, verbose=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = _host_srcs()
    if not srcs:
        raise RuntimeError("csrc/host has no sources")
    if force or _stale(HOST_SO, _host_deps()):
        cmd = ["g++", *CXX_FLAGS, "-shared", "-o", HOST_SO, *srcs, "-lz"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_SO


def build_cli(force=False, verbose=False):
    src = os.path.join(CSRC, "tools", "rt_render.cpp")
    if not os.path.exists(src):
        return None
    build_device(force=False)
    build_host(force=False)
    if force or _stale(CLI, [src, DEVICE_SO, HOST_SO]):
        cmd = ["g++", *CXX_FLAGS, "-o", CLI, src, "-L" + OUT, "-lmi355rt", "-lmi355rt_host",
               "-Wl,-rpath,$ORIGIN", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return CLI


def build_verify(force=False, verbose=False):
    """tools/verify/short_arithmetic: the exhaustive comparison of rt_math.h's short reciprocal / square root / division (and rt_rng.h's
    range conversion) with the compiler's correctly rounded forms -- the product's headers, the product's flags, a standalone program
    that a gpu test runs (tests/test_gpu_short_arithmetic.py)."""
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "verify", "short_arithmetic.hip")
    exe = src[:-4]
    if force or _stale(exe, [src] + DEVICE_HEADERS):
        cmd = [hipcc_path(), *[f for f in HIPCC_FLAGS if f not in ("-fPIC", "-shared")], "-Wno-unused-value", "-Wno-unused-result", "-o", exe, src]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return exe


def build_all(force=False, verbose=False):
    return build_device(force, verbose=verbose), build_host(force, verbose=verbose), build_cli(force, verbose=verbose)


if __name__ == "__main__":
    import sys
    print(build_all(force="--force" in sys.argv, verbose=True))
