"""Python binding of the CPU oracle (oracle/rt_oracle.cpp).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package.  The oracle reuses the product's *public* ABI
declarations (include/mi355rt.h, mirrored in raytracer-rust_amd/abi.py) as its input format.
"""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_SO = os.path.join(_HERE, "_build", "librt_oracle.so")
_SRC = os.path.join(_HERE, "rt_oracle.cpp")

abi = importlib.import_module("raytracer-rust_amd.abi")

CXXFLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-pthread"]


def build(force=False):
    """g++ the oracle into oracle/_build/ (git-ignored, travels to the GPU box with the snapshot)."""
    hdr = os.path.join(_ROOT, "include", "mi355rt.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(_SRC), os.path.getmtime(hdr),
                                             os.path.getmtime(os.path.join(os.path.dirname(_SRC), "rust_sort_unstable.hpp")))):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    subprocess.check_call(["g++", *CXXFLAGS, "-o", _SO, _SRC])
    return _SO


class Counters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("prim_tests", C.c_uint64),
                ("bvh_nodes", C.c_uint64), ("tri_tests", C.c_uint64), ("rng_words", C.c_uint64),
                ("depth_exhausted", C.c_uint64), ("seconds", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.oracle_render.restype = C.c_int
        L.oracle_render.argtypes = [C.POINTER(abi.Scene), C.POINTER(abi.Camera), C.POINTER(abi.Settings),
                                    C.POINTER(abi.Options), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.POINTER(Counters)]
        L.oracle_chacha_key.argtypes = [C.c_uint64, C.c_void_p]
        L.oracle_chacha_words.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p]
        L.oracle_u32_to_f01.restype = C.c_float
        L.oracle_u32_to_f01.argtypes = [C.c_uint32]
        L.oracle_u32_to_range11.restype = C.c_float
        L.oracle_u32_to_range11.argtypes = [C.c_uint32]
        L.oracle_philox4x32_10.argtypes = [C.c_uint32] * 6 + [C.c_void_p]
        L.oracle_philox4x32.argtypes = [C.c_uint32] * 6 + [C.c_int, C.c_void_p]
        L.oracle_pcg4d.argtypes = [C.c_uint32] * 4 + [C.c_void_p]
        L.oracle_ctr_block.argtypes = [C.c_int] + [C.c_uint32] * 6 + [C.c_void_p]
        L.oracle_set_ctr_gen.restype = C.c_int
        L.oracle_set_ctr_gen.argtypes = [C.c_int]
        L.oracle_get_ctr_gen.restype = C.c_int
        L.oracle_color_to_u32.restype = C.c_uint32
        L.oracle_color_to_u32.argtypes = [C.c_float] * 3
        L.oracle_scene_hit.restype = C.c_int
        L.oracle_scene_hit.argtypes = [C.POINTER(abi.Scene), C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_scatter_ctr.restype = C.c_int
        L.oracle_scatter_ctr.argtypes = [C.POINTER(abi.Material), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_void_p]
        L.oracle_sort_paths.restype = None
        L.oracle_sort_paths.argtypes = [C.POINTER(C.c_uint64), C.c_int]
        L.oracle_bvh_dump.restype = C.c_int
        L.oracle_bvh_dump.argtypes = [C.POINTER(abi.Triangle), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _lib = L
    return _lib


def chacha_key(seed):
    out = np.zeros(8, np.uint32)
    lib().oracle_chacha_key(seed, out.ctypes.data)
    return out


def chacha_words(seed, n):
    out = np.zeros(n, np.uint32)
    lib().oracle_chacha_words(seed, n, out.ctypes.data)
    return out


def philox(k0, k1, c0, c1, c2, c3):
    out = np.zeros(4, np.uint32)
    lib().oracle_philox4x32_10(k0, k1, c0, c1, c2, c3, out.ctypes.data)
    return out


def philox_rounds(k0, k1, c0, c1, c2, c3, rounds):
    out = np.zeros(4, np.uint32)
    lib().oracle_philox4x32(k0, k1, c0, c1, c2, c3, rounds, out.ctypes.data)
    return out


def pcg4d(x, y, z, w):
    out = np.zeros(4, np.uint32)
    lib().oracle_pcg4d(x, y, z, w, out.ctypes.data)
    return out


def ctr_block(k0, k1, x, s, ray, j, gen=-1):
    """The 4 words of block j of the event after ray `ray` of path (k0, k1; x, s); gen -1 = the generator in force."""
    out = np.zeros(4, np.uint32)
    lib().oracle_ctr_block(gen, k0, k1, x, s, ray, j, out.ctypes.data)
    return out


def set_ctr_gen(gen):
    """Select the counter-mode generator (0 Philox4x32-10, 1 Philox4x32-7, 2 pcg4d) the oracle checks against; returns the previous one."""
    return lib().oracle_set_ctr_gen(gen)


def ctr_gen():
    return lib().oracle_get_ctr_gen()


def render(scene, camera, settings, options=None, threads=0, fold=-1, want_linear=True):
    """Run the oracle.  `scene` is an abi.Scene (or anything with a `.c` attribute holding one).

    Returns (packed u32 [rows, W], linear f32 [rows, W, 3] or None, Counters).
    """
    sc = getattr(scene, "c", scene)
    rows = len(abi.rows_selected(settings.height, options))
    W = settings.width
    packed = np.zeros((rows, W), np.uint32)
    linear = np.zeros((rows, W, 3), np.float32) if want_linear else None
    cnt = Counters()
    rc = lib().oracle_render(C.byref(sc), C.byref(camera), C.byref(settings),
                             C.byref(options) if options is not None else None, threads, fold,
                             packed.ctypes.data, linear.ctypes.data if want_linear else None, C.byref(cnt))
    if rc != 0:
        raise RuntimeError(f"oracle_render failed: {rc}")
    return packed, linear, cnt


def scene_hit(scene, origin, direction):
    sc = getattr(scene, "c", scene)
    o = np.asarray(origin, np.float32)
    d = np.asarray(direction, np.float32)
    out = np.zeros(9, np.float32)
    rc = lib().oracle_scene_hit(C.byref(sc), o.ctypes.data, d.ctypes.data, out.ctypes.data)
    if rc < 0:
        raise RuntimeError(f"oracle_scene_hit failed: {rc}")
    return (rc == 1), out


SORT_PATHS = ("calls", "insertion_only", "run_ascending", "run_descending", "quicksort", "small_no_merge", "sort9", "sort13", "merge", "merge_odd",
              "median3", "median3_rec", "partition_lt", "partition_le_ancestor", "heapsort")


def sort_paths(reset=False):
    """Branch counters of the oracle's sort_unstable_by restatement since the last reset (rust_sort_unstable.hpp g_paths)."""
    L = lib()
    out = (C.c_uint64 * len(SORT_PATHS))()
    L.oracle_sort_paths(out, 1 if reset else 0)
    return dict(zip(SORT_PATHS, [int(v) for v in out]))


def bvh_dump(triangles):
    """triangles: ctypes array of abi.Triangle.  Returns dict(bounds, info, leaf_ids, max_depth)."""
    n = len(triangles)
    nn, nl, md = C.c_uint32(), C.c_uint32(), C.c_uint32()
    L = lib()
    rc = L.oracle_bvh_dump(triangles, n, None, None, None, C.byref(nn), C.byref(nl), C.byref(md))
    if rc != 0:
        raise RuntimeError("oracle_bvh_dump failed")
    bounds = np.zeros((nn.value, 6), np.float32)
    info = np.zeros((nn.value, 3), np.uint32)
    leaf = np.zeros(nl.value, np.uint32)
    L.oracle_bvh_dump(triangles, n, bounds.ctypes.data, info.ctypes.data, leaf.ctypes.data,
                      C.byref(nn), C.byref(nl), C.byref(md))
    return {"bounds": bounds, "info": info, "leaf_ids": leaf, "max_depth": md.value}
