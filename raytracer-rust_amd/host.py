"""ctypes binding of libmi355rt_host.so: the CPU-side producers (scene loader, mesh readers, BVH
build, PNG writer) that stand in for the Rust host.  Pure CPU -- importable without a GPU."""
import ctypes as C
import os

import numpy as np

from . import abi, build

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(build.HOST_SO):
            raise RuntimeError(f"{build.HOST_SO} is missing: run __graft_entry__.build() first")
        L = C.CDLL(build.HOST_SO)
        L.mi355rt_host_last_error.restype = C.c_char_p
        L.mi355rt_bvh_build.restype = C.c_int
        L.mi355rt_bvh_build.argtypes = [C.POINTER(abi.Triangle), C.c_uint32, C.POINTER(abi.BvhNode), C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.mi355rt_host_struct_sizes.restype = C.c_uint32
        L.mi355rt_host_struct_sizes.argtypes = [C.POINTER(C.c_uint32), C.c_uint32]
        if hasattr(L, "mi355rt_scene_load_json"):
            L.mi355rt_scene_load_json.restype = C.c_int
            L.mi355rt_scene_load_json.argtypes = [C.c_char_p, C.POINTER(abi.LoadOverrides), C.POINTER(C.c_void_p)]
            L.mi355rt_scene_free.argtypes = [C.c_void_p]
            L.mi355rt_loaded_scene_get.restype = C.POINTER(abi.Scene)
            L.mi355rt_loaded_scene_get.argtypes = [C.c_void_p]
            L.mi355rt_loaded_scene_camera.restype = C.POINTER(abi.Camera)
            L.mi355rt_loaded_scene_camera.argtypes = [C.c_void_p]
            L.mi355rt_loaded_scene_settings.restype = C.POINTER(abi.Settings)
            L.mi355rt_loaded_scene_settings.argtypes = [C.c_void_p]
            L.mi355rt_write_pfm.restype = C.c_int
            L.mi355rt_write_pfm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
            L.mi355rt_write_exr.restype = C.c_int
            L.mi355rt_write_exr.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
            L.mi355rt_write_png.restype = C.c_int
            L.mi355rt_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        _lib = L
    return _lib


def last_error():
    return lib().mi355rt_host_last_error().decode()


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {last_error()}")


def bvh_build(triangles):
    """triangles: ctypes array of abi.Triangle.  Returns (nodes array, indices array, max_depth)."""
    n = len(triangles)
    nn, ni, md = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    _check(lib().mi355rt_bvh_build(triangles, n, None, C.byref(nn), None, C.byref(ni), C.byref(md)), "mi355rt_bvh_build")
    nodes = (abi.BvhNode * nn.value)()
    idx = (C.c_uint32 * ni.value)()
    _check(lib().mi355rt_bvh_build(triangles, n, nodes, C.byref(nn), idx, C.byref(ni), C.byref(md)), "mi355rt_bvh_build")
    return nodes, idx, md.value


def attach_bvh(scene):
    """Give an abi.Scene that only carries triangle soups (e.g. from the oracle-side loader) the
    nodes / tri_indices / mesh ranges the device library needs, using the product builder.
    Returns an object that keeps the new arrays alive."""
    sc = getattr(scene, "c", scene)
    all_nodes, all_idx, keep = [], [], []
    for m in range(sc.n_meshes):
        mesh = sc.meshes[m]
        tris = (abi.Triangle * mesh.triangle_count).from_address(
            C.addressof(sc.triangles.contents) + mesh.first_triangle * C.sizeof(abi.Triangle))
        nodes, idx, md = bvh_build(tris)
        mesh.first_node, mesh.node_count = sum(len(a) for a in all_nodes), len(nodes)
        mesh.first_index, mesh.index_count = sum(len(a) for a in all_idx), len(idx)
        mesh.max_depth = md
        all_nodes.append(nodes)
        all_idx.append(idx)
    nn, ni = sum(len(a) for a in all_nodes), sum(len(a) for a in all_idx)
    nodes_c = (abi.BvhNode * max(nn, 1))()
    idx_c = (C.c_uint32 * max(ni, 1))()
    o = 0
    for a in all_nodes:
        C.memmove(C.addressof(nodes_c) + o * C.sizeof(abi.BvhNode), a, len(a) * C.sizeof(abi.BvhNode))
        o += len(a)
    o = 0
    for a in all_idx:
        C.memmove(C.addressof(idx_c) + o * 4, a, len(a) * 4)
        o += len(a)
    sc.nodes, sc.n_nodes = nodes_c, nn
    sc.tri_indices, sc.n_tri_indices = idx_c, ni
    keep.extend([nodes_c, idx_c])
    return keep


class LoadedScene:
    """A scene loaded by the C++ host loader (mi355rt_scene_load_json)."""

    def __init__(self, path, width=0, height=0, spp=0, max_depth=0, skip_unknown_primitives=False, wo3_four_index_stride=False):
        ov = abi.LoadOverrides(width, height, spp, max_depth, 1 if skip_unknown_primitives else 0, 1 if wo3_four_index_stride else 0)
        h = C.c_void_p()
        _check(lib().mi355rt_scene_load_json(os.fsencode(path), C.byref(ov), C.byref(h)), f"mi355rt_scene_load_json({path})")
        self._h = h
        self.c = lib().mi355rt_loaded_scene_get(h).contents
        self.camera = lib().mi355rt_loaded_scene_camera(h).contents
        self.settings = lib().mi355rt_loaded_scene_settings(h).contents

    def close(self):
        if self._h:
            lib().mi355rt_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_pfm(path, linear, width, height):
    """linear: float32 [height, width, 3] pre-gamma image -> little-endian Portable FloatMap."""
    a = np.ascontiguousarray(linear, dtype=np.float32)
    assert a.size == width * height * 3
    _check(lib().mi355rt_write_pfm(os.fsencode(path), a.ctypes.data, width, height), "mi355rt_write_pfm")


def write_exr(path, linear, width, height):
    """linear: float32 [height, width, 3] pre-gamma image -> uncompressed 32-bit float OpenEXR (channels B, G, R)."""
    a = np.ascontiguousarray(linear, dtype=np.float32)
    assert a.size == width * height * 3
    _check(lib().mi355rt_write_exr(os.fsencode(path), a.ctypes.data, width, height), "mi355rt_write_exr")


def write_png(path, packed, width, height):
    a = np.ascontiguousarray(packed, dtype=np.uint32)
    _check(lib().mi355rt_write_png(os.fsencode(path), a.ctypes.data, width, height), "mi355rt_write_png")
