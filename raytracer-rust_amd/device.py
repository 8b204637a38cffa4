"""ctypes binding of libmi355rt.so -- the HIP path behind the C ABI (include/mi355rt.h).

There is no CPU fallback: `lib()` raises if the HIP library has not been built, and every render call
raises if no GPU is visible.  torch is NOT needed here; callers that hold torch tensors pass
`tensor.data_ptr()` and `torch.cuda.current_stream().cuda_stream`.
"""
import ctypes as C
import os

import numpy as np

from . import abi, build

EXPORTS = ["mi355rt_render", "mi355rt_render_multi", "mi355rt_render_progressive", "mi355rt_context_create", "mi355rt_context_destroy", "mi355rt_context_set_scene",
           "mi355rt_rows_selected", "mi355rt_context_render", "mi355rt_context_render_progressive", "mi355rt_context_set_timing", "mi355rt_context_read_timing",
           "mi355rt_context_check", "mi355rt_context_set_share", "mi355rt_last_error", "mi355rt_abi_version"]

_lib = None
_extra = {}


def load(so):
    """Bind another build of the device library (diagnostic / reference builds; the product path uses lib())."""
    so = os.path.abspath(so)
    if so not in _extra:
        _extra[so] = _bind(so)
    return _extra[so]


def refs():
    """The tests' reference build: the product sources plus the retired mesh kernel (the state machine), -DMI355RT_REFS."""
    return load(build.build_device_variant("refs", ["MI355RT_REFS"]))


def lib():
    global _lib
    if _lib is None:
        _lib = _bind(os.environ.get("MI355RT_DEVICE_SO", build.DEVICE_SO))     # override: diagnostic builds only (tools/)
    return _lib


def _bind(so):
    if True:
        if not os.path.exists(so):
            raise RuntimeError(f"{so} is missing: the HIP extension must be built "
                               "(__graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(so)
        L.mi355rt_last_error.restype = C.c_char_p
        L.mi355rt_abi_version.restype = C.c_uint32
        L.mi355rt_render.restype = C.c_int
        L.mi355rt_render.argtypes = [C.POINTER(abi.Scene), C.POINTER(abi.Camera), C.POINTER(abi.Settings),
                                     C.POINTER(abi.Options), C.c_void_p, C.c_void_p, C.POINTER(abi.Stats)]
        L.mi355rt_render_multi.restype = C.c_int
        L.mi355rt_render_multi.argtypes = [C.POINTER(abi.Scene), C.POINTER(abi.Camera), C.POINTER(abi.Settings), C.POINTER(abi.Options),
                                           C.POINTER(C.c_int), C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(abi.Stats)]
        L.mi355rt_context_create.restype = C.c_int
        L.mi355rt_context_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.mi355rt_context_destroy.argtypes = [C.c_void_p]
        L.mi355rt_context_set_scene.restype = C.c_int
        L.mi355rt_context_set_scene.argtypes = [C.c_void_p, C.POINTER(abi.Scene), C.POINTER(abi.Camera), C.POINTER(abi.Settings)]
        L.mi355rt_rows_selected.restype = C.c_int
        L.mi355rt_rows_selected.argtypes = [C.POINTER(abi.Settings), C.POINTER(abi.Options), C.POINTER(C.c_uint32)]
        L.mi355rt_context_render.restype = C.c_int
        L.mi355rt_context_render.argtypes = [C.c_void_p, C.POINTER(abi.Options), C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.POINTER(abi.Stats)]
        L.mi355rt_context_render_progressive.restype = C.c_int
        L.mi355rt_context_render_progressive.argtypes = [C.c_void_p, C.POINTER(abi.Options), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                                         C.c_void_p, C.c_void_p, C.POINTER(abi.Stats)]
        L.mi355rt_context_set_timing.restype = C.c_int
        L.mi355rt_context_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.mi355rt_context_read_timing.restype = C.c_int
        L.mi355rt_context_read_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
        L.mi355rt_context_check.restype = C.c_int
        L.mi355rt_context_check.argtypes = [C.c_void_p]
        L.mi355rt_context_set_share.restype = C.c_int
        L.mi355rt_context_set_share.argtypes = [C.c_void_p, C.c_uint32]
        L.mi355rt_debug_set_knob.restype = C.c_int
        L.mi355rt_debug_set_knob.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.mi355rt_debug_has_variant.restype = C.c_int
        L.mi355rt_debug_has_variant.argtypes = [C.c_uint32]
        if L.mi355rt_abi_version() != abi.ABI_VERSION:
            raise RuntimeError("libmi355rt.so ABI version does not match abi.py")
    return L


class RenderError(RuntimeError):
    def __init__(self, what, rc, library=None):
        super().__init__(f"{what} failed ({rc}): {(library or lib()).mi355rt_last_error().decode()}")
        self.rc = rc


def _check(rc, what, library=None):
    if rc != 0:
        raise RenderError(what, rc, library)


def set_knob(name, value, library=None):
    """Diagnostic: process-wide default knob for every context created afterwards (also inside the one-shot calls).
    Knobs: kernel, guided_mult, spin_idle, spin_entry, wave_times, row_order, inline_steps, trav_min (rt_api.cpp)."""
    L = library or lib()
    _check(L.mi355rt_debug_set_knob(None, name.encode(), int(value)), f"mi355rt_debug_set_knob({name})", L)


def clear_knobs(library=None):
    L = library or lib()
    _check(L.mi355rt_debug_set_knob(None, None, 0), "mi355rt_debug_set_knob(clear)", L)


def render(scene, camera, settings, options=None, want_linear=True, want_stats=True, library=None):
    """One-shot mi355rt_render with host buffers (what src/main.rs:57 would call).
    Returns (packed u32 [rows, W], linear f32 [rows, W, 3] or None, abi.Stats or None)."""
    L = library or lib()
    sc = getattr(scene, "c", scene)
    rows = len(abi.rows_selected(settings.height, options))
    W = settings.width
    packed = np.zeros((rows, W), np.uint32)
    linear = np.zeros((rows, W, 3), np.float32) if want_linear else None
    stats = abi.Stats() if want_stats else None
    _check(L.mi355rt_render(C.byref(sc), C.byref(camera), C.byref(settings),
                            C.byref(options) if options is not None else None,
                            packed.ctypes.data, linear.ctypes.data if want_linear else None,
                            C.byref(stats) if want_stats else None), "mi355rt_render", L)
    return packed, linear, stats


def render_multi(scene, camera, settings, devices, options=None, want_linear=True, library=None):
    """mi355rt_render_multi: one process, the listed HIP devices (a device may repeat).  Same returns as render()."""
    sc = getattr(scene, "c", scene)
    rows = len(abi.rows_selected(settings.height, options)) if options is not None else settings.height
    packed = np.zeros((rows, settings.width), np.uint32)
    linear = np.zeros((rows, settings.width, 3), np.float32) if want_linear else None
    stats = abi.Stats()
    devs = (C.c_int * len(devices))(*devices)
    L = library or lib()
    _check(L.mi355rt_render_multi(C.byref(sc), C.byref(camera), C.byref(settings), C.byref(options) if options is not None else None,
                                  devs, len(devices), packed.ctypes.data, linear.ctypes.data if want_linear else None, C.byref(stats)),
           "mi355rt_render_multi", L)
    return packed, linear, stats


def debug_scatter(materials, records, hip_device=0, textures=None):
    """Diagnostic: one Material::scatter per record on the device.  materials: ctypes array of abi.Material;
    records: (material index, front_face, rd[3], p[3], n[3], (k0, k1, x, s, ray)).  Returns float32 [n, 10] rows
    (scattered, origin[3], direction[3], attenuation[3]) -- the layout of the oracle's hook."""
    sc = abi.Scene()
    sc.materials, sc.n_materials = materials, len(materials)
    if textures is not None:
        sc.textures, sc.n_textures = textures, len(textures)
    sc.miss_color[:] = (0.5, 0.5, 0.5)
    ctx = Context(hip_device)
    try:
        ctx.set_scene(sc, abi.Camera(), abi.Settings(1, 1, 1, 1))
        n = len(records)
        rin = np.zeros((n, 16), np.uint32)
        for i, (mi, ff, rd, p, nn, ctr) in enumerate(records):
            rin[i, 0], rin[i, 1] = mi, 1 if ff else 0
            rin[i, 2:11] = np.concatenate([np.asarray(rd, np.float32), np.asarray(p, np.float32), np.asarray(nn, np.float32)]).view(np.uint32)
            rin[i, 11:16] = ctr
        out = np.zeros((n, 16), np.float32)
        _check(lib().mi355rt_debug_scatter(ctx._h, C.c_void_p(rin.ctypes.data), n, C.c_void_p(out.ctypes.data)), "mi355rt_debug_scatter")
        return out[:, :10].copy()
    finally:
        ctx.close()


def debug_hit(scene, rays, hip_device=0):
    """Diagnostic: closest hit of each (origin, unnormalised direction) against `scene` on the device.  Returns float32 [n, 10]:
    position[3], normal[3], t, material, front_face, hit."""
    ctx = Context(hip_device)
    try:
        ctx.set_scene(scene, abi.Camera(), abi.Settings(1, 1, 1, 1))
        n = len(rays)
        rin = np.zeros((n, 6), np.float32)
        for i, (o, d) in enumerate(rays):
            rin[i, :3], rin[i, 3:] = o, d
        out = np.zeros((n, 12), np.float32)
        _check(lib().mi355rt_debug_hit(ctx._h, C.c_void_p(rin.ctypes.data), n, C.c_void_p(out.ctypes.data)), "mi355rt_debug_hit")
        return out[:, :10].copy()
    finally:
        ctx.close()


class Context:
    """Resident-scene API: upload once, render many times into DEVICE buffers."""

    def __init__(self, hip_device=0, library=None):
        self._L = library or lib()
        self._h = C.c_void_p()
        _check(self._L.mi355rt_context_create(hip_device, C.byref(self._h)), "mi355rt_context_create", self._L)
        self.settings = None

    def set_knob(self, name, value):
        """Diagnostic knob of this context (before set_scene); see device.set_knob."""
        _check(self._L.mi355rt_debug_set_knob(self._h, name.encode(), int(value)), f"mi355rt_debug_set_knob({name})", self._L)

    def set_share(self, share_of):
        """mi355rt_context_set_share: this context is one of `share_of` contexts whose frames are in flight together (own stream each): its
        persistent kernels launch 1 / share_of of the grid that fills the device."""
        _check(self._L.mi355rt_context_set_share(self._h, int(share_of)), "mi355rt_context_set_share", self._L)

    def check(self):
        """mi355rt_context_check: waits for every render enqueued on this context and raises if a kernel left an image incomplete."""
        _check(self._L.mi355rt_context_check(self._h), "mi355rt_context_check", self._L)

    def set_scene(self, scene, camera, settings):
        sc = getattr(scene, "c", scene)
        _check(self._L.mi355rt_context_set_scene(self._h, C.byref(sc), C.byref(camera), C.byref(settings)),
               "mi355rt_context_set_scene", self._L)
        self.settings = abi.Settings(settings.width, settings.height, settings.samples_per_pixel, settings.max_depth)

    def rows_selected(self, options=None):
        n = C.c_uint32()
        _check(self._L.mi355rt_rows_selected(C.byref(self.settings), C.byref(options) if options is not None else None,
                                           C.byref(n)), "mi355rt_rows_selected", self._L)
        return n.value

    def render(self, d_out_packed, d_out_linear=None, options=None, stream=None, want_stats=False):
        """d_out_*: integer device addresses (e.g. torch tensor .data_ptr()); stream: hipStream_t handle or None."""
        stats = abi.Stats() if want_stats else None
        _check(self._L.mi355rt_context_render(self._h, C.byref(options) if options is not None else None,
                                            C.c_void_p(d_out_packed), C.c_void_p(d_out_linear) if d_out_linear else None,
                                            C.c_void_p(stream) if stream else None,
                                            C.byref(stats) if want_stats else None), "mi355rt_context_render", self._L)
        return stats

    def render_progressive(self, sample_begin, sample_end, d_accum, d_out_packed, d_out_linear=None, options=None, stream=None, want_stats=False):
        """Samples [sample_begin, sample_end) added to the running sums in d_accum (float4 per selected pixel, device address)."""
        stats = abi.Stats() if want_stats else None
        _check(self._L.mi355rt_context_render_progressive(self._h, C.byref(options) if options is not None else None,
                                                        int(sample_begin), int(sample_end), C.c_void_p(d_accum), C.c_void_p(d_out_packed),
                                                        C.c_void_p(d_out_linear) if d_out_linear else None,
                                                        C.c_void_p(stream) if stream else None,
                                                        C.byref(stats) if stats is not None else None), "mi355rt_context_render_progressive", self._L)
        return stats

    def kernel_variant(self):
        """Diagnostic: the counter-mode kernel chosen for the resident scene (0 lockstep, 1 lockstep+mesh, 2 state machine, 3 lockstep simple)."""
        v = C.c_uint32()
        _check(self._L.mi355rt_debug_kernel_variant(self._h, C.byref(v)), "mi355rt_debug_kernel_variant", self._L)
        return v.value

    def row_tables(self):
        """Diagnostic: (natural, processing, out_row) row tables of the last render on this context and the per-image-row cost (rays per path
        of set_scene's probe; empty when the rows are processed in image order)."""
        n, nc = C.c_uint32(), C.c_uint32()
        f = self._L.mi355rt_debug_read_row_tables
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        _check(f(self._h, None, 0, C.byref(n), None, 0, C.byref(nc)), "mi355rt_debug_read_row_tables", self._L)
        t = np.zeros(3 * n.value, np.uint32); cost = np.zeros(nc.value, np.float32)
        _check(f(self._h, t.ctypes.data, t.size, C.byref(n), cost.ctypes.data, cost.size, C.byref(nc)), "mi355rt_debug_read_row_tables", self._L)
        return t[:n.value], t[n.value:2 * n.value], t[2 * n.value:], cost

    def set_timing(self, enable=True):
        _check(self._L.mi355rt_context_set_timing(self._h, 1 if enable else 0), "mi355rt_context_set_timing", self._L)

    def read_timing(self):
        """(render_kernel_ms, resolve_kernel_ms, launches) summed since the last read; call after syncing the stream."""
        a, b, n = C.c_double(), C.c_double(), C.c_uint32()
        _check(self._L.mi355rt_context_read_timing(self._h, C.byref(a), C.byref(b), C.byref(n)), "mi355rt_context_read_timing", self._L)
        return a.value, b.value, n.value

    def close(self):
        if self._h:
            self._L.mi355rt_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
