"""A/B timing of device-library build variants in ONE process, interleaved rounds (cdna guide rule 24).
usage: python tools/ab.py [workload] -- variants are defined in VARIANTS below."""
import ctypes as C, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg
build, abi, host = pkg("build"), pkg("abi"), pkg("host")
VARIANTS = {   # name: (defines, flags)
    "base": ((), ()),
    "noslp": ((), ("-fno-slp-vectorize",)),
}
if os.environ.get("AB_VARIANTS"):
    import json; VARIANTS = {k: (tuple(v[0]), tuple(v[1])) for k, v in json.loads(os.environ["AB_VARIANTS"]).items()}
WL = {"cornell": ("data/scenes/tungsten/cornell-box/scene.json", 800, 600, 256, 30, False),
      "semesterbild": ("data/scenes/semesterbild.json", 800, 600, 64, 30, False),
      "veach": ("data/scenes/tungsten/veach-mis/scene.json", 1280, 720, 64, 16, False),
      "teapot": ("data/scenes/tungsten/teapot/scene.json", 800, 600, 64, 30, True)}
names = sys.argv[1:] or ["cornell"]
import torch; torch.zeros(1, device="cuda")
libs = {}
for v, (defs, flags) in VARIANTS.items():
    so = build.build_device_variant("ab_" + v, defs, flags=flags)
    L = C.CDLL(so)
    L.mi355rt_context_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.mi355rt_context_set_scene.argtypes = [C.c_void_p, C.POINTER(abi.Scene), C.POINTER(abi.Camera), C.POINTER(abi.Settings)]
    L.mi355rt_context_render.argtypes = [C.c_void_p, C.POINTER(abi.Options), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(abi.Stats)]
    L.mi355rt_context_destroy.argtypes = [C.c_void_p]; L.mi355rt_last_error.restype = C.c_char_p
    libs[v] = L
for wl in names:
    path, W, H, spp, depth, skip = WL[wl]
    sc = host.LoadedScene(os.path.join(ROOT, path), W, H, spp, depth, skip_unknown_primitives=skip)
    out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    ctxs, times, sums = {}, {v: [] for v in libs}, {}
    for v, L in libs.items():
        h = C.c_void_p(); assert L.mi355rt_context_create(0, C.byref(h)) == 0, L.mi355rt_last_error()
        assert L.mi355rt_context_set_scene(h, C.byref(sc.c), C.byref(sc.camera), C.byref(sc.settings)) == 0, L.mi355rt_last_error()
        ctxs[v] = h
    opt = abi.Options.make()
    for rnd in range(6):
        for v, L in libs.items():
            st = abi.Stats()
            assert L.mi355rt_context_render(ctxs[v], C.byref(opt), C.c_void_p(out.data_ptr()), None, None, C.byref(st)) == 0, L.mi355rt_last_error()
            if rnd: times[v].append(st.render_kernel_ms)
            sums[v] = (int(out.to(torch.int64).sum().item()), st.kernel_vgprs, st.resolve_kernel_ms)
    for v in libs:
        print(f"{wl:13s} {v:14s} render median {statistics.median(times[v]):8.3f} ms  min {min(times[v]):8.3f}  resolve {sums[v][2]:.3f} ms  vgprs {sums[v][1]}  checksum {sums[v][0]}", flush=True)
    for v, L in libs.items(): L.mi355rt_context_destroy(ctxs[v])
