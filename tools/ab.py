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
ENVS = {}
if os.environ.get("AB_ENVS"):        # {"name": {"lib": "<variant>", "knobs": {"kernel": 7, ...}}}: same library, different diagnostic knobs (mi355rt_debug_set_knob)
    import json; ENVS = json.loads(os.environ["AB_ENVS"])
WL = {"cornell": ("data/scenes/tungsten/cornell-box/scene.json", 800, 600, 256, 30, False),
      "semesterbild": ("data/scenes/semesterbild.json", 800, 600, 64, 30, False),
      "veach": ("data/scenes/tungsten/veach-mis/scene.json", 1280, 720, 64, 16, False),
      "teapot": ("data/scenes/tungsten/teapot/scene.json", 800, 600, 64, 30, True)}
if os.environ.get("AB_SPP"):          # samples per pixel for every workload (the mesh scenes default to 64)
    WL = {k: (v[0], v[1], v[2], int(os.environ["AB_SPP"]), v[4], v[5]) for k, v in WL.items()}
names = sys.argv[1:] or ["cornell"]
import torch; torch.zeros(1, device="cuda")
libs = {}
PREBUILT = dict(kv.split("=", 1) for kv in os.environ.get("AB_PREBUILT", "").split(",") if kv)   # name=path of an already built .so
for v in PREBUILT: VARIANTS[v] = ((), ())
PARTS = int(os.environ.get("AB_PARTS", "1"))          # render only part 0 of PARTS row strips (multi-GPU sized launch)
for v, (defs, flags) in VARIANTS.items():
    so = PREBUILT[v] if v in PREBUILT else build.build_device_variant("ab_" + v, defs, flags=flags)
    L = C.CDLL(so)
    L.mi355rt_context_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.mi355rt_context_set_scene.argtypes = [C.c_void_p, C.POINTER(abi.Scene), C.POINTER(abi.Camera), C.POINTER(abi.Settings)]
    L.mi355rt_context_render.argtypes = [C.c_void_p, C.POINTER(abi.Options), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(abi.Stats)]
    L.mi355rt_context_destroy.argtypes = [C.c_void_p]; L.mi355rt_last_error.restype = C.c_char_p
    L.mi355rt_context_set_timing.argtypes = [C.c_void_p, C.c_int]
    L.mi355rt_context_read_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.mi355rt_debug_set_knob.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    libs[v] = L
ALL_LIBS = dict(libs)
for wl in names:
    libs = dict(ALL_LIBS)
    path, W, H, spp, depth, skip = WL[wl]
    sc = host.LoadedScene(os.path.join(ROOT, path), W, H, spp, depth, skip_unknown_primitives=skip)
    out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    ctxs, times, sums = {}, {v: [] for v in libs}, {}
    if ENVS:
        base_libs = dict(ALL_LIBS); libs = {}
        for n, spec in ENVS.items(): libs[n] = base_libs[spec["lib"]]
        times = {v: [] for v in libs}
    for v, L in libs.items():
        h = C.c_void_p(); assert L.mi355rt_context_create(0, C.byref(h)) == 0, L.mi355rt_last_error()
        if ENVS:
            for k, val in ENVS[v].get("knobs", {}).items(): assert L.mi355rt_debug_set_knob(h, k.encode(), int(val)) == 0, L.mi355rt_last_error()
        assert L.mi355rt_context_set_scene(h, C.byref(sc.c), C.byref(sc.camera), C.byref(sc.settings)) == 0, L.mi355rt_last_error()
        ctxs[v] = h
    opt = abi.Options.make(strip_rows=5, n_parts=PARTS, part=0)
    for rnd in range(8):
        for v, L in libs.items():
            st = abi.Stats()
            assert L.mi355rt_context_render(ctxs[v], C.byref(opt), C.c_void_p(out.data_ptr()), None, None, C.byref(st)) == 0, L.mi355rt_last_error()
            if rnd: times[v].append(st.render_kernel_ms)
            sums[v] = (int(out.to(torch.int64).sum().item()), st.kernel_vgprs, st.resolve_kernel_ms)
    # pipelined: K renders enqueued back to back, HIP-event kernel times read after one sync (what bench.py does)
    import time
    pipe = {}
    for v, L in libs.items():
        K = 12
        for _ in range(2): L.mi355rt_context_render(ctxs[v], C.byref(opt), C.c_void_p(out.data_ptr()), None, None, None)
        torch.cuda.synchronize(); L.mi355rt_context_set_timing(ctxs[v], 1); t0 = time.perf_counter()
        for _ in range(K): L.mi355rt_context_render(ctxs[v], C.byref(opt), C.c_void_p(out.data_ptr()), None, None, None)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / K * 1e3
        a, b, n = C.c_double(), C.c_double(), C.c_uint32(); L.mi355rt_context_read_timing(ctxs[v], C.byref(a), C.byref(b), C.byref(n)); L.mi355rt_context_set_timing(ctxs[v], 0)
        pipe[v] = (wall, a.value / max(n.value, 1))
    for v in libs:
        print(f"{wl:13s} {v:14s} pipelined wall/step {pipe[v][0]:8.3f} ms  kernel {pipe[v][1]:8.3f} ms | synced: render median {statistics.median(times[v]):8.3f} ms  min {min(times[v]):8.3f}  resolve {sums[v][2]:.3f} ms  vgprs {sums[v][1]}  checksum {sums[v][0]}", flush=True)
    for v, L in libs.items(): L.mi355rt_context_destroy(ctxs[v])
