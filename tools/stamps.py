"""Diagnostic: shares of wave time per section of k_render_ctr (s_memtime stamps, -DMI355RT_STAMPS build).
Never quote this build's run time; read its SHARES (cdna_hip_programming.md section 7)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg
build = pkg("build")
os.environ["MI355RT_DEVICE_SO"] = build.build_device_variant("stamps", ["MI355RT_STAMPS"])
import numpy as np
import torch
torch.zeros(1, device="cuda")   # initialise torch's HIP context before the library's (as bench.py does)
host, device, abi = pkg("host"), pkg("device"), pkg("abi")
NAMES = ["BVH rounds (state machine)", "top-level list / hit_scene", "classify + finish + deal", "cooperative rejection + finish", "loop tail / vote", "philox + camera + material switch"]
for name, path, W, H, spp, depth, skip in [("cornell", "data/scenes/tungsten/cornell-box/scene.json", 800, 600, 64, 30, False),
                                           ("semesterbild", "data/scenes/semesterbild.json", 800, 600, 256, 30, False),
                                           ("veach", "data/scenes/tungsten/veach-mis/scene.json", 1280, 720, 32, 16, False),
                                           ("teapot", "data/scenes/tungsten/teapot/scene.json", 800, 600, 256, 30, True)]:
    sc = host.LoadedScene(os.path.join(ROOT, path), W, H, spp, depth, skip_unknown_primitives=skip)
    ctx = device.Context(0); ctx.set_scene(sc, sc.camera, sc.settings)
    out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    st = ctx.render(out.data_ptr(), None, abi.Options.make(), None, want_stats=True)
    raw = (C.c_ulonglong * 40)()
    assert device.lib().mi355rt_debug_read_counters(ctx._h, raw) == 0
    acc = np.array(list(raw)[2:8], dtype=np.float64)
    print(f"{name}: kernel {st.render_kernel_ms:.2f} ms (stamped build), rays/sample {st.rays / st.samples:.2f}")
    for n, v in zip(NAMES, acc):
        if v: print(f"   {n:38s} {100 * v / acc.sum():6.2f} %")
    cl = list(raw)[24:34]
    if sum(cl) and ctx.kernel_variant() in (0, 3, 9, 14):
        iters = cl[8]                   # wave iterations with at least one lane to shade
        print(f"   shading step: {iters / (st.rays / 64.0):.2f} wave iterations per 64 rays, {cl[9] / max(iters, 1):.1f} lanes with a ray to produce")
        for i, n in enumerate(["rough conductor branch", "Lambert-style bounce (cooperative unit ball)", "metal / dielectric branch", "camera ray (fresh path)"]):
            e, l = cl[2 * i], cl[2 * i + 1]
            if e: print(f"   {n:46s} runs in {100 * e / max(iters, 1):5.1f} % of the wave iterations at a mean of {l / e:5.1f} lanes")
    blk = list(raw)[8:24]
    if sum(blk):
        rays = st.rays
        wf = ctx.kernel_variant() in (7, 8, 10, 11, 12, 13)
        names = (["WALK passes", "TOP1 passes", "TOP passes that park walks (lanes = walks parked)", "SHADE+TOP0 passes", "WALK box-test steps (lanes stepping)",
                  "WALK leaf phases (lanes with a leaf)", "WALK passes (lanes = walks finished)", "inline box-test steps in TOP"] if wf
                 else ["inner-node steps", "leaf phases", "TOP passes", "SHADE passes"])
        for i, n in enumerate(names):
            e, l = blk[2 * i], blk[2 * i + 1]
            if e: print(f"   {n:52s} executions per 64 rays {e / (rays / 64):7.2f}   mean lanes {l / e:5.1f}")
    ctx.close()
