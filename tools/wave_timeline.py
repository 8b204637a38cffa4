"""Diagnostic: per-wave start/end times of k_render_ctr (stamps build) -> ramp-up and drain tail."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg
build = pkg("build")
os.environ["MI355RT_DEVICE_SO"] = build.build_device_variant("stamps", ["MI355RT_STAMPS"])
import numpy as np, torch
torch.zeros(1, device="cuda")
host, device, abi = pkg("host"), pkg("device"), pkg("abi")
SCENES = {"cornell": ("data/scenes/tungsten/cornell-box/scene.json", 30, False), "semesterbild": ("data/scenes/semesterbild.json", 30, False),
          "teapot": ("data/scenes/tungsten/teapot/scene.json", 64, True)}
path, depth, skip = SCENES[sys.argv[1] if len(sys.argv) > 1 else "cornell"]
sc = host.LoadedScene(os.path.join(ROOT, path), 800, 600, 256, depth, skip_unknown_primitives=skip)
ctx = device.Context(0); ctx.set_knob("wave_times", 1)
if os.environ.get("ROW_ORDER"): ctx.set_knob("row_order", int(os.environ["ROW_ORDER"]))
ctx.set_scene(sc, sc.camera, sc.settings)
out = torch.zeros(800 * 600, dtype=torch.int32, device="cuda")
for parts in (1, 8):
    opt = abi.Options.make(strip_rows=5, n_parts=parts, part=0)
    for _ in range(3): st = ctx.render(out.data_ptr(), None, opt, None, want_stats=True)
    buf = np.zeros(6 * 16384, np.uint64); n = C.c_uint32()
    device.lib().mi355rt_debug_read_wave_times.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_uint32, C.POINTER(C.c_uint32)]
    assert device.lib().mi355rt_debug_read_wave_times(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), 16384, C.byref(n)) == 0
    w = buf[:6 * n.value].reshape(-1, 6).astype(np.float64)
    t0 = w[:, 0].min(); start = (w[:, 0] - t0) / 100.0; end = (w[:, 1] - t0) / 100.0     # microseconds (100 MHz)
    if parts == 1:
        nat, proc, outr, cost = ctx.row_tables()
        print("   row cost (rays per path) by image row, every 25th:", " ".join(f"{c:.2f}" for c in cost[::25]) if len(cost) else "none (image order)")
        print("   processing order, first 12 rows of shard 0:", list(proc[:12]), "... last 6 of shard 0:", list(proc[len(proc) // 8 - 6:len(proc) // 8]))
    print(f"parts={parts}: kernel {st.render_kernel_ms:.3f} ms, waves {n.value}, paths/wave mean {w[:,2].mean():.0f} min {w[:,2].min():.0f} max {w[:,2].max():.0f}")
    print("   wave start  us: p50 %.0f p99 %.0f max %.0f" % tuple(np.percentile(start, [50, 99, 100])))
    print("   wave end    us: p1 %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % tuple(np.percentile(end, [1, 10, 50, 90, 99, 100])))
    busy = (w[:, 1] - w[:, 0]).sum() / 100.0
    print("   sum(wave lifetimes) / (waves * makespan) = %.3f" % (busy / (n.value * end.max())))
    dry = (w[:, 3] - t0) / 100.0
    print("   work ran dry us: p1 %.0f p50 %.0f p99 %.0f max %.0f" % tuple(np.percentile(dry, [1, 50, 99, 100])))
    print("   drain (end - dry) us: p10 %.0f p50 %.0f p90 %.0f max %.0f | iterations p50 %.0f p90 %.0f max %.0f | live lanes at dry p50 %.0f" % (
        *np.percentile(end - dry, [10, 50, 90, 100]), *np.percentile(w[:, 4], [50, 90, 100]), np.percentile(w[:, 5], 50)))
    d = end - dry; it = np.maximum(w[:, 4], 1)
    print("   us per drain iteration: p10 %.1f p50 %.1f p90 %.1f" % tuple(np.percentile(d / it, [10, 50, 90])))
    # how much of the chip is still at work, and when: share of the waves alive at a few instants before the end
    mk = end.max()
    print("   waves still running at makespan - x: " + "  ".join(f"-{x} us: {(end > mk - x).mean():.3f}" for x in (100, 250, 500, 1000, 1500, 2000, 3000)))
    print("   wave-time lost to the tail (waves x (makespan - end)) / (waves x makespan) = %.3f" % ((mk - end).sum() / (n.value * mk)))
