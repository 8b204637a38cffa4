"""A/B of the counter-mode kernel variants / trav_min in one process, on the reference build (-DMI355RT_REFS: the product sources plus
the retired state-machine kernel); variants are chosen with the diagnostic knobs of mi355rt_debug_set_knob."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg
abi, host, device = pkg("abi"), pkg("host"), pkg("device")
import torch; torch.zeros(1, device="cuda")
WL = {"cornell": ("data/scenes/tungsten/cornell-box/scene.json", 800, 600, 256, 30, False),
      "semesterbild": ("data/scenes/semesterbild.json", 800, 600, 64, 30, False),
      "veach": ("data/scenes/tungsten/veach-mis/scene.json", 1280, 720, 64, 16, False),
      "teapot": ("data/scenes/tungsten/teapot/scene.json", 800, 600, 64, 30, True)}
SETTINGS = [("lockstep", {"kernel": "1"}), ("wavefront", {"kernel": "7"})] + [(f"sm{t}", {"kernel": "2", "trav_min": str(t)}) for t in (16, 24, 32)] \
         + [("sm24i1", {"kernel": "2", "trav_min": "24", "inline_steps": "1"})]
ENV_KEYS = ("kernel", "trav_min", "inline_steps")
REFS = device.refs()
if os.environ.get("AB_KERNEL_SETTINGS"):     # "name=KERNEL:TRAV_MIN[:INLINE_STEPS[:WALKERS[:PATIENCE]]],..."
    SETTINGS = [(kv.split("=")[0], dict(zip(ENV_KEYS, kv.split("=")[1].split(":")))) for kv in os.environ["AB_KERNEL_SETTINGS"].split(",")]
for wl in sys.argv[1:] or ["semesterbild", "teapot"]:
    path, W, H, spp, depth, skip = WL[wl]
    spp = int(os.environ.get("AB_SPP", spp))
    sc = host.LoadedScene(os.path.join(ROOT, path), W, H, spp, depth, skip_unknown_primitives=skip)
    out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    ctxs = {}
    for name, env in SETTINGS:
        c = device.Context(0, REFS)
        for k, val in env.items(): c.set_knob(k, int(val))
        c.set_scene(sc, sc.camera, sc.settings); ctxs[name] = c
    times = {n: [] for n in ctxs}; info = {}
    for rnd in range(5):
        for n, c in ctxs.items():
            st = c.render(out.data_ptr(), None, abi.Options.make(), None, want_stats=True)
            if rnd: times[n].append(st.render_kernel_ms)
            info[n] = (int(out.to(torch.int64).sum().item()), st.kernel_vgprs, st.rays)
    for n in ctxs:
        print(f"{wl:13s} {n:10s} median {statistics.median(times[n]):8.3f} ms  min {min(times[n]):8.3f}  vgprs {info[n][1]}  rays {info[n][2]}  checksum {info[n][0]}", flush=True)
    for c in ctxs.values(): c.close()
