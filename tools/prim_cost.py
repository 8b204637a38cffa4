"""Ad-hoc: how much of the cornell kernel time each primitive type accounts for (scene variants, same camera)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg
import torch; torch.zeros(1, device="cuda")
abi, host, device = pkg("abi"), pkg("host"), pkg("device")
sc = host.LoadedScene(os.path.join(ROOT, "data/scenes/tungsten/cornell-box/scene.json"), 800, 600, 64, 30)
prims = [sc.c.primitives[i] for i in range(sc.c.n_primitives)]
out = torch.zeros(800 * 600, dtype=torch.int32, device="cuda")
def run(name, keep):
    s2 = abi.Scene(); C.memmove(C.byref(s2), C.byref(sc.c), C.sizeof(abi.Scene))
    arr = (abi.Primitive * max(len(keep), 1))(*[prims[i] for i in keep])
    s2.primitives, s2.n_primitives = arr, len(keep)
    ctx = device.Context(0); ctx.set_scene(s2, sc.camera, sc.settings)
    ts = []
    for _ in range(4):
        st = ctx.render(out.data_ptr(), None, abi.Options.make(), None, want_stats=True); ts.append(st.render_kernel_ms)
    per_ray = min(ts) * 1e6 / st.rays     # ns of kernel time per ray (whole chip)
    print(f"{name:28s} prims {len(keep)}  kernel {min(ts):7.3f} ms  rays/sample {st.rays / st.samples:5.2f}  ps/ray {per_ray * 1e3:7.2f}")
    ctx.close()
run("full (6 quads + 2 cubes)", list(range(8)))
run("quads only (6)", [0, 1, 2, 3, 4, 7])
run("walls without light (5)", [0, 1, 2, 3, 4])
run("cubes only (2)", [5, 6])
run("floor quad only (1)", [0])
run("empty", [])
