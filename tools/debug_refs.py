import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import pkg, load_for_both
import oracle; oracle.build()
host, device, abi = pkg("host"), pkg("device"), pkg("abi")
R = device.refs()
for name in ("teapot", "semesterbild"):
    sc = load_for_both(name, oracle, host, width=96, height=64, spp=6, max_depth=12)
    for rep in range(3):
        for library, kv in ((None, {"kernel": 1}), (None, {}), (R, {"kernel": 1}), (R, {"kernel": 7}),
                            (R, {"kernel": 2, "trav_min": 1}), (R, {"kernel": 2}), (R, {"kernel": 2, "inline_steps": 0})):
            L = library or device.lib()
            device.clear_knobs(L)
            for k, v in kv.items(): device.set_knob(k, v, L)
            t = time.time()
            try:
                gp, gl, st = device.render(sc, sc.camera, sc.settings, abi.Options.make(), library=library)
                print(name, rep, "refs" if library else "prod", kv, "ok", round(time.time() - t, 3), st.rays, int(gp.astype("int64").sum()), flush=True)
            except Exception as e:
                print(name, rep, "refs" if library else "prod", kv, "FAILED", round(time.time() - t, 3), str(e)[:150], flush=True)
            device.clear_knobs(L)
