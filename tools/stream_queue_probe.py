#!/usr/bin/env python3
"""Which HIP streams run concurrently?  Frames in flight (mi355rt_context_set_share) only overlap when their streams sit on different hardware
queues, and HIP multiplexes streams onto GPU_MAX_HW_QUEUES (4) queues in a way the caller does not see.  This probe creates a pool of torch streams
and times the same 4-frames-in-flight loop (1/8 image, share 4) on different subsets of them; a subset whose streams share a queue serialises.
usage: python tools/stream_queue_probe.py [workload] [n_streams]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import pkg
import bench
abi, host, device, rtdist = pkg("abi"), pkg("host"), pkg("device"), pkg("distributed")
wl = sys.argv[1] if len(sys.argv) > 1 else "cornell-box-800x600x256-d30"
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
path, W, H, spp, depth, skip = bench.WORKLOADS[wl]
scene = host.LoadedScene(os.path.join(ROOT, path), W, H, spp, depth, skip_unknown_primitives=skip)
plan = rtdist.make_plan(H, W, 8); o = plan.options_for(abi, 0)
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default)"), "| streams:", [hex(s.cuda_stream)[-6:] for s in streams], flush=True)
ctxs = []
for i in range(4):
    c = device.Context(0); c.set_share(4); c.set_scene(scene, scene.camera, scene.settings)
    ctxs.append((c, torch.zeros((plan.max_rows, W), dtype=torch.int32, device=dev)))
def run(sub, n=24):
    for (c, out), si in zip(ctxs, sub): c.render(out.data_ptr(), None, o, streams[si].cuda_stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        c, out = ctxs[i % 4]; c.render(out.data_ptr(), None, o, streams[sub[i % 4]].cuda_stream)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
subsets = [[0, 1, 2, 3], [1, 2, 3, 4], [2, 3, 4, 5], [4, 5, 6, 7], [0, 2, 4, 6], [0, 4, 8, 1], [0, 4, 8, 11][:4], [0, 0, 0, 0], [0, 1, 0, 1], [3, 5, 8, 10], [7, 8, 9, 10], [0, 1, 2, 3]]
for sub in subsets:
    sub = [s % NS for s in sub]
    print(f"streams {sub}: {run(sub):.3f} ms per 1/8 frame", flush=True)
# pairwise: which streams overlap with stream 0?  two frames in flight on (0, k), share 2
for (c, _) in ctxs: c.set_share(2)
def run2(a, b, n=16):
    pair = [a, b]
    for k in range(2): ctxs[k][0].render(ctxs[k][1].data_ptr(), None, o, streams[pair[k]].cuda_stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): ctxs[i % 2][0].render(ctxs[i % 2][1].data_ptr(), None, o, streams[pair[i % 2]].cuda_stream)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("two frames in flight (share 2) on streams (0, k):", " ".join(f"{k}:{run2(0, k):.2f}" for k in range(1, NS)), flush=True)
print("                                      on (1, k):", " ".join(f"{k}:{run2(1, k):.2f}" for k in range(2, NS)), flush=True)
