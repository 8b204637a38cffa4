#!/usr/bin/env python3
"""What ONE rank of an 8-GPU run does per frame, on a one-GPU box, INCLUDING the collective: strip part 0 of 8 of the image is rendered by F frame slots
(own context + probed stream, mi355rt_context_set_share) and every frame ends with an RCCL all_gather_into_tensor (world size 1: the call, the process
group's own collective stream and its event hand-shakes with the frame's stream are the real ones; the payload is what one rank contributes) and the
de-interleaving index_select -- the step of bench.py's rank path.  The question it answers: does the collective's stream, which torch puts on a hardware
queue of its own choosing, disturb the frames in flight?  (bench.py --tail-parts measures the same frames without a collective.)
usage: python tools/rank_rehearsal.py [workload] [FxD,FxD,...]"""
import os, sys, time, datetime
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29511")): os.environ.setdefault(k, v)
import torch, torch.distributed as dist
from conftest import pkg
import bench
abi, host, device, rtdist = pkg("abi"), pkg("host"), pkg("device"), pkg("distributed")
wl = sys.argv[1] if len(sys.argv) > 1 else "cornell-box-800x600x256-d30"
spec = sys.argv[2] if len(sys.argv) > 2 else "1x1,2x2,3x3,4x4,4x2"
PARTS = 8
path, W, H, spp, depth, skip = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(seconds=60))
scene = host.LoadedScene(os.path.join(ROOT, path), W, H, spp, depth, skip_unknown_primitives=skip)
plan = rtdist.make_plan(H, W, PARTS); o = plan.options_for(abi, 0)
perm = torch.arange(plan.max_rows, device=dev)
for item in spec.split(","):
    F, D = (int(t) for t in item.split("x"))
    streams, info = bench.concurrent_streams(torch, dev, F)
    slots = []
    for k in range(F):
        c = device.Context(0); c.set_share(D); c.set_scene(scene, scene.camera, scene.settings)
        slots.append((c, streams[k], torch.zeros((plan.max_rows, W), dtype=torch.int32, device=dev)))
    def step(i, collective):
        c, s, local = slots[i % F]
        with torch.cuda.stream(s):
            c.render(local.data_ptr(), None, o, s.cuda_stream)
            if collective:
                stacked = torch.empty((plan.max_rows, W), dtype=torch.int32, device=dev)
                dist.all_gather_into_tensor(stacked, local)
                return stacked.index_select(0, perm)
    res = {}
    for collective in (False, True):
        for i in range(2 * F): step(i, collective)
        torch.cuda.synchronize(); n = F * 10; t0 = time.perf_counter()
        for i in range(n): img = step(i, collective)
        torch.cuda.synchronize(); res[collective] = (time.perf_counter() - t0) / n * 1e3
    for c, _, _ in slots: c.check(); c.close()
    print(f"{wl} {F} frames in flight on 1/{D} ({info['distinct']} queues): {res[False]:.3f} ms per 1/8 frame without, {res[True]:.3f} with the all_gather + index_select per frame", flush=True)
dist.destroy_process_group()
