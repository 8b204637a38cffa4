#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X render loop (contract in the task statement / DESIGN.md).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE full render of the BASELINE.json config[1] workload -- cornell-box scene.json at
800x600, 256 spp, max_bounces 30 -- through the C ABI with the scene already resident in HBM:
path-tracing kernel + ordered resolve kernel on every rank's row strips, then (N > 1) one RCCL gather of
the packed rows to rank 0 and the de-interleave.  The image is fixed, so scaling is STRONG.
value = width*height*spp*K / max-over-ranks wall time of the K steps, in Msamples/s.

Extra objects in the JSON line:
  roofline     dominant kernel (k_render_ctr): algorithmic bytes per launch (SURVEY.md 8d formula with the
               measured rays/sample) / mean launch duration from HIP events recorded on the launch
               stream over the timed region.  The scene records live in SGPRs/L2, so `frac` can exceed 1
               against HBM; the truly binding unit is the f32 VALU (DESIGN.md "Roofline").
  cpu_baseline the C++ oracle ("port" of the reference's rayon loop, reference RNG stream) timed on the
               host cores on a bounded row sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1..3]; configs[0] is the CPU plumbing case, configs[4] the 8-GPU case
    "cornell-box-800x600x256-d30": ("data/scenes/tungsten/cornell-box/scene.json", 800, 600, 256, 30, False),
    "teapot-800x600x256-d64": ("data/scenes/tungsten/teapot/scene.json", 800, 600, 256, 64, True),
    "veach-mis-1280x720x1024-d16": ("data/scenes/tungsten/veach-mis/scene.json", 1280, 720, 1024, 16, False),
    "semesterbild-800x600x256-d30": ("data/scenes/semesterbild.json", 800, 600, 256, 30, False),
    "semesterbild-1920x1080x4096-d30": ("data/scenes/semesterbild.json", 1920, 1080, 4096, 30, False),
}
REC_BYTES = {0: 16, 1: 24, 2: 64, 3: 128, 4: 128}     # sphere, plane, quad, cube, mesh header
HBM_PEAK_GBS = 8000.0                                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_SIMDS, VALU_CLOCK_HZ, VALU_CYCLES_PER_WAVE64_INST = 1024, 2.4e9, 2     # 256 CUs x 4 SIMDs; 157.3 TFLOP/s f32 = 1024 x 32 lanes x 2 x 2.4 GHz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cornell-box-800x600x256-d30", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample; 0 disables it")
    ap.add_argument("--save-png", default="")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    import torch.distributed as dist
    abi = importlib.import_module("raytracer-rust_amd.abi")
    host = importlib.import_module("raytracer-rust_amd.host")
    device = importlib.import_module("raytracer-rust_amd.device")
    rtdist = importlib.import_module("raytracer-rust_amd.distributed")

    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    # Rehearsal on a one-GPU box (MI355RT_BENCH_REHEARSE=1): every rank shares cuda:0 and the gather goes through
    # gloo on host copies, because RCCL refuses two ranks on one GPU.  It checks the multi-rank plumbing only;
    # its numbers mean nothing and the JSON says so.
    rehearse = os.environ.get("MI355RT_BENCH_REHEARSE") == "1" and world > 1
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    path, W, H, spp, depth, skip_unknown = WORKLOADS[args.workload]
    scene = host.LoadedScene(os.path.join(ROOT, path), W, H, spp, depth, skip_unknown_primitives=skip_unknown)   # product loader (C++)
    ctx = device.Context(local_rank)
    ctx.set_scene(scene, scene.camera, scene.settings)                 # scene resident in HBM from here on
    plan = rtdist.make_plan(H, W, world)
    opt = plan.options_for(abi, rank)
    n_local_rows = len(plan.rows[rank])
    local = torch.zeros((plan.max_rows, W), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    image = None

    def step():
        nonlocal image
        ctx.render(local.data_ptr(), None, opt, stream)               # enqueue only: no host sync inside
        image = rtdist.gather_image(local.cpu() if rehearse else local, plan, rank)

    for _ in range(args.warmup):
        step()
    sync_all()
    ctx.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    k_render_ms, k_resolve_ms, launches = ctx.read_timing()
    ctx.set_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # one extra (untimed) step with counters for rays/sample
    st = ctx.render(local.data_ptr(), None, opt, stream, want_stats=True)
    local_samples = n_local_rows * W * spp
    rays_per_sample = st.rays / max(st.samples, 1)

    total_samples = W * H * spp
    value = total_samples * args.steps / elapsed / 1e6
    result = None
    if rank == 0:
        sc = scene.c
        rec = sum(REC_BYTES[sc.primitives[i].kind] for i in range(sc.n_primitives))
        nodes_per_ray = tris_per_ray = 0.0
        cpu_baseline = None
        try:                                   # the CPU leg must never cost the GPU measurement its JSON line
            if world == 1 and args.cpu_seconds > 0:
                cpu_baseline, nodes_per_ray, tris_per_ray = run_cpu_baseline(abi, scene, W, H, spp, args.cpu_seconds)
            elif sc.n_meshes:
                _, nodes_per_ray, tris_per_ray = run_cpu_baseline(abi, scene, W, H, spp, 0.5)
        except Exception as e:                 # e.g. no g++ on the box
            cpu_baseline = {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
        bytes_per_sample = rays_per_sample * (rec + 48 + nodes_per_ray * 32 + tris_per_ray * 48) + 16.0 / spp
        ms_per_launch = k_render_ms / max(launches, 1)
        achieved = bytes_per_sample * local_samples / (ms_per_launch * 1e-3) / 1e9 if ms_per_launch > 0 else 0.0
        traffic = None
        valu = None                              # the unit that really binds: f32 VALU issue slots (PMC instruction count / live kernel time)
        pmc = os.path.join(ROOT, "profiles", "pmc_hbm_bytes.json")
        if os.path.exists(pmc):
            try:
                rec_pmc = json.load(open(pmc)).get(args.workload, {})
                traffic = rec_pmc.get("k_render_ctr_hbm_bytes_per_launch")
                if traffic is not None:          # measured on the whole image in one launch; a rank renders local_samples of it
                    traffic = int(traffic * local_samples / total_samples)
                insts = rec_pmc.get("k_render_ctr_valu_insts_per_launch")
                if insts is not None and ms_per_launch > 0:
                    peak = VALU_SIMDS * VALU_CLOCK_HZ / VALU_CYCLES_PER_WAVE64_INST
                    rate = insts * local_samples / total_samples / (ms_per_launch * 1e-3)
                    valu = {"wave_insts_per_launch": int(insts * local_samples / total_samples), "achieved_ginst_s": round(rate / 1e9, 1),
                            "peak_ginst_s": round(peak / 1e9, 1), "frac": round(rate / peak, 4),
                            "lane_utilisation": rec_pmc.get("k_render_ctr_valu_lane_utilisation"),
                            "note": "SQ_INSTS_VALU (PMC, profiles/) / live kernel time vs 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction"}
            except Exception:
                traffic = None
        result = {
            "metric": "Msamples/s (pixels x spp / s) at 800x600x256spp; 1/2/4/8-GPU scaling",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (reference's own cornell-box scene.json; no external data)",
            "config": {"workload": args.workload, "scene": path, "width": W, "height": H, "spp": spp, "max_bounces": depth,
                       "rng": "ctr (Philox4x32-10 per ray)", "parallelism": f"row strips of {plan.strip_rows} dealt round-robin over {world} GPU(s)"
                       + (", RCCL gather to rank 0" if world > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": "k_render_ctr", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_sample": round(bytes_per_sample, 1), "rays_per_sample": round(rays_per_sample, 4),
                         "kernel_ms_per_launch": round(ms_per_launch, 4), "resolve_ms_per_launch": round(k_resolve_ms / max(launches, 1), 4),
                         "launches_timed": launches, "samples_per_launch": local_samples,
                         "valu": valu,
                         "note": "scene records are SGPR/L2 resident, so algorithmic bytes never reach HBM; the binding unit is the f32 VALU (see valu)"},
            "cpu_baseline": cpu_baseline,
            **({"rehearsal": "all ranks on cuda:0 over gloo -- plumbing check only, NOT a measurement"} if rehearse else {}),
            "kernel": {"vgprs": st.kernel_vgprs, "grid_blocks": st.grid_blocks, "block_threads": st.block_threads, "bands": st.bands},
        }
        if image is not None:
            result["image_checksum"] = int(image.to(torch.int64).sum().item())      # identical for every N (RNG keyed by absolute row)
        if args.save_png and image is not None:
            import numpy as np
            host.write_png(args.save_png, image.cpu().numpy().astype(np.uint32), W, H)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    return result


def run_cpu_baseline(abi, scene, W, H, spp, target_seconds):
    """Times the CPU oracle (reference RNG stream, tail-first folding, all host cores given to this
    job) on every k-th row of the same workload; rows are independent and their cost is additive."""
    import oracle
    oracle.build()
    cores = min(len(os.sched_getaffinity(0)), 16) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # Rows are the unit of parallelism (one row per thread at a time, like rayon's par_chunks_mut), so the sample
    # is always a multiple of `cores` rows, spread evenly over the image.  Pilot: one row per core.
    def every(n_rows):
        n_rows = max(cores, min(H, (n_rows // cores) * cores))
        return abi.Options.make(rng_mode=abi.RNG_REF, strip_rows=1, n_parts=max(1, -(-H // n_rows)), part=0)   # ceil: never more than n_rows rows
    _, _, c0 = oracle.render(scene, scene.camera, scene.settings, every(cores), threads=cores, want_linear=False)
    rate = c0.samples / max(c0.seconds, 1e-9)
    opt = every(int(target_seconds * rate / (W * spp)))
    parts = opt.n_parts
    _, _, c = oracle.render(scene, scene.camera, scene.settings, opt, threads=cores, want_linear=False)
    rows = len(abi.rows_selected(H, opt))
    base = {"value": round(c.samples / c.seconds / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"every {parts}th row ({rows} of {H} rows, {c.samples} samples, {c.seconds:.1f} s) of the same scene/resolution/spp/depth; "
                      "C++ restatement of the reference's rayon path with its ChaCha12 row streams",
            "rays_per_sample": round(c.rays / max(c.samples, 1), 4)}
    return base, c.bvh_nodes / max(c.rays, 1), c.tri_tests / max(c.rays, 1)


if __name__ == "__main__":
    main()
