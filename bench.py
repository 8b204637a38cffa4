#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X render loop (contract in the task statement / DESIGN.md section 6).

  python bench.py --gpus N --steps K --warmup W
  N > 1, two ways in:
   * under a launcher (RANK / WORLD_SIZE in the environment): python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
     -- this process is one rank;
   * as typed, `python bench.py --gpus N`: this process never touches the GPU; it starts that launcher itself as a fresh child process
     (127.0.0.1, a free port), relays its output and exits with its code.  If the ranks cannot be started or end without a result line, it
     starts a second fresh child in --single-process mode: ONE process drives all N devices (a context + stream per device, the same strip plan,
     the gather by device-to-device copies into device 0).  The JSON line says which way ran ("launch").

A "step" is ONE full render of the workload -- by default BASELINE.json configs[1]: cornell-box scene.json at
800x600, 256 spp, max_bounces 30 -- through the C ABI with the scene already resident in HBM: path-tracing
kernel + ordered resolve kernel on every rank's row strips, then (N > 1) one RCCL all-gather of the packed rows
and the de-interleave.  The image is fixed, so scaling is STRONG.
value = width*height*spp*K / max-over-ranks wall time of the K steps, in Msamples/s.

Extra objects in the JSON line:
  roofline     the dominant kernel (k_render_ctr_*) against the unit that binds it, the f32 VALU issue rate:
               achieved = SQ_INSTS_VALU per step (PMC, profiles/pmc_counters.json, taken on THIS kernel -- the file
               carries a hash of the kernel sources; the line says "pmc": "committed (hash-matched)" when it is the hash of the library
               that ran and "stale" instead of numbers when it differs) / the kernel's time per step, measured live with HIP events on the launch stream over the
               timed region; peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction.
               `hbm` = the HBM bytes the counters saw (2*FETCH_SIZE + WRITE_SIZE, gfx950 correction) against 8 TB/s.
               `logical_bytes_model` = SURVEY.md 8d's algorithmic bytes; they are served from SGPRs / L2, never reach
               HBM, and therefore carry no fraction.
  cpu_baseline the C++ oracle ("port" of the reference's rayon loop, reference RNG stream) timed on the host cores
               on a bounded row sample of the same workload (rank 0, N = 1 only).
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
    # BASELINE.json configs[0..4]; configs[0] is the reference's own CPU-runnable "plumbing" case (tests/test_cfg1_cornell.py pins it
    # whole-image; here it is only selectable -- a 1.9 M-sample frame lasts 0.3 ms on the GPU and measures launch overhead, not the kernel)
    "cornell-box-400x300x16-d4": ("data/scenes/tungsten/cornell-box/scene.json", 400, 300, 16, 4, False),
    "cornell-box-800x600x256-d30": ("data/scenes/tungsten/cornell-box/scene.json", 800, 600, 256, 30, False),
    "teapot-800x600x256-d64": ("data/scenes/tungsten/teapot/scene.json", 800, 600, 256, 64, True),
    "veach-mis-1280x720x1024-d16": ("data/scenes/tungsten/veach-mis/scene.json", 1280, 720, 1024, 16, False),
    "semesterbild-800x600x256-d30": ("data/scenes/semesterbild.json", 800, 600, 256, 30, False),
    "semesterbild-1920x1080x4096-d30": ("data/scenes/semesterbild.json", 1920, 1080, 4096, 30, False),
}
REC_BYTES = {0: 16, 1: 24, 2: 64, 3: 128, 4: 128}     # sphere, plane, quad, cube, mesh header
HBM_PEAK_GBS = 8000.0                                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_SIMDS, VALU_CLOCK_HZ, VALU_CYCLES_PER_WAVE64_INST = 1024, 2.4e9, 2     # 256 CUs x 4 SIMDs; 157.3 TFLOP/s f32 = 1024 x 32 lanes x 2 x 2.4 GHz
KERNEL_NAMES = {0: "k_render_ctr_nomesh", 1: "k_render_ctr_mesh", 2: "k_render_ctr_sm", 3: "k_render_ctr_simple", 4: "k_render_ctr_sm_fixaabb",
                5: "(retired)", 6: "(retired)", 7: "k_render_ctr_wf", 8: "k_render_ctr_wf_fixaabb",
                9: "k_render_ctr_nospec", 10: "k_render_ctr_wf_nometal", 11: "k_render_ctr_wf_meshfree", 12: "k_render_ctr_wf_nometal_ident", 13: "k_render_ctr_wf_nometal_shallow", 14: "k_render_ctr_simple_qc"}
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_counters.json")
# What "roofline.pmc" says when the counters are used: they are NOT measured by this run.  SQ_INSTS_VALU per frame is a deterministic property of
# (kernel binary, workload) -- the same paths, the same instructions -- so it is collected once per kernel hash by tools/pmc_collect.py (separate
# rocprofv3 --pmc passes) and committed; this run contributes the live kernel time it is divided by.
PMC_OK = "committed (hash-matched)"


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """Cores this job may really use: the scheduler affinity, cut down to the cgroup CPU quota when there is one (a GPU box
    shows all of the host's cores to every tenant but gives each a share: 256 visible, 16 granted -- 256 threads on a 16-core
    quota ran the oracle at HALF the 16-thread rate).  MI355RT_CPU_THREADS overrides."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    why = f"sched affinity {n}"
    env = os.environ.get("MI355RT_CPU_THREADS")
    if env:
        return max(1, int(env)), f"MI355RT_CPU_THREADS={env}"
    quota = None
    try:                                                     # cgroup v2
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(period)
    except Exception:
        try:                                                 # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except Exception:
            pass
    if quota is not None and quota < n:
        n = max(1, int(quota + 0.5))
        why += f", cgroup quota {quota:.1f}"
    return n, why


def load_pmc(workload, kernel_hash, kernel_name=None):
    """Counters of the dominant kernel per STEP (all bands of one full render) for `workload`, or the reason there are none:
    PMC_OK when they may be used, else "absent", "unreadable", "stale" (taken on other kernel sources), "other-kernel" (taken while another kernel variant served the
    workload), "partial" (a counter pass is missing: tools/pmc_collect.py leaves such workloads out, older files may not)."""
    if not os.path.exists(PMC_FILE):
        return None, "absent"
    try:
        doc = json.load(open(PMC_FILE))
    except Exception:
        return None, "unreadable"
    if doc.get("kernel_hash") != kernel_hash:
        return None, "stale"
    rec = doc.get("workloads", {}).get(workload)
    if not rec:
        return None, "absent"
    if kernel_name is not None and rec.get("kernel") != kernel_name:
        return None, "other-kernel"
    if not all(isinstance(rec.get(k), (int, float)) for k in ("valu_wave_insts_per_step", "hbm_bytes_per_step", "valu_lane_utilisation")):
        return None, "partial"
    return rec, PMC_OK


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cornell-box-800x600x256-d30", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the cpu_baseline sample; 0 disables it")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="frames in flight (own stream + workspace each): 2 lets frame k's resolve + gather overlap frame k+1's tracing. "
                         "0 = auto: 1 on one GPU (clean per-kernel timing), 2 on several")
    ap.add_argument("--share", type=int, default=0,
                    help="mi355rt_context_set_share of every frame slot: each frame's persistent kernel launches 1/share of the resident grid (0 = auto: the workload's "
                         "measured choice with several GPUs, 1 on one GPU)")
    ap.add_argument("--tail-parts", type=int, default=0,
                    help="after the timed region also time 1/P-image launches (strip part p of P, every p) on this GPU: "
                         "what one of P GPUs would run; reports ideal (full/P) vs measured")
    ap.add_argument("--inflight", default="",
                    help="after the timed region also time frames in flight: a comma list of FxD (F contexts + streams, each launching 1/D of the resident "
                         "grid: mi355rt_context_set_share), at full size and -- with --tail-parts P -- at 1/P image; e.g. 1x1,2x1,2x2,3x3,4x4")
    ap.add_argument("--save-png", default="")
    ap.add_argument("--launch-timeout", type=float, default=200.0,
                    help="`--gpus N` as typed: seconds the self-started N-rank launch has to print its result line before its process group is "
                         "killed and the --single-process fallback is started (same bound); 0 = unbounded.  Well inside a 600 s driver limit: "
                         "a healthy 8-rank launch needs well under a minute once the image is paged in")
    ap.add_argument("--teardown-grace", type=float, default=20.0,
                    help="`--gpus N` as typed: seconds a launch may take to exit after its result line before it is ended (its line stands, exit code 0)")
    ap.add_argument("--rendezvous-timeout", type=float, default=120.0,
                    help="rank path: timeout of init_process_group and of every collective (torch's default for nccl is 10 minutes, longer than a driver's whole "
                         "limit; 120 s leaves room for ranks whose first `import torch` on a fresh box pages the image in at different speeds)")
    ap.add_argument("--no-one-shot", action="store_true", help="skip the one-shot mi355rt_render timing (N = 1) that the line carries as `one_shot`")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1 without a launcher: one process, one context + stream per device, gather by device-to-device copies into device 0 "
                         "(what `bench.py --gpus N` falls back to when the one-rank-per-GPU launch cannot start)")
    args = ap.parse_args(argv)

    # `python bench.py --gpus N` as typed (no launcher around it): start the ranks ourselves, in fresh child processes, BEFORE anything here
    # imports torch or touches the GPU.  (MI355RT_BENCH_FORCE_DIST=1 takes the same route with N = 1: the RCCL branch with one rank.)
    if "RANK" not in os.environ and not args.single_process and (args.gpus > 1 or os.environ.get("MI355RT_BENCH_FORCE_DIST") == "1"):
        sys.exit(self_launch(args, list(sys.argv[1:] if argv is None else argv)))
    if args.single_process:
        return main_single_process(args)

    # RCCL sets up its intra-node transport with HIP IPC handles; on this pool the host driver only supports dmabuf IPC, and the
    # legacy mode fails in `hipIpcGetMemHandle: invalid argument` as soon as there are two ranks.  The variable has to be in the
    # environment before the HIP runtime initialises, i.e. before `import torch`.  setdefault: a caller's
    # explicit choice wins.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:                                   # under a launcher the launcher's world size is the truth
        args.gpus = world

    import torch
    import torch.distributed as dist
    abi, host, device, rtdist, build = product_modules()

    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    # Rehearsal on a one-GPU box (MI355RT_BENCH_REHEARSE=1): every rank shares cuda:0 and the gather goes through
    # gloo on host copies, because RCCL refuses two ranks on one GPU.  It checks the multi-rank plumbing only;
    # its numbers mean nothing and the JSON says so.
    rehearse = os.environ.get("MI355RT_BENCH_REHEARSE") == "1" and world > 1
    # MI355RT_BENCH_FORCE_DIST=1 under `torch.distributed.run --nproc-per-node 1`: the whole multi-GPU branch -- RCCL process group,
    # barrier, all_gather_into_tensor on the launch stream, frames in flight, all_reduce of the time -- with ONE rank.  It is the
    # only way to execute the `nccl` calls on a one-GPU box (RCCL refuses two ranks on one device); the line says so.
    use_dist = world > 1 or (os.environ.get("MI355RT_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # One line per rank BEFORE the first collective, on stderr: a rendezvous that forms only in part is then visible in the caller's log
        # (which ranks came up, on which devices) instead of ending as a silent wait.
        print(f"bench.py: rank {rank}/{world} pid {os.getpid()} on device {local_rank} ({torch.cuda.get_device_name(local_rank)}) up; "
              f"joining the process group at {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')} (timeout {args.rendezvous_timeout:.0f} s)", file=sys.stderr, flush=True)
        limit = datetime.timedelta(seconds=max(1.0, args.rendezvous_timeout))   # torch's default for nccl is 10 minutes: longer than a driver's whole limit
        if rehearse:
            dist.init_process_group(backend="gloo", timeout=limit)
        else:
            dist.init_process_group(backend="nccl", device_id=dev, timeout=limit)  # "nccl" is RCCL on ROCm
        print(f"bench.py: rank {rank}/{world}: process group formed ({dist.get_backend()}, world size {dist.get_world_size()})", file=sys.stderr, flush=True)

    path, W, H, spp, depth, skip_unknown = WORKLOADS[args.workload]
    scene = host.LoadedScene(os.path.join(ROOT, path), W, H, spp, depth, skip_unknown_primitives=skip_unknown)   # product loader (C++)
    plan = rtdist.make_plan(H, W, world)
    opt = plan.options_for(abi, rank)
    n_local_rows = len(plan.rows[rank])
    depth_pipe, share = (args.pipeline, args.share if args.share > 0 else 1) if args.pipeline > 0 else (frames_in_flight(args.workload) if use_dist else (1, 1))
    if args.share > 0:
        share = args.share
    args.share_used = share
    # One frame slot = its own context (radiance workspace, work counters), output buffer and stream, so that two frames
    # never share scratch memory.  The scene is resident in HBM in every slot from here on.
    slots = []
    streams, stream_info = ([torch.cuda.current_stream()], None) if depth_pipe == 1 else concurrent_streams(torch, dev, depth_pipe)
    depth_pipe, share, stream_info = fit_frames_to_queues(depth_pipe, share, stream_info, auto=args.pipeline <= 0 and args.share <= 0)
    args.stream_info = stream_info; args.share_used = share
    for i in range(depth_pipe):
        ctx = device.Context(local_rank)
        ctx.set_share(share)
        ctx.set_scene(scene, scene.camera, scene.settings)
        slots.append({"ctx": ctx, "local": torch.zeros((plan.max_rows, W), dtype=torch.int32, device=dev), "stream": streams[i]})
    ctx0 = slots[0]["ctx"]

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    image = None
    frame = 0

    def step():
        nonlocal image, frame
        s = slots[frame % depth_pipe]
        frame += 1
        with torch.cuda.stream(s["stream"]):
            s["ctx"].render(s["local"].data_ptr(), None, opt, s["stream"].cuda_stream)     # enqueue only: no host sync inside
            image = rtdist.gather_image(s["local"].cpu() if rehearse else s["local"], plan, rank, always=use_dist)

    if depth_pipe > 1:                                       # initialisation, not a step: every frame slot renders once, so that its radiance workspace is allocated
        for s in slots:                                      # and its row table uploaded before anything is timed (with W < frames in flight the warm-up would not reach every slot)
            s["ctx"].render(s["local"].data_ptr(), None, opt, s["stream"].cuda_stream)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    sync_all()
    for s in slots:
        s["ctx"].set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    # The timed renders were enqueued asynchronously (stats == NULL); a kernel that left an image incomplete is reported here
    # (mi355rt_context_check, sticky error word) and the run ends WITHOUT a result line: a partial image is not a measurement.
    for s in slots:
        try:
            s["ctx"].check()
        except device.RenderError as e:
            sys.exit(f"bench.py: rank {rank}: a timed render did not complete -- {e}")
    k_render_ms = k_resolve_ms = 0.0
    launches = 0
    for s in slots:
        a, b, n = s["ctx"].read_timing()
        k_render_ms += a; k_resolve_ms += b; launches += n
        s["ctx"].set_timing(False)
    my_elapsed = elapsed
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    torch.cuda.synchronize()
    image_checksum = None if image is None else int(image.to(torch.int64).sum().item())    # identical for every N and pipeline depth (RNG keyed by absolute row)
    final_image = None if image is None else image.clone()                                  # the slot buffers are reused below

    # one extra (untimed) step with counters for rays/sample
    torch.cuda.synchronize()
    st = ctx0.render(slots[0]["local"].data_ptr(), None, opt, slots[0]["stream"].cuda_stream, want_stats=True)
    local_samples = n_local_rows * W * spp
    variant = ctx0.kernel_variant()
    render_ms_per_step = k_render_ms / max(args.steps, 1)    # kernel time per STEP (= per full render of this rank's rows, all workspace bands together),
    resolve_ms_per_step = k_resolve_ms / max(args.steps, 1)  # from the HIP events the library recorded around every launch of the timed region

    # What proves that N ranks on N devices took part: every rank's identity and its own kernel time, collected with the process group.
    ranks = None
    if use_dist:
        me = dict(device_identity(torch, local_rank), rank=rank, local_rank=local_rank, pid=os.getpid(), rows=n_local_rows,
                  kernel_ms_per_step=round(render_ms_per_step, 4), resolve_ms_per_step=round(resolve_ms_per_step, 4),
                  step_wall_ms=round(my_elapsed / max(args.steps, 1) * 1e3, 4), rays=int(st.rays))
        ranks = [None] * world
        dist.all_gather_object(ranks, me)

    one_shot = None
    if world == 1 and not use_dist and not args.no_one_shot:
        one_shot = measure_one_shot(abi, device, scene)
    tail = None
    if args.tail_parts > 1 and world == 1:
        tail = measure_tail(abi, rtdist, slots, H, W, args.tail_parts, max(3, args.steps // 2), render_ms_per_step, torch, device, scene, resolve_ms_per_step, args.workload)

    inflight = None
    if args.inflight and world == 1:
        inflight = measure_inflight(abi, rtdist, torch, device, scene, local_rank, H, W, args.inflight, args.tail_parts, max(4, args.steps // 2))
    result = None
    if rank == 0:
        launch = os.environ.get("MI355RT_BENCH_LAUNCH") or ("external launcher (RANK / WORLD_SIZE were in the environment)" if "RANK" in os.environ else "direct: one process, one GPU")
        dist_info = None
        if use_dist:
            dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": ranks,
                         "distinct_devices": len({(r["device_index"], r["pci_bus_id"]) for r in ranks})}
        result = make_result(args, abi, build, rtdist, scene, plan, world, elapsed, render_ms_per_step, resolve_ms_per_step,
                             launches / max(args.steps, 1), local_samples, st, variant, depth_pipe,
                             gather=(f"RCCL {rtdist.collective_name(rehearse)} of the packed rows" if use_dist else ""),
                             launch=launch, dist_info=dist_info, tail=tail, image_checksum=image_checksum,
                             extras={**({"one_shot": one_shot} if one_shot else {}), **({"inflight": inflight} if inflight else {}),
                                     **({"rehearsal": "all ranks on cuda:0 over gloo -- plumbing check only, NOT a measurement"} if rehearse else {}),
                                     **({"forced_dist": "one rank through the RCCL branch (process group, barrier, all_gather_into_tensor, all_reduce) -- a check of the calls, not a multi-GPU measurement"} if use_dist and world == 1 else {})})
        if args.save_png and final_image is not None:
            import numpy as np
            host.write_png(args.save_png, final_image.cpu().numpy().astype(np.uint32), W, H)
        print(json.dumps(result), flush=True)
    if use_dist:
        try:                                                 # the line is out: a failure in teardown must not turn a complete measurement into a failed run
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:
            print(f"bench.py: rank {rank}: teardown of the process group failed after the result line: {e}", file=sys.stderr, flush=True)
    for s in slots:
        s["ctx"].close()
    return result


# Frames in flight for N > 1 (own context, workspace, output buffer and stream each): the next frame's workgroups start as this frame's
# leave, which hides part of the launch tail of a 1/N-image frame -- where the workspace is small.  Measured at 1/8 image
# (profiles/r04/tail_eighth_image_two_streams.txt): cornell 0.856 -> 0.888, teapot 0.539 -> 0.742, semesterbild 0.559 -> 0.672 of ideal, but
# veach-mis 0.941 -> 0.795 (two 1.4 GB workspaces alternating cost more than the overlap buys).
# Round 5: every frame slot's context is told its SHARE of the device (mi355rt_context_set_share): with F frames in flight each persistent kernel launches
# 1 / share of the resident grid, so the launches are co-resident and a draining frame shares the SIMDs with frames in their steady state.  Measured at
# 1/8 image, steady-state time per frame against (full-frame kernels / 8) (profiles/r05/inflight_*.txt): cornell 0.857 (1 frame) -> 0.953 (4 frames, share 4),
# teapot 0.518 -> 0.836 (4, share 2), semesterbild 0.572 -> 0.876 (4, 4), veach-mis 0.940 -> 0.949 (4, 4).  More than 4 streams buy nothing: HIP multiplexes
# streams onto 4 hardware queues (6 x 1/6: 0.69, 8 x 1/8: 0.52).
FRAMES_IN_FLIGHT = {"teapot-800x600x256-d64": (4, 2)}     # workload -> (frames in flight, share of the device per frame); default (4, 4)


def frames_in_flight(workload):
    return FRAMES_IN_FLIGHT.get(workload, (4, 4))


_STREAM_CLASSES = {}


def concurrent_streams(torch, dev, want, pool=24):
    """`want` torch streams of device `dev` that really run CONCURRENTLY with each other, plus a record of how they were found.
    Frames in flight only overlap when their streams sit on different hardware queues, and HIP multiplexes streams onto GPU_MAX_HW_QUEUES (4)
    queues in an order the caller cannot see: of 12 streams created in a row, (0, 1, 2, 3) shared a queue between two of them while (4, 5, 6, 7) did
    not (tools/stream_queue_probe.py, profiles/r05/stream_queue_probe.txt) -- and four frames on three queues lose a third of the device.  So the
    streams are PROBED: a one-block spin kernel (torch.cuda._sleep) on two streams takes T when they overlap and 2T when they share a queue; every
    new stream is compared with one representative of each queue class found so far.  A few milliseconds, once per process and device."""
    key = (dev.index if hasattr(dev, "index") else int(dev))
    if key not in _STREAM_CLASSES:
        info = {"method": "torch.cuda._sleep pair probe", "pool": pool}
        streams = [torch.cuda.Stream(device=dev) for _ in range(pool)]
        classes = []
        try:
            def pair_ms(a, b, cycles):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                with torch.cuda.stream(a):
                    torch.cuda._sleep(cycles)
                with torch.cuda.stream(b):
                    torch.cuda._sleep(cycles)
                torch.cuda.synchronize(dev)
                return (time.perf_counter() - t0) * 1e3
            cycles = 1_000_000
            pair_ms(streams[0], streams[0], cycles)                                  # warm
            serial = min(pair_ms(streams[0], streams[0], cycles) for _ in range(3))  # two spins on ONE stream: 2T
            if serial < 0.8:                                                         # aim at T ~ 1 ms: well above launch overheads
                cycles = int(cycles * 2.0 / max(serial, 1e-3)); serial = min(pair_ms(streams[0], streams[0], cycles) for _ in range(3))
            info.update(spin_pair_serial_ms=round(serial, 3))
            for s in streams:
                shared = False
                for rep in classes:
                    if min(pair_ms(rep, s, cycles), pair_ms(rep, s, cycles)) > 0.75 * serial:   # (overlap: ~0.5 x serial)
                        shared = True
                        break
                if not shared:
                    classes.append(s)
                if len(classes) >= 8:
                    break
            info.update(queue_classes_found=len(classes))
        except Exception as e:                                                        # no _sleep on this build, ...: the first streams, unprobed -- and the line says so
            info.update(method=f"unprobed ({e})", queue_classes_found=None)
            classes = []
        rest = [s for s in streams if all(s is not c for c in classes)]
        _STREAM_CLASSES[key] = (classes, rest, info)
    classes, rest, info = _STREAM_CLASSES[key]
    picked = (classes + rest)[:want]
    return picked, dict(info, wanted=want, distinct=min(want, len(classes)))


def fit_frames_to_queues(frames, share, info, auto=True):
    """Frames in flight that share a hardware queue serialise -- four quarter-grid kernels on three queues leave a third of the device idle (0.68 of
    ideal instead of 0.95, profiles/r05/stream_queue_probe.txt).  When the probe found fewer queue classes than frames were wanted (another runtime
    configuration, other tenants of the queues), the automatic choice falls back to as many frames as there are classes, each on that share of the
    device; an explicit --pipeline / --share is left as typed."""
    if not info or not auto:
        return frames, share, info
    found = info.get("queue_classes_found")
    if found is not None and 1 <= found < frames:
        info = dict(info, reduced_from=[frames, share], why=f"only {found} hardware-queue class(es) found")
        frames = found
        share = min(share, frames)
    return frames, share, info


def product_modules():
    return tuple(importlib.import_module("raytracer-rust_amd." + m) for m in ("abi", "host", "device", "distributed", "build"))


def device_identity(torch, index):
    """Which physical device a rank / a part ran on: index, name, PCI address (domain:bus:device), uuid where torch has them."""
    p = torch.cuda.get_device_properties(index)
    dom, bus, devid = getattr(p, "pci_domain_id", None), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None)
    pci = None if bus is None else f"{dom if dom is not None else 0:04x}:{bus:02x}:{devid if devid is not None else 0:02x}"
    return {"device_index": index, "device_name": p.name, "pci_bus_id": pci, "uuid": str(getattr(p, "uuid", "")) or None,
            "visible_devices": torch.cuda.device_count()}


def make_result(args, abi, build, rtdist, scene, plan, world, elapsed, render_ms_per_step, resolve_ms_per_step, launches_per_step,
                local_samples, st, variant, depth_pipe, gather, launch, dist_info, tail, image_checksum, extras):
    """The one JSON line (rank 0 / the single process).  `st` = stats of one untimed render of the caller's own rows (rays, grid);
    render_ms_per_step / local_samples describe the same rows."""
    path, W, H, spp, depth, _ = WORKLOADS[args.workload]
    total_samples = W * H * spp
    value = total_samples * args.steps / elapsed / 1e6
    rays_per_sample = st.rays / max(st.samples, 1)
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
    share = local_samples / total_samples            # a rank renders its strips only; counters were taken on the whole image
    khash = build.kernel_hash()
    pmc, pmc_state = load_pmc(args.workload, khash, KERNEL_NAMES.get(variant))
    peak_inst = VALU_SIMDS * VALU_CLOCK_HZ / VALU_CYCLES_PER_WAVE64_INST
    achieved = frac = lane_util = traffic = insts = None
    hbm = None
    if pmc and render_ms_per_step > 0:
        insts = pmc["valu_wave_insts_per_step"] * share
        achieved = insts / (render_ms_per_step * 1e-3)
        frac = achieved / peak_inst
        lane_util = pmc.get("valu_lane_utilisation")
        traffic = int(pmc["hbm_bytes_per_step"] * share)
        gbs = traffic / (render_ms_per_step * 1e-3) / 1e9
        hbm = {"traffic_bytes": traffic, "achieved_GBs": round(gbs, 1), "peak_GBs": HBM_PEAK_GBS, "frac": round(gbs / HBM_PEAK_GBS, 4),
               "formula": "(2*FETCH_SIZE + WRITE_SIZE) KiB x 1024 per step; x2 = gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md, HBM)"}
    bytes_per_sample = rays_per_sample * (rec + 48 + nodes_per_ray * 32 + tris_per_ray * 48) + 16.0 / spp
    result = {
        "metric": f"Msamples/s (pixels x spp / s) at {W}x{H}x{spp}spp; 1/2/4/8-GPU scaling",
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": f"synthetic (the reference's own scene file {path}; no external data)",
        "config": {"workload": args.workload, "scene": path, "width": W, "height": H, "spp": spp, "max_bounces": depth,
                   "rng": "ctr (pcg4d counter hash: per-path base = pcg4d(x, sample, row key); block j after ray r = pcg4d(base + (0, 0, r, j)))", "parallelism": f"row strips of {plan.strip_rows} dealt round-robin over {world} GPU(s)"
                   + (f", {gather}" if gather else ""),
                   "frames_in_flight": depth_pipe, "share_of_device_per_frame": getattr(args, "share_used", 1),
                   **({"streams": args.stream_info} if getattr(args, "stream_info", None) else {})},
        "launch": launch,
        **({"distributed": dist_info} if dist_info else {}),
        "roofline": {"bound": "valu", "kernel": KERNEL_NAMES.get(variant, "k_render_ctr"),
                     "achieved": None if achieved is None else round(achieved / 1e9, 1), "peak": round(peak_inst / 1e9, 1),
                     "unit": "G wave-instructions/s", "frac": None if frac is None else round(frac, 4), "traffic": traffic,
                     "lane_utilisation": lane_util,
                     "wave_insts_per_step": None if achieved is None else int(insts),
                     "wave_insts_per_ray": None if achieved is None or st.rays == 0 else round(insts * 64.0 / st.rays, 1),   # VALU instructions a wave issues per 64 rays
                     "useful_frac": None if frac is None or lane_util is None else round(frac * lane_util, 4),
                     "hbm": hbm,
                     "kernel_ms_per_step": round(render_ms_per_step, 4), "resolve_ms_per_step": round(resolve_ms_per_step, 4),
                     "kernel_launches_per_step": launches_per_step, "samples_per_step": local_samples,
                     "pmc": pmc_state, "kernel_hash": khash,
                     "logical_bytes_model": {"bytes_per_sample": round(bytes_per_sample, 1), "rays_per_sample": round(rays_per_sample, 4),
                                             "bytes_per_step": int(bytes_per_sample * local_samples),
                                             "note": "SURVEY.md 8d algorithmic bytes; SGPR/L2 resident, never reach HBM -- a model, not an HBM fraction"},
                     "note": "achieved = SQ_INSTS_VALU per step (PMC, profiles/pmc_counters.json, same kernel hash) / live HIP-event kernel time per step; "
                             "peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (nominal: a loop of nothing but independent "
                             "v_mul_f32 / v_add_f32 reaches 0.82-0.94 of it at a sustained 2.32-2.39 GHz -- tools/microbench/clock.hip, "
                             "profiles/r03_microbench_clock_and_issue.txt -- and v_fma / v_rcp / v_sqrt / 64-bit multiplies take more than one slot)"
                             + (f"; {depth_pipe} frames are in flight on their own streams, each on 1/{getattr(args, 'share_used', 1)} of the resident grid: a kernel's event time includes the time it shared the chip -- "
                                "it is NOT the time of one frame alone, and `frac` is that of rank 0's kernel while the others ran beside it" if depth_pipe > 1 else "")
                             + ("; kernel figures are those of rank / device 0's rows" if world > 1 else "")},
        "cpu_baseline": cpu_baseline,
        **({"tail": tail} if tail else {}),
        **extras,
        "kernel": {"vgprs": st.kernel_vgprs, "grid_blocks": st.grid_blocks, "block_threads": st.block_threads, "bands": st.bands},
    }
    if image_checksum is not None:
        result["image_checksum"] = image_checksum
    return result


# ---------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` as typed: the parent process.  It imports neither torch nor the product and never touches a GPU.
# ---------------------------------------------------------------------------------------------------
def _kill_group(proc, why):
    """Ends the child AND everything it started (it was made the leader of its own session / process group): SIGTERM, five seconds, SIGKILL."""
    import signal
    print(f"bench.py: {why}; ending the child's process group {proc.pid}", file=sys.stderr, flush=True)
    for sig, wait_s in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass
        try:
            proc.wait(timeout=wait_s)
            break
        except Exception:
            continue


def run_relay(cmd, env, deadline_s=None, grace_s=20.0):
    """Runs `cmd` as a child process IN ITS OWN PROCESS GROUP, passes its stdout through line by line (stderr is inherited) and
    returns {"rc", "saw", "timed_out", "killed_in_teardown", "seconds"}.  `saw`: a result line -- a JSON object with "metric" -- went by.
    Two bounds, so that this parent can never sit on a hung child until ITS caller's limit kills both without a line:
      * deadline_s (None = unbounded): the first result line must arrive within it, else the whole group is killed (`timed_out`);
      * grace_s: once a result line has gone by, the child has this long to finish (destroy_process_group, the launcher's own teardown);
        a child that hangs there is killed too, but the measurement it printed is complete: rc 0, `killed_in_teardown`.
    The parent never touched the GPU, so killing the group is the "fresh child" case, not an exec."""
    import queue
    import signal
    import subprocess
    import threading
    t0 = time.monotonic()
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1, start_new_session=True)
    lines = queue.Queue()

    def reader():
        try:
            for line in proc.stdout:
                lines.put(line)
        finally:
            lines.put(None)
    threading.Thread(target=reader, daemon=True).start()

    def on_signal(signum, _frame):                               # the parent itself is being ended (a driver's limit): do not orphan the ranks
        _kill_group(proc, f"bench.py received signal {signum}")
        sys.exit(128 + signum)
    old = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    out = {"rc": None, "saw": False, "timed_out": False, "killed_in_teardown": False}
    t_line = None
    try:
        while True:
            limit = (t_line + grace_s) if out["saw"] else (None if not deadline_s else t0 + deadline_s)
            try:
                line = lines.get(timeout=None if limit is None else max(0.0, limit - time.monotonic()))
            except queue.Empty:
                if out["saw"]:
                    _kill_group(proc, f"the result line is out but the child has not finished {grace_s:.0f} s later (teardown hang)")
                    out["killed_in_teardown"] = True; out["rc"] = 0
                else:
                    _kill_group(proc, f"no result line within {deadline_s:.0f} s (--launch-timeout)")
                    out["timed_out"] = True; out["rc"] = 124
                break
            if line is None:                                     # end of the child's stdout
                break
            if line.lstrip().startswith("{") and '"metric"' in line:
                out["saw"] = True; t_line = time.monotonic()
            sys.stdout.write(line); sys.stdout.flush()
        if out["rc"] is None:
            try:
                out["rc"] = proc.wait(timeout=grace_s if (out["saw"] or deadline_s) else None)
            except subprocess.TimeoutExpired:
                _kill_group(proc, f"stdout closed but the child did not exit within {grace_s:.0f} s")
                out["killed_in_teardown"] = out["saw"]; out["timed_out"] = not out["saw"]; out["rc"] = 0 if out["saw"] else 124
    except BaseException:
        _kill_group(proc, "interrupted")
        raise
    finally:
        for sig, h in old.items():
            signal.signal(sig, h)
    out["seconds"] = round(time.monotonic() - t0, 1)
    return out


def launcher_command(n, argv):
    """The command line of the one-rank-per-GPU launch: torch.distributed.run on 127.0.0.1 with a port that is free right now.
    MI355RT_BENCH_LAUNCHER (a command prefix) replaces the launcher -- tests put a stub there and read back what it was given."""
    import shlex
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    launcher = shlex.split(os.environ.get("MI355RT_BENCH_LAUNCHER", "")) or [sys.executable, "-m", "torch.distributed.run"]
    return launcher + ["--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv


def self_launch(args, argv):
    """`python bench.py --gpus N` as typed.  The one-rank-per-GPU launch gets --launch-timeout seconds to print its result line; when it
    fails, prints none or runs out of time, ONE fresh --single-process child drives all N devices instead (same bound), and its line says so
    in machine-readable form ("fallback": true, "rank_launch_rc", "rank_launch_timed_out").  A launch whose line went by is never repeated."""
    n = max(1, args.gpus)
    limit = args.launch_timeout if args.launch_timeout > 0 else None
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # before any child initialises HIP (see main)
    env["MI355RT_BENCH_LAUNCH"] = f"self: bench.py --gpus {n} started torch.distributed.run ({n} fresh child process(es), 127.0.0.1)"
    try:
        r = run_relay(launcher_command(n, argv), env, deadline_s=limit, grace_s=args.teardown_grace)
    except OSError as e:                                     # the launcher itself could not be started (not found, no more processes, ...)
        print(f"bench.py: could not start the launcher: {e}", file=sys.stderr, flush=True)
        r = {"rc": 127, "saw": False, "timed_out": False, "killed_in_teardown": False, "seconds": 0.0}
    if r["saw"]:
        # The measurement is complete (the line is printed behind the timed region and the per-context checks).  A child that then hung in
        # teardown was killed and counts as done; one that exited by itself hands its own code on (0 normally) -- and no second line is made.
        if r["killed_in_teardown"]:
            print("bench.py: result line relayed; the launch was ended in its teardown -- exit code 0", file=sys.stderr, flush=True)
        elif r["rc"] != 0:
            print(f"bench.py: result line relayed, but the launch then ended with code {r['rc']} (teardown); no fallback, the code is passed on", file=sys.stderr, flush=True)
        return r["rc"]
    why = (f"timed out: no result line within {limit:.0f} s" if r["timed_out"] else f"ended with code {r['rc']} and printed no result line")
    if n > 1 and os.environ.get("MI355RT_BENCH_NO_FALLBACK") != "1":
        print(f"bench.py: the {n}-rank launch {why}; starting ONE fresh process that drives all {n} devices (--single-process)", file=sys.stderr, flush=True)
        env["MI355RT_BENCH_LAUNCH"] = f"fallback: one process drives {n} devices (--single-process) because the {n}-rank launch {why}"
        env["MI355RT_BENCH_FALLBACK_INFO"] = json.dumps({"fallback": True, "rank_launch_rc": r["rc"], "rank_launch_timed_out": r["timed_out"],
                                                         "rank_launch_seconds": r["seconds"]})
        try:
            import shlex
            fallback = shlex.split(os.environ.get("MI355RT_BENCH_FALLBACK_CMD", "")) or [sys.executable, os.path.abspath(__file__)]   # (tests put a stub there)
            r2 = run_relay([*fallback, *argv, "--single-process"], env, deadline_s=limit, grace_s=args.teardown_grace)
        except OSError as e:
            print(f"bench.py: could not start the fallback child: {e}", file=sys.stderr, flush=True)
            return 127
        if r2["saw"]:
            return r2["rc"]
        print(f"bench.py: the fallback child {'timed out' if r2['timed_out'] else 'ended with code ' + str(r2['rc'])} without a result line", file=sys.stderr, flush=True)
        return r2["rc"] if r2["rc"] != 0 else 1
    print(f"bench.py: the {n}-rank launch {why}", file=sys.stderr, flush=True)
    return r["rc"] if r["rc"] != 0 else 1


# ---------------------------------------------------------------------------------------------------
# --single-process: ONE process, N devices.  The strip plan, the contexts and the kernels are those of the rank path; the exchange step is
# N device-to-device copies into device 0 (hipMemcpyPeerAsync under torch's copy_) instead of the RCCL all-gather, then the same index_select.
# ---------------------------------------------------------------------------------------------------
def main_single_process(args):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    abi, host, device, rtdist, build = product_modules()
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    n = max(1, args.gpus)
    # MI355RT_BENCH_REHEARSE=1 on a one-GPU box: every part runs on cuda:0 (own context, buffers and stream each) -- a check of the plumbing.
    rehearse = os.environ.get("MI355RT_BENCH_REHEARSE") == "1"
    if not rehearse and torch.cuda.device_count() < n:
        sys.exit(f"bench.py: --single-process --gpus {n} but only {torch.cuda.device_count()} device(s) are visible")
    devs = [0 if rehearse else d for d in range(n)]
    path, W, H, spp, depth, skip_unknown = WORKLOADS[args.workload]
    scene = host.LoadedScene(os.path.join(ROOT, path), W, H, spp, depth, skip_unknown_primitives=skip_unknown)
    plan = rtdist.make_plan(H, W, n)
    depth_pipe, share = (args.pipeline, max(args.share, 1)) if args.pipeline > 0 else frames_in_flight(args.workload)
    if args.share > 0:
        share = args.share
    args.share_used = share
    parts = []                                               # parts[d] = that device's frame slots
    for d in range(n):
        with torch.cuda.device(devs[d]):
            slots = []
            streams, info = concurrent_streams(torch, torch.device("cuda", devs[d]), depth_pipe * (n if rehearse else 1))
            if d == 0:
                if not rehearse:
                    depth_pipe, share, info = fit_frames_to_queues(depth_pipe, share, info, auto=args.pipeline <= 0 and args.share <= 0)
                args.stream_info = info; args.share_used = share
            for k in range(depth_pipe):
                ctx = device.Context(devs[d])
                ctx.set_share(share)
                ctx.set_scene(scene, scene.camera, scene.settings)
                slots.append({"ctx": ctx, "local": torch.zeros((plan.max_rows, W), dtype=torch.int32, device=f"cuda:{devs[d]}"),
                              "stream": streams[(d * depth_pipe + k) % len(streams)] if rehearse else streams[k], "ready": torch.cuda.Event()})
            parts.append({"slots": slots, "opt": plan.options_for(abi, d)})
    dev0 = torch.device("cuda", devs[0])
    stacked = [torch.empty((n * plan.max_rows, W), dtype=torch.int32, device=dev0) for _ in range(depth_pipe)]
    gather_streams = [torch.cuda.Stream(device=dev0) for _ in range(depth_pipe)]
    perm = plan.perm_on(dev0)
    image = None
    frame = 0

    def step():
        nonlocal image, frame
        k = frame % depth_pipe
        frame += 1
        for d in range(n):                                    # every device's strips, enqueued without a host sync
            s = parts[d]["slots"][k]
            with torch.cuda.device(devs[d]), torch.cuda.stream(s["stream"]):
                s["ctx"].render(s["local"].data_ptr(), None, parts[d]["opt"], s["stream"].cuda_stream)
                s["ready"].record(s["stream"])
        with torch.cuda.device(devs[0]), torch.cuda.stream(gather_streams[k]):
            for d in range(n):                                # the exchange step: device d's packed rows -> its block of device 0's buffer
                gather_streams[k].wait_event(parts[d]["slots"][k]["ready"])
                stacked[k][d * plan.max_rows:(d + 1) * plan.max_rows].copy_(parts[d]["slots"][k]["local"], non_blocking=True)
            image = stacked[k].index_select(0, perm)
            done = torch.cuda.Event(); done.record(gather_streams[k])
        for d in range(n):                                    # frame k + depth_pipe may not overwrite `local` before this gather has read it
            parts[d]["slots"][k]["stream"].wait_event(done)

    def sync_all():
        for d in sorted(set(devs)):
            torch.cuda.synchronize(d)

    for d in range(n):                                       # initialisation, not a step: every frame slot of every device renders once (workspace, row table)
        for s in parts[d]["slots"]:
            with torch.cuda.device(devs[d]):
                s["ctx"].render(s["local"].data_ptr(), None, parts[d]["opt"], s["stream"].cuda_stream)
    sync_all()
    for _ in range(args.warmup):
        step()
    sync_all()
    for p in parts:
        for s in p["slots"]:
            s["ctx"].set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    per_part = []
    for d, p in enumerate(parts):
        a = b = 0.0; ln = 0
        for s in p["slots"]:
            try:
                s["ctx"].check()
            except device.RenderError as e:
                sys.exit(f"bench.py: device {devs[d]}: a timed render did not complete -- {e}")
            x, y, m = s["ctx"].read_timing()
            a += x; b += y; ln += m
            s["ctx"].set_timing(False)
        per_part.append((a / max(args.steps, 1), b / max(args.steps, 1), ln / max(args.steps, 1)))
    image_checksum = int(image.to(torch.int64).sum().item())
    final_image = image.clone()
    sync_all()
    stats = []
    for d, p in enumerate(parts):
        s = p["slots"][0]
        with torch.cuda.device(devs[d]):
            stats.append(s["ctx"].render(s["local"].data_ptr(), None, p["opt"], s["stream"].cuda_stream, want_stats=True))
    variant = parts[0]["slots"][0]["ctx"].kernel_variant()
    ranks = [dict(device_identity(torch, devs[d]), part=d, pid=os.getpid(), rows=len(plan.rows[d]), kernel_ms_per_step=round(per_part[d][0], 4),
                  resolve_ms_per_step=round(per_part[d][1], 4), rays=int(stats[d].rays)) for d in range(n)]
    result = make_result(args, abi, build, rtdist, scene, plan, n, elapsed, per_part[0][0], per_part[0][1], per_part[0][2],
                         len(plan.rows[0]) * W * spp, stats[0], variant, depth_pipe,
                         gather=f"{n} device-to-device copies of the packed rows into device 0" if n > 1 else "",
                         launch=os.environ.get("MI355RT_BENCH_LAUNCH") or "direct: --single-process",
                         dist_info={"backend": "none (one process, hipMemcpyPeer)", "world_size": n, "ranks": ranks,
                                    "distinct_devices": len({(r["device_index"], r["pci_bus_id"]) for r in ranks})},
                         tail=None, image_checksum=image_checksum,
                         extras={**json.loads(os.environ.get("MI355RT_BENCH_FALLBACK_INFO", "{}")),
                                 **({"rehearsal": "every part on cuda:0 -- plumbing check only, NOT a measurement"} if rehearse else {})})
    if args.save_png:
        import numpy as np
        host.write_png(args.save_png, final_image.cpu().numpy().astype(np.uint32), W, H)
    print(json.dumps(result), flush=True)
    for p in parts:
        for s in p["slots"]:
            s["ctx"].close()
    return result


def measure_one_shot(abi, device, scene):
    """What `render_scene(&scene, &camera, &settings)` at src/main.rs:57 would pay through the one-shot entry point, mi355rt_render with HOST
    buffers: context, workspace, scene upload, both kernels, the copy of the packed image back -- wall time of the call, outside the timed
    region.  `first_call_ms`: the first such call of this process (the HIP runtime and the code object are already up: bench.py has rendered);
    `second_call_ms`: the call again.  Never part of `value`, which is the resident-scene rate."""
    opt = abi.Options.make(rng_mode=abi.RNG_CTR)
    out = {}
    try:
        for key in ("first_call_ms", "second_call_ms"):
            t0 = time.perf_counter()
            _, _, st = device.render(scene, scene.camera, scene.settings, opt, want_linear=False)
            out[key] = round((time.perf_counter() - t0) * 1e3, 3)
            out["kernels_ms"] = round(st.total_ms, 3)
        out["what"] = ("wall ms of mi355rt_render (host buffers in and out: context + workspace + scene upload + path tracing + resolve + D2H of the "
                       "packed image), what src/main.rs:57 would call; HIP runtime already initialised by this process")
    except Exception as e:                                   # never costs the GPU measurement its line
        out = {"error": str(e)}
    return out


def measure_inflight(abi, rtdist, torch, device, scene, dev_index, H, W, spec, parts, steps):
    """Frames in flight, measured: for every FxD of `spec`, F contexts (own workspace, output buffer, stream) whose persistent kernels each launch
    1/D of the grid that fills the device; frames are enqueued round-robin, wall time per frame in steady state.  Full-size frames, and the frames
    one of `parts` GPUs renders (strip part 0 of `parts`) when parts > 1."""
    out = []
    dev = torch.device("cuda", dev_index)
    plans = [("full", abi.Options.make(), H)]
    if parts > 1:
        plan = rtdist.make_plan(H, W, parts)
        plans.append((f"1/{parts}", plan.options_for(abi, 0), plan.max_rows))
    for item in spec.split(","):
        F, D = (int(t) for t in item.lower().split("x"))
        ctxs = []
        streams, info = concurrent_streams(torch, dev, F)
        for k in range(F):
            c = device.Context(dev_index)
            c.set_share(D)
            c.set_scene(scene, scene.camera, scene.settings)
            ctxs.append({"ctx": c, "stream": streams[k], "local": torch.zeros((H, W), dtype=torch.int32, device=dev)})
        rec = {"frames_in_flight": F, "grid_div": D, "distinct_queues": info["distinct"]}
        for tag, o, _rows in plans:
            for s in ctxs:                                           # warm: row tables, workspace
                s["ctx"].render(s["local"].data_ptr(), None, o, s["stream"].cuda_stream)
            torch.cuda.synchronize()
            n = F * max(steps, 4)
            t0 = time.perf_counter()
            for i in range(n):
                s = ctxs[i % F]
                s["ctx"].render(s["local"].data_ptr(), None, o, s["stream"].cuda_stream)
            torch.cuda.synchronize()
            rec[f"ms_per_frame_{tag}"] = round((time.perf_counter() - t0) / n * 1e3, 4)
            rec[f"checksum_{tag}"] = int(ctxs[-1]["local"].to(torch.int64).sum().item())
        for s in ctxs:
            s["ctx"].check(); s["ctx"].close()
        out.append(rec)
    return out


def measure_tail(abi, rtdist, slots, H, W, parts, steps, full_render_ms, torch, device, scene, full_resolve_ms=0.0, workload=""):
    """What ONE of `parts` GPUs would run: strip part p of `parts` of the same image, timed on this GPU for every p.
    ideal = full-image kernel time / parts; the difference is the launch tail (waves draining their last paths).
    Then the same 1/P-image frames back to back on TWO streams (a context each), as bench.py runs them for N > 1: the next frame's
    waves fill the CUs the draining frame leaves idle, so the steady-state time per frame is the honest per-GPU bound of the N-GPU line."""
    plan = rtdist.make_plan(H, W, parts)
    slot = slots[0]
    ctx = slot["ctx"]
    stream = slot["stream"].cuda_stream
    per_part = []
    for p in range(parts):
        o = plan.options_for(abi, p)
        ctx.render(slot["local"].data_ptr(), None, o, stream)                  # warm: uploads this part's row table
        torch.cuda.synchronize()
        ctx.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.render(slot["local"].data_ptr(), None, o, stream)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        a, b, _ = ctx.read_timing()
        ctx.set_timing(False)
        per_part.append((a / steps, b / steps, wall))
    worst_p = max(range(parts), key=lambda p: per_part[p][2])
    worst = per_part[worst_p]
    ideal = full_render_ms / parts
    # frames in flight, as bench.py runs them for N > 1: F contexts + streams, each on 1 / share of the resident grid (mi355rt_context_set_share)
    o = plan.options_for(abi, worst_p)
    dev = slot["local"].device

    def steady_state(F, share):
        ctxs = []
        streams, _info = concurrent_streams(torch, dev, F)
        for k in range(F):
            c = device.Context(dev.index)
            c.set_share(share)
            c.set_scene(scene, scene.camera, scene.settings)
            ctxs.append({"ctx": c, "local": torch.zeros_like(slot["local"]), "stream": streams[k]})
        for s in ctxs:
            s["ctx"].render(s["local"].data_ptr(), None, o, s["stream"].cuda_stream)
        torch.cuda.synchronize()
        n = F * max(steps, 4)
        t0 = time.perf_counter()
        for i in range(n):
            s = ctxs[i % F]
            s["ctx"].render(s["local"].data_ptr(), None, o, s["stream"].cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        for s in ctxs:
            s["ctx"].check(); s["ctx"].close()
        return ms
    two = steady_state(2, 1)                                   # round 4's form, kept for comparison: two full-size grids alternating
    F, share = frames_in_flight(workload)
    chosen = steady_state(F, share)
    ideal_step = (full_render_ms + full_resolve_ms) / parts
    return {"parts": parts, "strip_rows": plan.strip_rows, "ideal_render_ms": round(ideal, 4),
            "render_ms_max": round(max(x[0] for x in per_part), 4), "render_ms_mean": round(sum(x[0] for x in per_part) / parts, 4),
            "resolve_ms_max": round(max(x[1] for x in per_part), 4), "step_wall_ms_max": round(worst[2], 4),
            "tail_efficiency": round(ideal / max(x[0] for x in per_part), 4),
            "two_streams": {"part": worst_p, "ms_per_frame": round(two, 4), "ideal_ms_per_frame": round(ideal_step, 4),
                            "efficiency": round(ideal_step / two, 4),
                            "note": "1/P-image frames back to back on two streams (own context each, full-size grids), wall time per frame incl. resolve; ideal = (full-image render + resolve kernel ms) / P"},
            "frames_in_flight": {"part": worst_p, "frames": F, "share_of_device_per_frame": share, "ms_per_frame": round(chosen, 4), "ideal_ms_per_frame": round(ideal_step, 4),
                                 "efficiency": round(ideal_step / chosen, 4),
                                 "note": "what bench.py runs per GPU for N > 1: F frames in flight, each context told its share (mi355rt_context_set_share); steady-state wall time per 1/P-image frame"},
            "note": "one GPU renders strip part p of P of the image, every p in turn (no gather): an upper bound for P-GPU strong scaling"}


def run_cpu_baseline(abi, scene, W, H, spp, target_seconds):
    """Times the CPU oracle (reference RNG stream, tail-first folding) with one thread per host core this job may use on
    rows spread evenly over the image; rows are the reference's unit of parallelism (rayon par_chunks_mut, renderer.rs:87)
    and are handed to the threads from a shared queue.  At least 4 rows per thread; when whole rows at full spp would
    overshoot the budget (1920 x 4096 spp: 26 core-seconds per row) the sample keeps the rows and lowers spp -- the cost of a
    sample does not depend on how many of them a pixel gets."""
    import copy
    import oracle
    oracle.build()
    nproc = os.cpu_count() or 1
    cores, why = usable_cores()
    threads = cores                                          # every core this job may use

    def every(n_rows):
        n_rows = max(1, min(H, n_rows))
        return abi.Options.make(rng_mode=abi.RNG_REF, strip_rows=1, n_parts=max(1, H // n_rows), part=0)
    # pilot: one row per thread at a reduced spp
    st = copy.copy(scene.settings)
    pilot = abi.Settings(W, H, max(1, min(spp, 8)), st.max_depth)
    _, _, c0 = oracle.render(scene, scene.camera, pilot, every(threads), threads=threads, want_linear=False)
    rate = c0.samples / max(c0.seconds, 1e-9)                # samples/s with all threads busy
    budget = target_seconds * rate                            # samples the budget buys
    min_rows = 4 * threads
    s_spp = spp
    if budget < min_rows * W * spp:
        s_spp = max(1, int(budget / (min_rows * W)))
    n_rows = max(min_rows, int(budget / (W * s_spp)))
    opt = every(n_rows)
    sample_settings = abi.Settings(W, H, s_spp, st.max_depth)
    _, _, c = oracle.render(scene, scene.camera, sample_settings, opt, threads=threads, want_linear=False)
    rows = len(abi.rows_selected(H, opt))
    base = {"value": round(c.samples / c.seconds / 1e6, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "nproc": nproc, "cpu_model": cpu_model(), "threads": threads, "threads_rule": why,
            "sample": f"every {opt.n_parts}th row ({rows} of {H} rows = {rows / threads:.1f} per thread) at {s_spp} of {spp} spp, {c.samples} samples, {c.seconds:.1f} s; "
                      "same scene/resolution/depth; C++ restatement of the reference's rayon path with its ChaCha12 row streams",
            "rays_per_sample": round(c.rays / max(c.samples, 1), 4)}
    return base, c.bvh_nodes / max(c.rays, 1), c.tri_tests / max(c.rays, 1)


if __name__ == "__main__":
    main()
