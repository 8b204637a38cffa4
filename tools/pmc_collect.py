#!/usr/bin/env python3
"""Collect the PMC counters bench.py's `roofline` object is computed from -- run ON THE GPU BOX (via gpurun).

  python3 tools/pmc_collect.py --out gpurun_out/pmc [--workloads W1 W2 ...]

For every workload and every counter group one `rocprofv3 --pmc <group> -- python3 bench.py --steps 1 --warmup 1
--cpu-seconds 0 --no-one-shot --workload W` pass (counters only: never together with --stats / trace domains; FETCH_SIZE and
WRITE_SIZE in passes of their own, MI355X_MICROARCH.md "rocprofv3 PMC slots").  One bench run = 3 full renders
(warm-up, timed step, counter step), each `bands` dispatches of the path-tracing kernel; the summary divides the
dispatch totals by 3, so every figure is PER STEP (one full render, all bands).

Writes <out>/pmc_counters.json -- copy it to profiles/pmc_counters.json.  It records build.kernel_hash() of the
sources the counters were taken on; bench.py prints "pmc": "stale" instead of numbers when the hash differs.
"""
import argparse
import csv
import glob
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GROUPS = [
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES"],
    ["SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "GRBM_GUI_ACTIVE"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"],
]
DEFAULT_WORKLOADS = ["cornell-box-800x600x256-d30", "teapot-800x600x256-d64", "veach-mis-1280x720x1024-d16",
                     "semesterbild-800x600x256-d30", "semesterbild-1920x1080x4096-d30"]
RENDERS_PER_RUN = 3      # --warmup 1 + --steps 1 + the counter step


def summarize(root):
    """Per kernel family: counter -> sum over all dispatches of the run / RENDERS_PER_RUN."""
    acc, names, disp = {}, {}, {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            fam = "k_render_ctr" if "k_render_ctr" in k else "k_resolve" if "k_resolve" in k else None
            if fam is None:
                continue
            names[fam] = k.split("(")[0].split("::")[-1]
            c = row["Counter_Name"]
            acc.setdefault(fam, {}).setdefault(c, 0.0)
            acc[fam][c] += float(row["Counter_Value"])
            disp.setdefault(fam, {}).setdefault(c, 0)
            disp[fam][c] += 1
    out = {}
    for fam, d in acc.items():
        out[fam] = {c: v / RENDERS_PER_RUN for c, v in d.items()}
        out[fam]["_kernel"] = names[fam]
        out[fam]["_dispatches_per_step"] = max(disp[fam].values()) / RENDERS_PER_RUN
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/pmc")
    ap.add_argument("--workloads", nargs="*", default=DEFAULT_WORKLOADS)
    ap.add_argument("--pass-timeout", type=int, default=300)
    ap.add_argument("--groups", default="", help="JSON list of counter lists: an exploratory collection instead of the standard five passes "
                                                 "(writes pmc_extra.json with the raw per-step sums only; never profiles/pmc_counters.json)")
    args = ap.parse_args()
    extra = bool(args.groups)
    if extra:
        GROUPS[:] = json.loads(args.groups)
    build = importlib.import_module("raytracer-rust_amd.build")
    os.makedirs(args.out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    doc = {"kernel_hash": build.kernel_hash(), "collected_unix": int(time.time()),
           "command": "rocprofv3 --pmc <group> --output-format csv -- python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --no-one-shot --workload <W>",
           "groups": GROUPS, "per": "step (one full render of the workload = all workspace bands)", "workloads": {}}
    failed = []
    for wl in args.workloads:
        wdir = os.path.join(args.out, wl)
        ok = True
        for i, grp in enumerate(GROUPS):
            pdir = os.path.join(wdir, f"pass{i}")
            cmd = ["timeout", "-k", "10", str(args.pass_timeout), "rocprofv3", "--pmc", *grp, "--output-format", "csv", "-d", pdir, "--",
                   "python3", os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--cpu-seconds", "0", "--no-one-shot", "--workload", wl]
            t0 = time.time()
            with open(os.path.join(args.out, f"{wl}.pass{i}.log"), "w") as log:
                rc = subprocess.call(cmd, stdout=log, stderr=subprocess.STDOUT, env=env, cwd=ROOT)
            print(f"{wl} pass {i} {grp} exit {rc} in {time.time() - t0:.0f} s", flush=True)
            if rc == 124 or rc == 137:
                print("a pass hit its time limit: stopping (no further GPU work after a kill)", flush=True)
                json.dump(doc, open(os.path.join(args.out, "pmc_counters.partial.json"), "w"), indent=1)
                sys.exit(1)
            if rc != 0:                       # an unsupported counter, a failing bench run, ...: this workload gets NO record
                print(f"{wl}: pass {i} failed (exit {rc}, see {wl}.pass{i}.log): the workload is left out of pmc_counters.json", flush=True)
                ok = False
                break
        if not ok:
            failed.append(wl)
            continue
        s = summarize(wdir)
        r = s.get("k_render_ctr", {})
        rec = {"raw": s}
        if extra:
            doc["workloads"][wl] = rec
            json.dump(doc, open(os.path.join(args.out, "pmc_extra.json"), "w"), indent=1, sort_keys=True)
            print(wl, json.dumps({k: (round(v / 1e9, 4) if isinstance(v, float) else v) for k, v in sorted(r.items())}), "(x1e9 per step)", flush=True)
            continue
        if not all(c in r for grp in GROUPS for c in grp):      # every pass ran, but a counter is missing from its CSV
            print(f"{wl}: counters missing from the CSVs {[c for grp in GROUPS for c in grp if c not in r]}: left out", flush=True)
            failed.append(wl)
            continue
        if "SQ_INSTS_VALU" in r:
            rec["kernel"] = r["_kernel"]
            rec["valu_wave_insts_per_step"] = r["SQ_INSTS_VALU"]
            if "SQ_THREAD_CYCLES_VALU" in r:
                rec["valu_lane_utilisation"] = round(r["SQ_THREAD_CYCLES_VALU"] / (64.0 * r["SQ_INSTS_VALU"]), 4)
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
            rec["hbm_bytes_per_step"] = (2.0 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024.0      # KiB; x2: gfx950 FETCH_SIZE correction
        if "SQ_WAIT_ANY" in r and "SQ_WAVE_CYCLES" in r:
            rec["wait_any_over_wave_cycles"] = round(r["SQ_WAIT_ANY"] / r["SQ_WAVE_CYCLES"], 4)
        q = s.get("k_resolve", {})
        if "FETCH_SIZE" in q and "WRITE_SIZE" in q:
            rec["resolve_hbm_bytes_per_step"] = (2.0 * q["FETCH_SIZE"] + q["WRITE_SIZE"]) * 1024.0
        doc["workloads"][wl] = rec
        json.dump(doc, open(os.path.join(args.out, "pmc_counters.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({w: {k: v for k, v in r.items() if k != "raw"} for w, r in doc["workloads"].items()}, indent=1))
    if failed:
        print(f"workloads without counters: {failed}", flush=True)
        sys.exit(2)


if __name__ == "__main__":
    main()
