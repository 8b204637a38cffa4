#!/bin/bash
# Full evidence pass on the GPU box (run through gpurun): gpu tests, one bench line per BASELINE config, the rocprofv3
# kernel-trace summary of the headline command and the PMC passes bench.py's roofline is computed from.
# usage: tools/gpu_evidence.sh <tag>   -> everything under gpurun_out/<tag>/
set -u -o pipefail
TAG=${1:-ev}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1 || { tail -30 "$OUT/pytest_gpu.log"; exit 1; }
tail -3 "$OUT/pytest_gpu.log"
if [ "${SKIP_PMC:-0}" != "1" ]; then     # (SKIP_PMC=1: profiles/pmc_counters.json already belongs to these kernel sources -- bench.py says "committed (hash-matched)" or "stale")
python3 tools/pmc_collect.py --out "$OUT/pmc" > "$OUT/pmc_collect.log" 2>&1 || { tail -30 "$OUT/pmc_collect.log"; exit 1; }
cp "$OUT/pmc/pmc_counters.json" profiles/pmc_counters.json
fi
for WL in cornell-box-800x600x256-d30 teapot-800x600x256-d64 veach-mis-1280x720x1024-d16 semesterbild-800x600x256-d30; do
  python3 bench.py --workload $WL --tail-parts 8 > "$OUT/bench_$WL.json" 2> "$OUT/bench_$WL.err" || { cat "$OUT/bench_$WL.err"; exit 1; }
  echo "$WL done"
done
python3 bench.py --workload semesterbild-1920x1080x4096-d30 --steps 3 --warmup 1 > "$OUT/bench_semesterbild-1920x1080x4096-d30.json" 2> "$OUT/bench_cfg5.err" || { cat "$OUT/bench_cfg5.err"; exit 1; }
python3 bench.py --pipeline 4 --share 4 --cpu-seconds 0 > "$OUT/bench_cornell_pipeline4_share4.json" 2>&1 || { cat "$OUT/bench_cornell_pipeline4_share4.json"; exit 1; }
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/stats" -- python3 "$OLDPWD/bench.py" --steps 20 --warmup 3 --cpu-seconds 0 > "$OLDPWD/$OUT/stats.log" 2>&1) || { tail -20 "$OUT/stats.log"; exit 1; }
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_cornell.csv" \;
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/stats_teapot" -- python3 "$OLDPWD/bench.py" --workload teapot-800x600x256-d64 --steps 20 --warmup 3 --cpu-seconds 0 > "$OLDPWD/$OUT/stats_teapot.log" 2>&1) || { tail -20 "$OUT/stats_teapot.log"; exit 1; }
find "$OUT/stats_teapot" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_teapot.csv" \;
python3 tools/ab_fuzz_scene.py 31 32 33 > "$OUT/ab_fuzz_scene.txt" 2>&1 || { tail -5 "$OUT/ab_fuzz_scene.txt"; exit 1; }
python3 tools/stamps.py > "$OUT/stamps.txt" 2>&1 || { tail -5 "$OUT/stamps.txt"; exit 1; }
python3 bench.py --workload cornell-box-400x300x16-d4 --steps 50 --warmup 5 > "$OUT/bench_cornell-box-400x300x16-d4.json" 2> "$OUT/bench_cfg1.err" || { cat "$OUT/bench_cfg1.err"; exit 1; }
# `python bench.py --gpus N` as typed: the self-launch (one RCCL rank; two rehearsal ranks on this one GPU over gloo) and the one-process fallback form
MI355RT_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1 --cpu-seconds 0 > "$OUT/bench_selflaunch_rccl_1rank.json" 2> "$OUT/selflaunch1.err" || { cat "$OUT/selflaunch1.err"; exit 1; }
MI355RT_BENCH_REHEARSE=1 python3 bench.py --gpus 2 --cpu-seconds 0 > "$OUT/bench_selflaunch_rehearsal_2ranks.json" 2> "$OUT/selflaunch2.err" || { cat "$OUT/selflaunch2.err"; exit 1; }
MI355RT_BENCH_REHEARSE=1 python3 bench.py --gpus 2 --single-process --cpu-seconds 0 > "$OUT/bench_single_process_rehearsal_2parts.json" 2> "$OUT/single2.err" || { cat "$OUT/single2.err"; exit 1; }
# where the render kernel's WRITE_SIZE comes from: 32-byte vs 64-byte write requests of the L2 (headline workload only)
python3 tools/pmc_collect.py --out "$OUT/pmc_writes" --workloads cornell-box-800x600x256-d30 --groups '[["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"]]' > "$OUT/pmc_writes.log" 2>&1 || tail -3 "$OUT/pmc_writes.log"
python3 tools/isa_stats.py > "$OUT/isa_stats.txt" 2>&1
cat "$OUT/bench_cornell-box-800x600x256-d30.json"
