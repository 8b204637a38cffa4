set -u -o pipefail
mkdir -p gpurun_out/r02b; O=gpurun_out/r02b
echo "cgroup: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) | v1: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) | nproc $(nproc)" > $O/cgroup.txt
python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
for WL in semesterbild-800x600x256-d30 teapot-800x600x256-d64 cornell-box-800x600x256-d30; do
  python3 bench.py --workload $WL --tail-parts 8 --cpu-seconds 3 > $O/bench_$WL.json 2> $O/bench_$WL.err || { cat $O/bench_$WL.err; exit 1; }
done
python3 tools/wave_timeline.py semesterbild > $O/wave_semesterbild.txt 2>&1 || { tail $O/wave_semesterbild.txt; exit 1; }
python3 tools/wave_timeline.py teapot > $O/wave_teapot.txt 2>&1 || true
cat $O/cgroup.txt; cat $O/wave_semesterbild.txt
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02b/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('bench_')[1], d['value'], d['ms_per_step'], r['kernel_ms_per_step'], d.get('tail',{}).get('render_ms_max'), d['cpu_baseline'])
PY
