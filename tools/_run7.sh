set -u -o pipefail
O=gpurun_out/r02final4; mkdir -p $O
for WL in cornell-box-800x600x256-d30 teapot-800x600x256-d64 veach-mis-1280x720x1024-d16 semesterbild-800x600x256-d30; do
  python3 bench.py --workload $WL --tail-parts 8 > "$O/bench_$WL.json" 2> "$O/bench_$WL.err" || { cat "$O/bench_$WL.err"; exit 1; }
done
python3 bench.py --workload semesterbild-1920x1080x4096-d30 --steps 3 --warmup 1 > "$O/bench_semesterbild-1920x1080x4096-d30.json" 2> "$O/bench_cfg5.err" || { cat "$O/bench_cfg5.err"; exit 1; }
python3 bench.py --pipeline 2 --cpu-seconds 0 > "$O/bench_cornell_pipeline2.json" 2>/dev/null
python3 tools/stamps.py > $O/stamps.txt 2>&1
python3 tools/wave_timeline.py semesterbild > $O/wave_semesterbild.txt 2>&1 || true
grep -h '"value"' $O/bench_*.json | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(d['config']['workload'], d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['pmc'])"
