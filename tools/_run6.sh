set -u -o pipefail
T=${1:-r02h}; O=gpurun_out/$T; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_fuzz_parity.py tests/test_oracle_quirks.py tests/test_fixtures.py -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
for WL in ${2:-semesterbild-800x600x256-d30 teapot-800x600x256-d64}; do
  python3 bench.py --workload $WL --cpu-seconds 0 --steps 10 > $O/bench_$WL.json 2> $O/bench_$WL.err || { cat $O/bench_$WL.err; exit 1; }
done
python3 - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('bench_')[1], d['value'], d['ms_per_step'], r['kernel_ms_per_step'], d['image_checksum'])
PY
