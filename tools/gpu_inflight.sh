# frames in flight x grid share, per workload (bench.py --inflight): usage  bash tools/gpu_inflight.sh <out dir> "<FxD list>" [workloads...]
set -e
OUT=${1:-gpurun_out/r05_inflight}; SPEC=${2:-1x1,2x1,2x2,3x3,4x4,4x2}; shift 2 || true
WLS=${*:-cornell-box-800x600x256-d30 teapot-800x600x256-d64 semesterbild-800x600x256-d30 veach-mis-1280x720x1024-d16}
mkdir -p $OUT
for wl in $WLS; do
  timeout -k 10 400 python bench.py --workload $wl --steps 12 --warmup 2 --cpu-seconds 0 --no-one-shot --tail-parts 8 --inflight $SPEC > $OUT/$wl.json
  python - "$OUT/$wl.json" "$wl" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
t=d["tail"]; print(sys.argv[2], "step", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms_per_step"], "| 1/8 one stream worst", t["render_ms_max"], "ideal", t["ideal_render_ms"], "eff", t["tail_efficiency"], "| two streams", t["two_streams"]["ms_per_frame"], "ideal", t["two_streams"]["ideal_ms_per_frame"], t["two_streams"]["efficiency"])
ideal=t["two_streams"]["ideal_ms_per_frame"]
for r in d["inflight"]: print(f"    {r['frames_in_flight']}x{r['grid_div']} ({r.get('distinct_queues')} queues)  full {r['ms_per_frame_full']:8.3f} ms   1/8 {r['ms_per_frame_1/8']:7.3f} ms = {ideal / r['ms_per_frame_1/8']:.3f} of ideal")
PY
done
