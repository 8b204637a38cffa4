#!/bin/bash
# 1/8-image launches (what each of 8 GPUs runs at the headline size) on the GPU box: tools/gpu_tail.sh <tag> [parts]
set -u -o pipefail
TAG=${1:-tail}; PARTS=${2:-8}; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
for WL in cornell-box-800x600x256-d30 teapot-800x600x256-d64 semesterbild-800x600x256-d30 veach-mis-1280x720x1024-d16; do
  python3 bench.py --workload $WL --tail-parts $PARTS --cpu-seconds 0 > "$OUT/tail_$WL.json" 2> "$OUT/tail_$WL.err" || { cat "$OUT/tail_$WL.err"; exit 1; }
  python3 - "$OUT/tail_$WL.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); t = d["tail"]
print(f'{d["config"]["workload"]:34s} step {d["ms_per_step"]:8.3f} ms  kernel {d["roofline"]["kernel_ms_per_step"]:8.3f}  1/{t["parts"]}: ideal {t["ideal_render_ms"]:.3f} max {t["render_ms_max"]:.3f} eff {t["tail_efficiency"]:.3f} | two streams: {t["two_streams"]["ms_per_frame"]:.3f} ms/frame vs ideal {t["two_streams"]["ideal_ms_per_frame"]:.3f} -> {t["two_streams"]["efficiency"]:.3f} | {t["frames_in_flight"]["frames"]} frames in flight on 1/{t["frames_in_flight"]["share_of_device_per_frame"]} each: {t["frames_in_flight"]["ms_per_frame"]:.3f} -> {t["frames_in_flight"]["efficiency"]:.3f}')
PY
done
