set -u -o pipefail
O=gpurun_out/r02e; mkdir -p $O
echo skip-tests

MI355RT_BENCH_REHEARSE=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 5 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 5 --steps 4 --warmup 2 --cpu-seconds 0 > $O/rehearse5.json 2> $O/rehearse5.err || { tail -30 $O/rehearse5.err; exit 1; }
python3 bench.py --steps 10 --cpu-seconds 0 > $O/bench1.json 2>$O/bench1.err || { cat $O/bench1.err; exit 1; }
python3 - <<'PY'
import json
for f in ("rehearse5","bench1"):
    d=json.loads(open(f"gpurun_out/r02e/{f}.json").read().strip().splitlines()[-1])
    print(f, d["n_gpus"], d["value"], d["ms_per_step"], d["image_checksum"], d["config"]["parallelism"], d["config"]["frames_in_flight"])
PY
