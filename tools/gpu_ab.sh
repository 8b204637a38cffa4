#!/bin/bash
# A/B of prebuilt device-library variants on the GPU box (run through gpurun): tools/gpu_ab.sh <tag> "<name1> <name2> ..." [workloads...]
# Each <name> is raytracer-rust_amd/_build/libmi355rt_ab_<name>.so, built beforehand in the container (build.build_device_variant).
set -u
TAG=$1; NAMES=$2; shift 2; WLS=${*:-semesterbild teapot}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
PRE=""; for n in $NAMES; do PRE="$PRE,$n=raytracer-rust_amd/_build/libmi355rt_ab_$n.so"; done; PRE=${PRE#,}
for SPP in ${AB_SPPS:-64 256}; do
  AB_SPP=$SPP AB_VARIANTS='{}' AB_PREBUILT=$PRE timeout -k 10 ${AB_TIMEOUT:-300} python3 tools/ab.py $WLS > "$OUT/ab$SPP.txt" 2>&1 || { echo "ab.py failed at $SPP spp"; tail -5 "$OUT/ab$SPP.txt"; exit 1; }
  echo "## $SPP spp"; grep -v amdgpu.ids "$OUT/ab$SPP.txt"
done
