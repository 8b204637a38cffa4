#!/bin/bash
# Build everything locally first; only then spend a gpurun call.  usage: tools/gpu.sh [--timeout S] '<command>'
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
import importlib, sys, json, os
sys.path.insert(0, '.')
b = importlib.import_module('raytracer-rust_amd.build')
b.build_device(); b.build_host(); b.build_cli()
for k, (d, f) in json.loads(os.environ.get('AB_VARIANTS', '{}')).items():
    b.build_device_variant('ab_' + k, d, flags=f)
if os.environ.get('BUILD_STAMPS'):
    b.build_device_variant('stamps', ['MI355RT_STAMPS'])
PY
T=600; if [ "$1" = "--timeout" ]; then T=$2; shift 2; fi
timeout 3400 /usr/local/graft/bin/gpurun --timeout $T -- "$1"
