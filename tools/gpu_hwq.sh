# frames in flight on probed streams, default hardware queues and GPU_MAX_HW_QUEUES=8
echo "== default HW queues"; bash tools/gpu_inflight.sh gpurun_out/r05_hwq_a 1x1,2x2,4x4,4x2 && bash tools/gpu_tail.sh r05_tail 8
echo "== GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 bash tools/gpu_inflight.sh gpurun_out/r05_hwq_b 4x4,6x6,8x8,8x4,6x3
