#!/bin/bash
# camshift A/B: fused / chunked schedule x LDS-cached region on / off (C3), and the C5 streaming line
pr() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('kernel_ms_per_track_call') or d.get('device_ms_per_30_frame_cycle'), d.get('latency_ms',''))"; }
python -m pytest tests/test_gpu_camshift.py tests/test_gpu_shapes.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | grep -E "^E |passed|failed" | head
python bench.py --workload c3 --steps 8 --cpu-seconds 0 --no-sub 2>/dev/null | pr "c3 fused region"
HT_DEBUG_CS_REGION=0 python bench.py --workload c3 --steps 8 --cpu-seconds 0 --no-sub 2>/dev/null | pr "c3 fused noregion"
HT_DEBUG_CS_FUSED_MIN=100000 python bench.py --workload c3 --steps 8 --cpu-seconds 0 --no-sub 2>/dev/null | pr "c3 chunked region"
HT_DEBUG_CS_FUSED_MIN=100000 HT_DEBUG_CS_REGION=0 python bench.py --workload c3 --steps 8 --cpu-seconds 0 --no-sub 2>/dev/null | pr "c3 chunked noregion"
python bench.py --workload c5 --cpu-seconds 0 2>/dev/null | pr "c5"
