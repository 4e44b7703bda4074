#!/usr/bin/env python3
"""BASELINE config 2 alone (the leg bench.py reports as configs.c2): one JSON line.  tools/profile_gpu.sh runs this under
rocprofv3 to get the kernel stats / PMC of ita_stream_kernel<128, false, 0, false>."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.bench_c2(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)))
