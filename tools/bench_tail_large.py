#!/usr/bin/env python3
"""BASELINE config 5 alone (the leg bench.py reports as configs.c5): one JSON line.  tools/profile_gpu.sh runs this under
rocprofv3 to get the kernel stats / PMC of ita_tail_big_kernel."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.bench_c5(int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 48)))
