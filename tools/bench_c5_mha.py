#!/usr/bin/env python3
"""configs.c5_mha of bench.py alone (GPU box): python tools/bench_c5_mha.py [frames] [seq]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.bench_c5_mha(int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 8192)))
