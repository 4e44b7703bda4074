#!/usr/bin/env python3
"""configs.b1 / b128 / b256 of bench.py alone (GPU box): python tools/bench_small.py [frames ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from drone_oa_iree_vit_accelerator_amd import params, synth
fx = params.load_fixture(os.path.join(bench.REPO, "tests", "golden", "vitlstm_E64_seed0_B2.npz"))
blob = params.blob_from_record(fx, synth.float_params(0, E=64), E=64)
for k, v in bench.bench_small(blob, tuple(int(a) for a in sys.argv[1:]) or (1, 64, 128, 256)).items():
    print(k, v["ms_per_step"], v["ms_per_step_one_stream"])
