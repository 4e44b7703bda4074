#!/usr/bin/env python3
"""Step time of host.PipelinedSteps with two stages (front | back) and three (encoder | folded GEMM | LSTM + fc), per
frames-per-GPU.  usage (GPU box): python tools/pipeline3_probe.py [frames ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import drone_oa_iree_vit_accelerator_amd as pkg
from drone_oa_iree_vit_accelerator_amd import host, params, synth

fx = params.load_fixture(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "vitlstm_E64_seed0_B2.npz"))
blob = params.blob_from_record(fx, synth.float_params(0, E=64), E=64)
for B in [int(a) for a in sys.argv[1:]] or [1, 64, 128, 256]:
    row = []
    for stages in (2, 3, 2, 3):
        eng = host.Engine(blob, device=0, reserve=B)
        ps = eng.pipelined_steps(B, 8, stages)
        fr = synth.frames(7, B)
        ps.img.copy_(torch.from_numpy(fr["img_u8"]).cuda().unsqueeze(0).expand(8, -1, -1, -1))
        for _ in range(100):
            ps()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            ps()
        torch.cuda.synchronize()
        row.append((stages, (time.perf_counter() - t0) / 1600 * 1e6))
        del ps
        eng.close()
    print(f"frames {B:5d}: " + "  ".join(f"{s} stages {us:6.2f} us/step" for s, us in row), flush=True)
