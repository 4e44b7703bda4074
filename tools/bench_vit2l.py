#!/usr/bin/env python3
"""Step time of the SECOND graph family (models/ITA/QAT/model.py:22-87: E = 128, two encoder layers, no fusion tail) on the
GPU box: frames/s and, under rocprofv3 --kernel-trace --stats, its kernels.  usage: python tools/bench_vit2l.py [frames]"""
import glob, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drone_oa_iree_vit_accelerator_amd import host, params, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
path = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "vit2l_*.npz")))[0]
d = params.load_fixture(path)
fp = synth.float_params(int(d["meta.seed"]), E=128, num_layers=2, tail=False)
eng = host.Engine(params.blob_from_record(d, fp, E=128, num_layers=2), device=0, reserve=B)
fr = synth.frames(11, B)
img, dv, qt = (torch.from_numpy(fr[k]).cuda() for k in ("img_u8", "desvel", "quat"))
state = [(torch.zeros((3, B, 128), device="cuda"), torch.zeros((3, B, 128), device="cuda")) for _ in range(2)]
vel = torch.empty((B, 3), device="cuda")
for i in range(20):
    eng.forward(img, dv, qt, state[i & 1], out=(vel, *state[(i + 1) & 1]))
torch.cuda.synchronize()
K = 100
t0 = time.perf_counter()
for i in range(K):
    eng.forward(img, dv, qt, state[i & 1], out=(vel, *state[(i + 1) & 1]))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"E=128 two-layer no-tail graph, {B} frames per step: {dt * 1e3:.4f} ms per step, {B / dt / 1e6:.2f} M frames/s")
eng.close()
