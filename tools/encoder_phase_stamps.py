#!/usr/bin/env python3
"""Diagnostic (GPU box): where does a frame spend its cycles inside ita_encoder_kernel?
Prints the median s_memtime delta per phase over all workgroups (frames 1..3 of each)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drone_oa_iree_vit_accelerator_amd import host, params, synth

d = params.load_fixture(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "vitlstm_E64_seed0_B2.npz"))
eng = host.Engine(params.blob_from_record(d, synth.float_params(0), E=64), device=0)
B = 1024
fused = len(sys.argv) > 1 and sys.argv[1] == "u8"      # u8 frames: tokenizer fused in front
x = torch.from_numpy(synth.frames(0, B)["img_u8"]).cuda() if fused else torch.randn((B, 128, 64), device="cuda")
for _ in range(3):
    st = eng.encoder_stamps(x)
torch.cuda.synchronize()
st = st.cpu().numpy().astype(np.int64)
names = ["quant", "P(qkv proj)", "A(attn)", "O(out_proj)", "L1(ln1+quant)", "F1(fc1)", "F2(fc2)", "L2(ln2+store)"]
frames = st[:, 1:4, :]                       # steady-state frames
dt = np.diff(frames[..., :9], axis=-1)       # 8 phases
print("s_memtime ticks (100 MHz constant clock? -> see total) per phase, median over WGs x frames")
for i, n in enumerate(names):
    print(f"  {n:16s} median {np.median(dt[..., i]):9.0f}  p90 {np.percentile(dt[..., i], 90):9.0f}")
tot = frames[..., 8] - frames[..., 0]
print(f"  frame total      median {np.median(tot):9.0f}")
w0, w4 = frames[..., 13] - frames[..., 1], frames[..., 14] - frames[..., 1]
# the two waves share a SIMD: what counts is their sum (the phase is VALU-issue bound), not their balance
print(f"  phase P work: wave 0 (11 Q/K tiles) median {np.median(w0):7.0f}, wave 4 (1 K + 6 V tiles) median {np.median(w4):7.0f}")
if fused:
    tk = frames[..., [8, 9, 10, 11, 12]]
    for i, n in enumerate(["T0+T1 image/weights -> LDS", "T2 blend", "T3 MFMA", "T4 LayerNorm"]):
        d = tk[..., i + 1] - tk[..., i]
        print(f"  tok {n:27s} median {np.median(d):9.0f}  p90 {np.percentile(d, 90):9.0f}")
whole = st[:, 3, 8] - st[:, 0, 0]
print(f"  4 frames (wg)    median {np.median(whole):9.0f}; first-frame start spread {st[:, 0, 0].max() - st[:, 0, 0].min()}")
