#!/usr/bin/env python3
"""Diagnostic (GPU box): where does a frame spend its cycles inside ita_stream_kernel?
Prints the median s_memtime delta per phase (waves 0 and 4) over all workgroups (frames 1..3 of each)."""
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
st = st.cpu().numpy().astype(np.int64)      # [blocks, 8, 2, 16]
# slots: 0 frame start | 1 projections done (before B1) | 2 after B1 | 3 logits | 4 softmax | 5 AV (before B2) | 6 after B2
#        | 7 out_proj + LN1 | 8 fc1 | 11 fc2 + LN2 + store | 9 tokenizer blend | 10 tokenizer MFMA + LN
seq = [0, 1, 2, 3, 4, 5, 6, 7, 8, 11] + ([9, 10] if fused else [])
names = ["quant + Q/K/V proj", "wait B1", "logits", "softmax", "AV", "wait B2", "out_proj + LN1", "fc1", "fc2 + LN2 + store"] + \
        (["tok blend", "tok MFMA + LN"] if fused else [])
for wv in (0, 1):
    fr = st[:, 1:4, wv, :]
    print(f"wave {4 * wv}: s_memtime ticks per phase, median / p90 over WGs x frames")
    for i, n in enumerate(names):
        dt = fr[..., seq[i + 1]] - fr[..., seq[i]]
        print(f"  {n:20s} {np.median(dt):9.0f} {np.percentile(dt, 90):9.0f}")
    nxt = st[:, 2:5, wv, 0] - st[:, 1:4, wv, 0]
    print(f"  frame period         {np.median(nxt):9.0f} {np.percentile(nxt, 90):9.0f}")
# constant-clock (100 MHz) timeline of the launch: entry / prologue end / exit of every workgroup
t0 = st[:, 0, :, 12].min()
ent, pro, ext = st[:, 0, :, 12] - t0, st[:, 0, :, 13] - t0, st[:, 0, :, 14] - t0
us = lambda v: v / 100.0
print(f"launch timeline (us from the first workgroup's entry): entry median {us(np.median(ent)):.2f} max {us(ent.max()):.2f} | "
      f"prologue end median {us(np.median(pro)):.2f} max {us(pro.max()):.2f} | exit median {us(np.median(ext)):.2f} max {us(ext.max()):.2f}")
print(f"prologue length median {us(np.median(pro - ent)):.2f} us p90 {us(np.percentile(pro - ent, 90)):.2f}; frames part median {us(np.median(ext - pro)):.2f} us")
whole = st[:, 3, 0, 11] - st[:, 0, 0, 0]
print(f"4 frames (wg) median {np.median(whole):9.0f}; first-frame start spread {st[:, 0, 0, 0].max() - st[:, 0, 0, 0].min()}")
