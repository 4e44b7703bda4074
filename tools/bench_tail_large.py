#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: the fusion tail on a 64 x 128 token grid, E = 128, 48 (or 9) conv outputs.
256 frames over 8 GPUs = 32 frames per GPU (weak scaling: every GPU runs this same launch).
Prints one JSON line: ms per launch, achieved algorithmic TFLOP/s against the dense f16 MFMA peak, and the
algorithmic HBM bytes (f32 tokens in, f32 map out) per second against 8 TB/s.

usage: python tools/bench_tail_large.py [--frames 32] [--out-ch 48] [--iters 20]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--out-ch", type=int, default=48)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    import torch
    from drone_oa_iree_vit_accelerator_amd import host, synth
    E, th, tw, co, B = 128, 64, 128, a.out_ch, a.frames
    c = synth.tail_large_case(0, E, th, tw, co, 1)
    eng = host.FusionTailLarge(c["conv_w"], c["conv_b"], device=0)
    x = torch.randn((B, th * tw, E), device="cuda")
    out = torch.empty((B, co, 2 * th, 2 * tw), device="cuda")
    for _ in range(3):
        eng(x, th, tw, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        eng(x, th, tw, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    macs = (E // 4 + E) * 9 * co * (2 * th) * (2 * tw)          # per frame
    flops = 2.0 * macs * B
    byts = (th * tw * E * 4 + co * 4 * th * tw * 4) * B
    print(json.dumps({"workload": f"fusion tail 64x128 tokens, E=128, out_ch={co}, {B} frames", "ms": round(ms, 4),
                      "frames_per_s": round(B / ms * 1e3, 1), "algorithmic_TFLOPs": round(flops / ms / 1e9, 2),
                      "frac_f16_mfma_peak_2500": round(flops / ms / 1e9 / 2500.0, 4),
                      "executed_TFLOPs_f16x3": round(3 * flops * (((co + 15) // 16) * 16 / co) / ms / 1e9, 2),
                      "algorithmic_GBs": round(byts / ms / 1e6, 1), "frac_hbm_peak_8000": round(byts / ms / 1e6 / 8000.0, 4)}))
    eng.close()


if __name__ == "__main__":
    main()
