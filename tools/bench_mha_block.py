#!/usr/bin/env python3
"""BASELINE config 2 (int8 MHA block alone) through ita_mha_int8: ms per launch and achieved fraction of the
int8 MFMA peak (5.0 POP/s dense) for the two model widths.  Algorithmic ops per frame:
E=128: 37.75 MOP, E=64: 25.17 MOP.

usage: python tools/bench_mha_block.py [--frames 1024] [--iters 50]
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    import torch
    from drone_oa_iree_vit_accelerator_amd import host, params
    for E, fixture, mop in ((128, "blocks_E128_seed0_B1.npz", 37.75e6), (64, "blocks_E64_seed2_B1.npz", 25.17e6)):
        d = params.load_fixture(os.path.join(REPO, "tests", "golden", fixture))
        eng = host.Engine(params.blob_from_record(d, None, E=E), device=0)
        x = torch.randn((a.frames, 128, E), device="cuda")
        for _ in range(5):
            eng.mha(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            eng.mha(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        tops = mop * a.frames / ms / 1e9
        print(json.dumps({"workload": f"int8 MHA block E={E}, {a.frames} frames", "ms": round(ms, 4),
                          "frames_per_s": round(a.frames / ms * 1e3, 1), "achieved_TOPs": round(tops, 1),
                          "frac_int8_mfma_peak_5000": round(tops / 5000.0, 4)}))
        eng.close()


if __name__ == "__main__":
    main()
