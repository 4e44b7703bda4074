#!/usr/bin/env python3
"""Replay PNG + data.csv trajectories through the MI355X engine (the reference's
samples/inference_trainingset_custom_dispatch/main.cpp).  Prints one line per frame like the reference
(model output, ground truth, error distance) and a per-trajectory summary.

  python tools/replay_trajectories.py --root <dir of trajectory dirs> --blob weights.itaw
  python tools/replay_trajectories.py --root <dir> --synthetic-weights      # seed-0 synthetic QAT weights (no checkpoint ships)
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", required=True)
    ap.add_argument("--blob")
    ap.add_argument("--synthetic-weights", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--quiet", action="store_true", help="summary only")
    a = ap.parse_args()
    from drone_oa_iree_vit_accelerator_amd import host, params, replay, synth
    if a.blob:
        blob = open(a.blob, "rb").read()
    elif a.synthetic_weights:
        fx = params.load_fixture(os.path.join(REPO, "tests", "golden", "vitlstm_E64_seed0_B2.npz"))
        blob = params.blob_from_record(fx, synth.float_params(0, E=64), E=64)
    else:
        raise SystemExit("give --blob or --synthetic-weights")
    eng = host.Engine(blob, device=a.device)
    res = replay.replay(eng, a.root)
    if not a.quiet:
        for r in res:
            print(f"{r.trajectory}/{r.frame}  Model Output: [{r.output[0]:.6g}, {r.output[1]:.6g}, {r.output[2]:.6g}]  "
                  f"Ground Truth Vel: [{r.ground_truth[0]:.6g}, {r.ground_truth[1]:.6g}, {r.ground_truth[2]:.6g}]  "
                  f"Error Distance: {r.error:.6g}" + ("" if r.telemetry_found else "  (no telemetry row: defaults)"))
    print(json.dumps(replay.summarize(res)))
    eng.close()


if __name__ == "__main__":
    main()
