#!/usr/bin/env python3
"""Timeline of ONE replay of host.PipelinedSteps (8 time steps on three graph branches) from a rocprofv3 kernel trace.
  run  : rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/graph_timeline.py run [frames] [stages]
  show : python3 tools/graph_timeline.py show OUT
Prints every kernel of the last replay with start / end (us from the replay's first kernel) and how many other kernels of
the replay were running when it started -- which branch waits for which is read off the gaps."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(frames, stages):
    import torch
    from drone_oa_iree_vit_accelerator_amd import host, params, synth
    fx = params.load_fixture(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "vitlstm_E64_seed0_B2.npz"))
    eng = host.Engine(params.blob_from_record(fx, synth.float_params(0, E=64), E=64), device=0, reserve=frames)
    ps = eng.pipelined_steps(frames, 8, stages)
    fr = synth.frames(7, frames)
    ps.img.copy_(torch.from_numpy(fr["img_u8"]).cuda().unsqueeze(0).expand(8, -1, -1, -1))
    for _ in range(30):
        ps()
    torch.cuda.synchronize()


def show(d):
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f))]
    rows = [r for r in rows if r[2].startswith(("void ita_", "ita_"))]
    rows.sort()
    # a replay = 8 encoder launches; take the last complete one
    enc = [i for i, r in enumerate(rows) if "ita_stream_kernel" in r[2]]
    a = enc[-8]
    rep = rows[a:]
    t0 = rep[0][0]
    short = lambda n: n.replace("void ", "").replace("ita_", "")[:34]
    print(f"{'kernel':36s} {'start':>8s} {'end':>8s} {'dur':>7s}  running at its start")
    for s, e, n in rep:
        running = [short(m) for (s2, e2, m) in rep if s2 < s < e2]
        print(f"{short(n):36s} {(s - t0) / 1e3:8.2f} {(e - t0) / 1e3:8.2f} {(e - s) / 1e3:7.2f}  {', '.join(running)}")
    print(f"replay span {(max(r[1] for r in rep) - t0) / 1e3:.1f} us for 8 steps; sum of kernel durations {sum(e - s for s, e, _ in rep) / 1e3:.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 128, int(sys.argv[3]) if len(sys.argv) > 3 else 3)
    else:
        show(sys.argv[2])
