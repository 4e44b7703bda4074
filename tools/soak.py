#!/usr/bin/env python3
"""Soak test on the GPU box: N forwards of the full 1024-frame batch from the same inputs and state; every result must
equal the first one bit for bit (catches intermittent races in the persistent encoder kernel, the split-K
reduction or the slot path).  usage: python tools/soak.py [iterations]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    import numpy as np
    import torch
    from drone_oa_iree_vit_accelerator_amd import host, params, synth
    fx = params.load_fixture(os.path.join(REPO, "tests", "golden", "vitlstm_E64_seed0_B2.npz"))
    eng = host.Engine(params.blob_from_record(fx, synth.float_params(0, E=64), E=64), device=0, reserve=1024)
    B = 1024
    fr = synth.frames(5, B)
    img, dv, qt = (torch.from_numpy(fr[k]).cuda() for k in ("img_u8", "desvel", "quat"))
    rs = np.random.RandomState(9)
    h = torch.from_numpy(rs.standard_normal((3, B, 128)).astype(np.float32) * 0.1).cuda()
    c = torch.from_numpy(rs.standard_normal((3, B, 128)).astype(np.float32) * 0.1).cuda()
    out = [(torch.empty((B, 3), device="cuda"), torch.empty_like(h), torch.empty_like(c)) for _ in range(2)]
    eng.forward(img, dv, qt, (h, c), out=out[0])
    torch.cuda.synchronize()
    ref = [t.clone() for t in out[0]]
    bad, t0 = 0, time.time()
    for i in range(n):
        o = out[i & 1]
        eng.forward(img, dv, qt, (h, c), out=o)
        if i % 50 == 49:
            if not all(torch.equal(a, b) for a, b in zip(o, ref)):
                bad += 1
        if i % 5000 == 4999:
            torch.cuda.synchronize()
            print(f"{i + 1} forwards, {bad} mismatching checks, {time.time() - t0:.1f} s", flush=True)
    torch.cuda.synchronize()
    print("soak:", "OK" if bad == 0 else f"{bad} MISMATCHES", f"({n} forwards of {B} frames)")
    # the library's two-thread, two-stream pipeline (ita_vitlstm_pipelined): 12 time steps of 130 streams, repeated; every
    # call must reproduce the sequential forwards bit for bit (host-side event ordering between the two threads)
    Bp, steps, calls = 130, 12, max(20, n // 100)
    frs = [synth.frames(700 + t, Bp) for t in range(steps)]
    imgs = [torch.from_numpy(f["img_u8"]).cuda() for f in frs]
    dvs = [torch.from_numpy(f["desvel"]).reshape(Bp).cuda() for f in frs]
    qts = [torch.from_numpy(f["quat"]).cuda() for f in frs]
    h0, c0 = h[:, :Bp].contiguous(), c[:, :Bp].contiguous()
    st, want = (h0, c0), []
    for t in range(steps):
        v, st = eng.forward(imgs[t], dvs[t], qts[t], st)
        want.append(v.clone())
    sf, sb = torch.cuda.Stream(), torch.cuda.Stream()
    badp = 0
    for _ in range(calls):
        hh, cc = h0.clone(), c0.clone()
        vels = [torch.empty((Bp, 3), device="cuda") for _ in range(steps)]
        torch.cuda.synchronize()
        eng.pipelined(imgs, dvs, qts, (hh, cc), vels, sf, sb)
        sf.synchronize()
        if not (all(torch.equal(vels[t], want[t]) for t in range(steps)) and torch.equal(hh, st[0]) and torch.equal(cc, st[1])):
            badp += 1
    print("pipelined soak:", "OK" if badp == 0 else f"{badp} MISMATCHES", f"({calls} calls of {steps} steps, {Bp} frames)")
    eng.close()
    sys.exit(1 if (bad or badp) else 0)


if __name__ == "__main__":
    main()
