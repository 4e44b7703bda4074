#!/usr/bin/env python3
"""Diagnostic (GPU box): what does the two-stream HIP graph of host.PipelinedSteps buy?  Per time step, from 8-step graphs:
front only (encoder + folded GEMM), back only (LSTM + fc), both on ONE stream, both on two streams (PipelinedSteps),
and the library's own two-stream loop (ita_vitlstm_pipelined), 8 and 40 steps per call.
usage: python tools/pipeline_probe.py [frames]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drone_oa_iree_vit_accelerator_amd import host, params, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
d = params.load_fixture(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "vitlstm_E64_seed0_B2.npz"))
eng = host.Engine(params.blob_from_record(d, synth.float_params(0), E=64), device=0)
N = 8
pipe = host.PipelinedSteps(eng, B, N)
fr = synth.frames(5, B)
for i in range(N):
    pipe.img[i].copy_(torch.from_numpy(fr["img_u8"]))


def capture(kind):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=s):
        for i in range(N):
            if kind in ("front", "serial"):
                eng.front(pipe.img[i], i & 1, stream=s)
            if kind in ("back", "serial"):
                eng.back(pipe.desvel[i], pipe.quat[i], (pipe.h, pipe.c), (pipe.vel[i], pipe.h, pipe.c), i & 1, stream=s)
    return g


def timeit(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps / N


res = {}
for kind in ("front", "back", "serial"):
    g = capture(kind)
    res[kind] = timeit(g.replay)
res["two streams"] = timeit(pipe)
sf, sb = torch.cuda.Stream(), torch.cuda.Stream()
imgs = [pipe.img[i] for i in range(N)]; dvs = [pipe.desvel[i] for i in range(N)]; qts = [pipe.quat[i] for i in range(N)]
vels = [pipe.vel[i] for i in range(N)]
# the library loop works on ITS two streams: time it on the front stream, where it joins (events on another stream would
# only see the host's enqueue time)
with torch.cuda.stream(sf):
    res["C++ loop"] = timeit(lambda: eng.pipelined(imgs, dvs, qts, (pipe.h, pipe.c), vels, sf, sb))
    M = 40
    args40 = ([pipe.img[0]] * M, [pipe.desvel[0]] * M, [pipe.quat[0]] * M, (pipe.h, pipe.c), [pipe.vel[i % N] for i in range(M)])
    res["C++ loop x40"] = timeit(lambda: eng.pipelined(*args40, sf, sb), reps=20) * N / M
print(f"{B} frames per step; us per time step from {N}-step HIP graphs")
for k, v in res.items():
    print(f"  {k:12s} {v:7.2f}")
eng.close()

# ---- the same library loop on streams with disjoint CU masks (front: every even CU bit, back: every odd one)
if os.environ.get("ITA_PROBE_CUMASK"):
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    def masked_stream(word):
        st = C.c_void_p()
        mask = (C.c_uint32 * 8)(*([word] * 8))
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, mask)
        assert rc == 0, rc
        return torch.cuda.ExternalStream(st.value)
    eng2 = host.Engine(params.blob_from_record(d, synth.float_params(0), E=64), device=0, reserve=B)
    for name, (wa, wb) in {"even | odd": (0x55555555, 0xAAAAAAAA), "lo16 | hi16 of 32": (0x0000FFFF, 0xFFFF0000),
                           "all | all (ext streams)": (0xFFFFFFFF, 0xFFFFFFFF)}.items():
        mf, mb = masked_stream(wa), masked_stream(wb)
        with torch.cuda.stream(mf):
            t = timeit(lambda: eng2.pipelined(*args40, mf, mb), reps=20) * N / M
        print(f"  C++ loop x40, CU masks {name:24s} {t:7.2f}")
    eng2.close()
