#!/usr/bin/env python3
"""Phase timeline of ita_tail_up_kernel (BASELINE config 5) from in-kernel s_memrealtime stamps (100 MHz).
Needs the diagnostic build:  tools/build_variant.sh upst -DITA_UP_STAMP  and that library copied over libita_mi355x.so
(GPU box only).  Prints, per phase, the mean / p10 / p90 over the workgroups of one launch for every wave (0-3 hold two token tiles, 4-7 one), and the launch's span.  usage: python tools/tail_up_stamps.py [frames]"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from drone_oa_iree_vit_accelerator_amd import host, synth

E, th, tw, co = 128, 64, 128, 48
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
c = synth.tail_large_case(0, E, th, tw, co, 1)
eng = host.FusionTailLarge(c["conv_w"], c["conv_b"], device=0)
x = torch.randn((B, th * tw, E), device="cuda")
out = torch.empty((B, co, 2 * th, 2 * tw), device="cuda")
for _ in range(3):
    eng(x, th, tw, out=out)
torch.cuda.synchronize()
L = host.lib()
nwg = (2 * tw // 32) * (2 * th // 16) * B
buf = np.zeros((nwg, 8, 12), np.uint64)
L.ita_debug_tail_up_stamps.argtypes = [C.c_void_p, C.c_int]
rc = L.ita_debug_tail_up_stamps(buf.ctypes.data, buf.size)
assert rc == 0, rc
t = buf.astype(np.float64) * 0.01            # microseconds
order = [0, 1, 2, 3, 4, 8, 9, 7, 5, 6]
names = ["tokens loaded + split (prologue)", "phase 1: nine taps (GEMM + blend)", "shuffle halo + weights staged", "phase 2 MFMA",
         "barrier (phase 2 done by all)", "bias loaded, U written", "barrier (U complete)", "U read, added, stores issued", "stores drained"]
print(f"{nwg} workgroups, launch span {t[:, :, 6].max() - t[:, :, 0].min():.1f} us, "
      f"workgroup lifetime mean {np.mean(t[:, 0, 6] - t[:, 0, 0]):.2f} us")
for w in range(8):
    print(f"wave {w}:")
    for i, n in enumerate(names):
        d = t[:, w, order[i + 1]] - t[:, w, order[i]]
        print(f"  {n:40s} mean {d.mean():7.2f}  p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f} us")
# how many workgroups does a CU run back to back, and with what gap?  (sort by start time; a CU's next workgroup starts when
# the previous one ends: count starts inside 0.5 us after some end)
# persistent kernel: workgroup w runs tiles w, w + G, ...: gap between a tile's end and the next tile's start on the same workgroup
G = 256
if nwg > G:
    gaps = t[G:, 0, 0] - t[:-G, 0, 6]
    print(f"tile-to-tile gap on a workgroup (end of stores -> next tile's first stamp): mean {gaps.mean():.2f} us; tile period mean {np.mean(t[G:, 0, 0] - t[:-G, 0, 0]):.2f} us")
eng.close()
