import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from drone_oa_iree_vit_accelerator_amd import host, synth
E, th, tw, co, B = 128, 64, 128, 48, 32
c = synth.tail_large_case(0, E, th, tw, co, 1)
eng = host.FusionTailLarge(c["conv_w"], c["conv_b"], device=0)
x = torch.randn((B, th * tw, E), device="cuda")
out = torch.empty((B, co, 2 * th, 2 * tw), device="cuda")
for _ in range(3): eng(x, th, tw, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): eng(x, th, tw, out=out)
e1.record(); torch.cuda.synchronize()
print(sys.argv[1] if len(sys.argv) > 1 else "", round(e0.elapsed_time(e1) / 20, 4), "ms")
