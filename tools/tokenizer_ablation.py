"""Time the stand-alone ita_tokenizer_kernel (u8 frames, B=1024) for the diagnostic masks of ITA_TOK_DBG.
Back-to-back launches through Python are host-bound below ~10 us per call: read differences, not absolutes.

Usage: python tools/tokenizer_ablation.py [B]     (spawns one child per mask: the mask is read once per process)
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(B):
    import torch
    from drone_oa_iree_vit_accelerator_amd import host, params, synth
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = params.load_fixture(os.path.join(repo, "tests", "golden", "vitlstm_E64_seed0_B2.npz"))
    blob = params.blob_from_record(fx, synth.float_params(0, E=64), E=64)
    eng = host.Engine(blob, device=0)
    img = synth.frames(0, B)["img_u8"]
    x = torch.from_numpy(img).cuda()
    for _ in range(20):
        eng.tokenizer(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for _ in range(n):
        eng.tokenizer(x)
    e1.record()
    torch.cuda.synchronize()
    print("%.2f us" % (e0.elapsed_time(e1) * 1000 / n))


if __name__ == "__main__":
    if os.environ.get("ITA_ABL_CHILD"):
        child(int(sys.argv[1]))
    else:
        B = sys.argv[1] if len(sys.argv) > 1 else "1024"
        for mask in (0, 1, 2, 4, 8, 15):   # ItaTokArgs::dbg: 1 skip image fill, 2 skip blend, 4 skip MFMA, 8 skip LN/store
            env = dict(os.environ, ITA_TOK_DBG=str(mask), ITA_ABL_CHILD="1")
            r = subprocess.run([sys.executable, __file__, B], env=env, capture_output=True, text=True)
            print("mask %2d: %s" % (mask, (r.stdout.strip() or r.stderr.strip()[-300:])))
