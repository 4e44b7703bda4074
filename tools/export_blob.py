#!/usr/bin/env python3
"""Row n4: reference checkpoint (training/qa_train.py's model_quantized_final.pth, i.e. the
state_dict of the converted int8 model) -> ITAW0001 weight blob for ita_load_weights / ita_udp_server.

    python tools/export_blob.py --checkpoint model_quantized_final.pth --out weights.itaw [--num-layers 1]

Quantized-Linear entries of such a state_dict are (qtensor, bias) tuples; they are loaded with
torch.load(weights_only=True) first and only with --unsafe-load through the unpickler."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--num-layers", type=int, default=1)
    ap.add_argument("--unsafe-load", action="store_true", help="fall back to the full unpickler (only for files you wrote)")
    a = ap.parse_args()
    import torch
    from drone_oa_iree_vit_accelerator_amd import params
    try:
        sd = torch.load(a.checkpoint, map_location="cpu", weights_only=True)
    except Exception as e:
        if not a.unsafe_load:
            raise SystemExit(f"safe loader refused {a.checkpoint}: {e}\n(re-run with --unsafe-load if you trust the file)")
        sd = torch.load(a.checkpoint, map_location="cpu", weights_only=False)
    blob = params.blob_from_state_dict(sd, a.num_layers)
    with open(a.out, "wb") as f:
        f.write(blob)
    print(f"wrote {a.out}: {len(blob)} bytes")


if __name__ == "__main__":
    main()
