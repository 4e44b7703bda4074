"""The float32 twin of the ITAViTLSTM graph -- what the reference exports and compiles for the CPU (.vmfb, SURVEY.md
section 8 row a11 / BASELINE config 1): true softmax attention, float Linear layers, no quantisation anywhere.

This is a restatement in torch functional ops of
    models/ITA/layers.py:8-27   OverlapPatchMerging  (conv7x7 s2 p3 -> bilinear 8x16 -> LayerNorm)
    models/ITA/layers.py:47-88  ITASelfAttention     (q/k/v/out Linear, softmax(Q K^T) V -- no 1/sqrt(d), like the int8 block)
    models/ITA/layers.py:29-45  ITAFeedForward       (fc1 -> ReLU -> fc2)
    models/ITA_single_layer_upsample_shuffle/model.py:88-140   ITALSTMNetVIT.forward (residual + LayerNorm glue,
                                PixelShuffle || Upsample -> conv3x3, decoder, cat, 3-layer LSTM, fc)
taking the parameters as a plain dict under the reference's state_dict names (spectral norm already folded:
params.fold_spectral_norm).  It is NOT on the product path: bench.py times it on the host cores as the CPU baseline of
record (kind "torch-f32-eager"), and tests/test_float_twin.py pins it against a fixture produced by the reference's own
module (tools/gen_golden.py: gen_float_twin)."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np


class FloatTwin:
    def __init__(self, fp: Dict[str, np.ndarray], num_layers: int = 1, device: str = "cpu"):
        import torch
        self.t = torch
        self.L = num_layers
        self.p = {k: torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(device) for k, v in fp.items()}
        self.E = int(self.p["tokenizer.conv.weight"].shape[0])
        self.lstm = torch.nn.LSTM(input_size=517, hidden_size=128, num_layers=3).to(device).eval()
        with torch.no_grad():
            for l in range(3):
                for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                    getattr(self.lstm, f"{nm}_l{l}").copy_(self.p[f"lstm.{nm}_l{l}"])

    def _attention(self, x, i):
        F, p = self.t.nn.functional, self.p
        a = f"attention_blocks.{i}."
        q = F.linear(x, p[a + "q_proj.weight"], p[a + "q_proj.bias"])
        k = F.linear(x, p[a + "k_proj.weight"], p[a + "k_proj.bias"])
        v = F.linear(x, p[a + "v_proj.weight"], p[a + "v_proj.bias"])
        w = self.t.softmax(self.t.matmul(q, k.transpose(-2, -1)), dim=-1)      # one head: (B,128,128)
        return F.linear(self.t.matmul(w, v), p[a + "out_proj.weight"], p[a + "out_proj.bias"])

    def _ffn(self, x, i):
        F, p = self.t.nn.functional, self.p
        f = f"ffn_blocks.{i}."
        return F.linear(F.relu(F.linear(x, p[f + "fc1.weight"], p[f + "fc1.bias"])), p[f + "fc2.weight"], p[f + "fc2.bias"])

    def forward(self, img, desvel, quat=None, hidden: Optional[Tuple] = None, taps: bool = False):
        """img (B,60,90) f32 in [0,1] or u8 wire frames; desvel (B,1) or (B,); quat (B,4) -> vel (B,3), (h, c) (3,B,128)"""
        torch, p = self.t, self.p
        F = torch.nn.functional
        with torch.no_grad():
            img = torch.as_tensor(img)
            img = img.float() / 255.0 if img.dtype == torch.uint8 else img.float()
            if img.shape[-2:] != (60, 90):
                img = F.interpolate(img.reshape(-1, 1, *img.shape[-2:]), size=(60, 90), mode="bilinear", align_corners=False)
            img = img.reshape(-1, 1, 60, 90)
            B, E = img.shape[0], self.E
            desvel = torch.as_tensor(desvel).float().reshape(B, 1)
            quat = torch.tensor([[1.0, 0, 0, 0]]).repeat(B, 1) if quat is None else torch.as_tensor(quat).float().reshape(B, 4)
            x = F.conv2d(img, p["tokenizer.conv.weight"], p["tokenizer.conv.bias"], stride=2, padding=3)
            x = F.interpolate(x, size=(8, 16), mode="bilinear", align_corners=False).flatten(2).transpose(1, 2)
            x = F.layer_norm(x, (E,), p["tokenizer.norm.weight"], p["tokenizer.norm.bias"])
            tp = {"tokens": x}
            for i in range(self.L):
                x = F.layer_norm(x + self._attention(x, i), (E,), p[f"norms1.{i}.weight"], p[f"norms1.{i}.bias"])
                tp["x1"] = x
                x = F.layer_norm(x + self._ffn(x, i), (E,), p[f"norms2.{i}.weight"], p[f"norms2.{i}.bias"])
            tp["x2"] = x
            x2d = x.transpose(1, 2).reshape(B, E, 8, 16)
            fused = torch.cat([F.pixel_shuffle(x2d, 2), F.interpolate(x2d, size=(16, 32), mode="bilinear", align_corners=True)], 1)
            feat = F.conv2d(fused, p["down_sample.weight"], p["down_sample.bias"], padding=1).flatten(1)
            dec = F.linear(feat, p["decoder.weight"], p["decoder.bias"])
            tp["dec"] = dec
            cat = torch.cat([dec, desvel / 10.0, quat], 1).unsqueeze(0)
            out, (h, c) = self.lstm(cat, None if hidden is None else (torch.as_tensor(hidden[0]).float(), torch.as_tensor(hidden[1]).float()))
            vel = F.linear(out.squeeze(0), p["nn_fc2.weight"], p["nn_fc2.bias"])
        return (vel, (h, c), tp) if taps else (vel, (h, c))
