"""Deterministic synthetic float parameters for the ITA ViT+LSTM graphs.

The reference ships no trained weights (its .pth/.onnx/.vmfb blobs are absent),
so every test, fixture and benchmark in this repo runs on seeded synthetic
parameters.  They are drawn from numpy's *legacy* ``RandomState`` (whose stream
is frozen across numpy versions), keyed by the reference's ``state_dict`` names
(models/ITA_single_layer_upsample_shuffle/QAT/model.py:36-90), so the fixture
generator can load them into the reference model and the GPU box can regenerate
the large float tensors (decoder 4608x512, LSTM) without shipping them.
"""
from __future__ import annotations

import hashlib
import numpy as np

# (E, S, P, F, H) of the reference graphs
DIMS_VITLSTM = dict(E=64, S=128, P=192, F=256, H=1)      # ITA_single_layer_upsample_shuffle/QAT/model.py:38
DIMS_SINGLE_LAYER = dict(E=128, S=128, P=192, F=256, H=1)  # ITA_single_layer/QAT/model.py:39


def _uniform(rs, shape, bound):
    return rs.uniform(-bound, bound, size=shape).astype(np.float32)


def float_params(seed: int, E: int = 64, P: int = 192, F: int = 256, num_layers: int = 1,
                 gain_qk: float = 3.0, tail: bool = True) -> dict:
    """All float32 parameters of the float twin of the graph, PyTorch-default-like
    init bounds (U(-1/sqrt(fan_in), 1/sqrt(fan_in))), LayerNorm affine perturbed so it
    is not the identity, q/k projections amplified so attention rows are peaky.
    tail=False: the graph family without the fusion tail (models/ITA/QAT/model.py:22-87: the decoder reads the
    flattened tokens, Linear(E * 128 -> 512)); the random stream of the tail=True form is untouched."""
    rs = np.random.RandomState(1000 + seed)
    p = {}
    p["tokenizer.conv.weight"] = _uniform(rs, (E, 1, 7, 7), 1 / 7.0)
    p["tokenizer.conv.bias"] = _uniform(rs, (E,), 1 / 7.0)
    p["tokenizer.norm.weight"] = (1.0 + 0.1 * rs.standard_normal(E)).astype(np.float32)
    p["tokenizer.norm.bias"] = (0.1 * rs.standard_normal(E)).astype(np.float32)
    for i in range(num_layers):
        a = f"attention_blocks.{i}."
        for nm, fo, fi, g in (("q_proj", P, E, gain_qk), ("k_proj", P, E, gain_qk),
                              ("v_proj", P, E, 1.0), ("out_proj", E, P, 1.0)):
            b = g / np.sqrt(fi)
            p[a + nm + ".weight"] = _uniform(rs, (fo, fi), b)
            p[a + nm + ".bias"] = _uniform(rs, (fo,), b)
        f = f"ffn_blocks.{i}."
        p[f + "fc1.weight"] = _uniform(rs, (F, E), 1 / np.sqrt(E))
        p[f + "fc1.bias"] = _uniform(rs, (F,), 1 / np.sqrt(E))
        p[f + "fc2.weight"] = _uniform(rs, (E, F), 1 / np.sqrt(F))
        p[f + "fc2.bias"] = _uniform(rs, (E,), 1 / np.sqrt(F))
        for nm in (f"norms1.{i}", f"norms2.{i}"):
            p[nm + ".weight"] = (1.0 + 0.1 * rs.standard_normal(E)).astype(np.float32)
            p[nm + ".bias"] = (0.1 * rs.standard_normal(E)).astype(np.float32)
    cin = E // 4 + E
    if tail:
        p["down_sample.weight"] = _uniform(rs, (9, cin, 3, 3), 1 / np.sqrt(cin * 9))
        p["down_sample.bias"] = _uniform(rs, (9,), 1 / np.sqrt(cin * 9))
    dec_in = 4608 if tail else E * 128
    p["decoder.weight"] = _uniform(rs, (512, dec_in), 1 / np.sqrt(dec_in))
    p["decoder.bias"] = _uniform(rs, (512,), 1 / np.sqrt(dec_in))
    k = 1 / np.sqrt(128)
    for l, fi in enumerate((517, 128, 128)):
        p[f"lstm.weight_ih_l{l}"] = _uniform(rs, (512, fi), k)
        p[f"lstm.weight_hh_l{l}"] = _uniform(rs, (512, 128), k)
        p[f"lstm.bias_ih_l{l}"] = _uniform(rs, (512,), k)
        p[f"lstm.bias_hh_l{l}"] = _uniform(rs, (512,), k)
    p["nn_fc2.weight"] = _uniform(rs, (3, 128), k)
    p["nn_fc2.bias"] = _uniform(rs, (3,), k)
    return p


def frames(seed: int, B: int, gain: float = 1.0) -> dict:
    """Synthetic inputs in the wire format of the reference host
    (samples/inference_udp_FPGA_custom_dispatch/main.cpp:37,168-169): u8 depth frame
    60x90 (-> f32 / 255 on the host), desired velocity, unit quaternion."""
    rs = np.random.RandomState(1234 + seed)
    # smooth-ish depth image: low-res random field upsampled + noise, so conv outputs vary
    base = rs.uniform(0, 255, size=(B, 6, 9))
    img = np.kron(base, np.ones((10, 10))) * 0.7 + rs.uniform(0, 255, size=(B, 60, 90)) * 0.3
    img = np.clip(img * gain, 0, 255).astype(np.uint8)
    desvel = rs.uniform(2.0, 8.0, size=(B, 1)).astype(np.float32)
    q = rs.standard_normal((B, 4))
    quat = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    return dict(img_u8=img, desvel=desvel, quat=quat)


def tail_large_case(seed: int, E: int, tok_h: int, tok_w: int, out_ch: int, B: int) -> dict:
    """Seeded inputs of the large-grid fusion tail (BASELINE config 5): tokens, conv weight (out_ch, 5E/4, 3, 3)
    with the fan-in scaling of nn.Conv2d's default init, bias.  numpy legacy RandomState: stable across versions."""
    rs = np.random.RandomState(4321 + seed)
    cin = E // 4 + E
    bound = 1.0 / np.sqrt(cin * 9)
    return dict(x=rs.standard_normal((B, tok_h * tok_w, E)).astype(np.float32),
                conv_w=rs.uniform(-bound, bound, size=(out_ch, cin, 3, 3)).astype(np.float32),
                conv_b=rs.uniform(-bound, bound, size=(out_ch,)).astype(np.float32))


def digest(params: dict) -> str:
    """sha256 over the float params in key order: committed with each fixture so a
    drift of the generator is detected instead of silently changing the expected outputs."""
    h = hashlib.sha256()
    for k in sorted(params):
        h.update(k.encode())
        h.update(np.ascontiguousarray(params[k]).tobytes())
    return h.hexdigest()
