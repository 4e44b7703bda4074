"""Weight/scale blob ("ITAW0001", include/ita_weights.h) packer.

Turns the reference's converted int8 blocks (int8 weights, float biases, scales -- what
training/qa_train.py:81-95 saves and tests/export_and_validation_W_B.py:47-62 extracts)
plus the float32 parameters of the non-quantised layers into the packed blob that
``ita_load_weights`` uploads.  All requantisation constants are derived here, in the
float32 arithmetic the reference's kernels use (pinned against tests/golden):

* ``nnq.Linear``:  bias_q = rne(b * (1f / f32(f32(s_w) * s_x)));  mult = f32(f32(s_w)*f32(s_x)) / f32(s_out)
* ``matmul1``:     mult = f32(f32(s_q)*f32(s_k)) / f32(s_logit)
* ``matmul2``:     mult = f32((1/255 * s_v) / s_ctx)   (python-double product, tests/export_and_validation_W_B.py:140)
* ``Quantize``:    q = rne(x * (1f / f32(s)))
"""
from __future__ import annotations

import struct
from typing import Dict

import numpy as np

f32 = np.float32
MAGIC = b"ITAW0001"
_DT = {np.dtype(np.float32): 0, np.dtype(np.int8): 1, np.dtype(np.int32): 2, np.dtype(np.uint8): 3,
       np.dtype(np.float16): 4}
A_NSCAL, F_NSCAL = 8, 4


def _bias_q(bias, s_w, s_x):
    bsc = f32(float(f32(s_w)) * float(s_x))          # double product of (float w_scale, double in_scale) -> float
    return np.rint(bias.astype(np.float32) * (f32(1.0) / bsc)).astype(np.int32)


def _lin_mult(s_w, s_x, s_out):
    return f32(f32(f32(s_w) * f32(s_x)) / f32(s_out))


def attention_tensors(rec: dict, prefix: str, i: int) -> Dict[str, np.ndarray]:
    """rec: fixture-style record holding '<prefix>q_proj.w_q', '.w_scale', '.bias', '.out_scale', ..."""
    g = lambda k: rec[prefix + k]
    s_x = float(g("quant.scale"))
    s_q, s_k, s_v = (float(g(f"{n}.out_scale")) for n in ("q_proj", "k_proj", "v_proj"))
    s_l, s_c, s_o = float(g("matmul1.scale")), float(g("matmul2.scale")), float(g("out_proj.out_scale"))
    t = {}
    for nm, key, s_in in (("q_proj", "q", s_x), ("k_proj", "k", s_x), ("v_proj", "v", s_x), ("out_proj", "o", s_c)):
        t[f"attn{i}.w{key}"] = np.ascontiguousarray(g(nm + ".w_q").astype(np.int8))
        t[f"attn{i}.b{key}"] = _bias_q(g(nm + ".bias"), g(nm + ".w_scale"), s_in)
    scal = np.zeros(A_NSCAL, np.float32)
    scal[0] = f32(1.0) / f32(s_x)
    scal[1] = _lin_mult(g("q_proj.w_scale"), s_x, s_q)
    scal[2] = _lin_mult(g("k_proj.w_scale"), s_x, s_k)
    scal[3] = _lin_mult(g("v_proj.w_scale"), s_x, s_v)
    scal[4] = f32(f32(f32(s_q) * f32(s_k)) / f32(s_l))
    scal[5] = f32(((1.0 / 255.0) * s_v) / s_c)
    scal[6] = _lin_mult(g("out_proj.w_scale"), s_c, s_o)
    scal[7] = f32(s_o)
    t[f"attn{i}.scal"] = scal
    return t


def ffn_tensors(rec: dict, prefix: str, i: int) -> Dict[str, np.ndarray]:
    g = lambda k: rec[prefix + k]
    s_x, s_1, s_2 = float(g("quant.scale")), float(g("fc1.out_scale")), float(g("fc2.out_scale"))
    t = {f"ffn{i}.w1": np.ascontiguousarray(g("fc1.w_q").astype(np.int8)),
         f"ffn{i}.b1": _bias_q(g("fc1.bias"), g("fc1.w_scale"), s_x),
         f"ffn{i}.w2": np.ascontiguousarray(g("fc2.w_q").astype(np.int8)),
         f"ffn{i}.b2": _bias_q(g("fc2.bias"), g("fc2.w_scale"), s_1)}
    scal = np.zeros(F_NSCAL, np.float32)
    scal[0] = f32(1.0) / f32(s_x)
    scal[1] = _lin_mult(g("fc1.w_scale"), s_x, s_1)
    scal[2] = _lin_mult(g("fc2.w_scale"), s_1, s_2)
    scal[3] = f32(s_2)
    t[f"ffn{i}.scal"] = scal
    return t


def float_tensors(fp: dict, num_layers: int = 1) -> Dict[str, np.ndarray]:
    """float32 layers, keyed by the reference's state_dict names -> blob names."""
    E = fp["tokenizer.conv.weight"].shape[0]
    t = {"tok.conv_w": fp["tokenizer.conv.weight"].reshape(E, 49), "tok.conv_b": fp["tokenizer.conv.bias"],
         "tok.ln_w": fp["tokenizer.norm.weight"], "tok.ln_b": fp["tokenizer.norm.bias"]}
    for i in range(num_layers):
        t[f"norm1_{i}.w"], t[f"norm1_{i}.b"] = fp[f"norms1.{i}.weight"], fp[f"norms1.{i}.bias"]
        t[f"norm2_{i}.w"], t[f"norm2_{i}.b"] = fp[f"norms2.{i}.weight"], fp[f"norms2.{i}.bias"]
    if "down_sample.weight" in fp:
        t["tail.conv_w"], t["tail.conv_b"] = fp["down_sample.weight"], fp["down_sample.bias"]
    if "decoder.weight" in fp:
        t["dec.w"], t["dec.b"] = fp["decoder.weight"], fp["decoder.bias"]
        for l in range(3):
            t[f"lstm.w_ih{l}"], t[f"lstm.w_hh{l}"] = fp[f"lstm.weight_ih_l{l}"], fp[f"lstm.weight_hh_l{l}"]
            t[f"lstm.b_ih{l}"], t[f"lstm.b_hh{l}"] = fp[f"lstm.bias_ih_l{l}"], fp[f"lstm.bias_hh_l{l}"]
        t["fc.w"], t["fc.b"] = fp["nn_fc2.weight"], fp["nn_fc2.bias"]
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in t.items()}


def pack_blob(tensors: Dict[str, np.ndarray], E: int, S: int = 128, P: int = 192, F: int = 256, H: int = 1,
              num_layers: int = 1, has_tail: bool = True) -> bytes:
    names = list(tensors)
    n = len(names)
    hdr_sz, ent_sz = 8 + 4 * 14, 32 + 4 * 6 + 8 * 2
    off = hdr_sz + n * ent_sz
    entries, chunks = [], []
    for nm in names:
        a = np.ascontiguousarray(tensors[nm])
        assert a.dtype in _DT, (nm, a.dtype)
        assert a.ndim <= 4 and len(nm) < 32
        pad = (-off) % 64
        chunks.append(b"\0" * pad)
        off += pad
        shape = list(a.shape) + [0] * (4 - a.ndim)
        entries.append(struct.pack("<32sii4iqq", nm.encode(), _DT[a.dtype], a.ndim, *shape, off, a.nbytes))
        chunks.append(a.tobytes())
        off += a.nbytes
    hdr = struct.pack("<8s14i", MAGIC, n, E, S, P, F, H, num_layers, int(has_tail), 0, 0, 0, 0, 0, 0)
    assert len(hdr) == hdr_sz and all(len(e) == ent_sz for e in entries)
    return hdr + b"".join(entries) + b"".join(chunks)


def blob_from_record(rec: dict, float_params: dict | None, E: int, num_layers: int = 1) -> bytes:
    t: Dict[str, np.ndarray] = {}
    for i in range(num_layers):
        t.update(attention_tensors(rec, f"attn{i}.", i))
        t.update(ffn_tensors(rec, f"ffn{i}.", i))
    has_tail = False
    if float_params is not None:
        t.update(float_tensors(float_params, num_layers))
        has_tail = "tail.conv_w" in t     # without it the decoder reads the flattened tokens (models/ITA/QAT/model.py:80-81)
    return pack_blob(t, E=E, num_layers=num_layers, has_tail=has_tail)


def load_fixture(path: str) -> dict:
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


# ---------------------------------------------------------------------------------------------
# Row n4: from a reference checkpoint to the blob.
#
# training/qa_train.py:81-95 saves ``state_dict()`` of the CONVERTED model
# (model_quantized_final.pth): quantized Linear layers appear as
# ``<name>._packed_params._packed_params = (qint8 weight, float bias)`` plus ``<name>.scale``;
# QuantStubs as ``<block>.quant.scale``; QFunctional matmuls as ``<block>.matmulN.scale``.
# tests/export_and_validation_W_B.py:47-62,233-245 extracts the same items through module hooks;
# here they are read straight from the state_dict, so no reference code is needed.

def record_from_state_dict(sd: dict, num_layers: int = 1) -> dict:
    """converted-model state_dict -> the fixture-style record consumed by blob_from_record"""
    def packed(name):
        w, b = sd[name + "._packed_params._packed_params"]
        if int(w.q_zero_point()) != 0:
            raise ValueError(f"{name}: weights must be symmetric (zero_point 0)")
        return w.int_repr().numpy().astype(np.int8), float(w.q_scale()), b.detach().numpy().astype(np.float32)

    def scalar(name):
        return float(np.asarray(sd[name].detach().cpu().numpy()).reshape(-1)[0])

    rec = {}
    for i in range(num_layers):
        a = f"attention_blocks.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            wq, ws, b = packed(a + nm)
            if int(sd[a + nm + ".zero_point"]) != 0:
                raise ValueError(f"{a + nm}: output zero_point must be 0 (ita_symmetric_qconfig)")
            rec[f"attn{i}.{nm}.w_q"], rec[f"attn{i}.{nm}.w_scale"], rec[f"attn{i}.{nm}.bias"] = wq, ws, b
            rec[f"attn{i}.{nm}.out_scale"] = scalar(a + nm + ".scale")
        rec[f"attn{i}.quant.scale"] = scalar(a + "quant.scale")
        rec[f"attn{i}.matmul1.scale"] = scalar(a + "matmul1.scale")
        rec[f"attn{i}.matmul2.scale"] = scalar(a + "matmul2.scale")
        f = f"ffn_blocks.{i}."
        for nm in ("fc1", "fc2"):
            wq, ws, b = packed(f + nm)
            rec[f"ffn{i}.{nm}.w_q"], rec[f"ffn{i}.{nm}.w_scale"], rec[f"ffn{i}.{nm}.bias"] = wq, ws, b
            rec[f"ffn{i}.{nm}.out_scale"] = scalar(f + nm + ".scale")
        rec[f"ffn{i}.quant.scale"] = scalar(f + "quant.scale")
    return rec


def fold_spectral_norm(sd: dict, name: str):
    """torch.nn.utils.spectral_norm stores weight_orig / weight_u / weight_v; in eval mode the layer
    uses weight_orig / sigma with sigma = u^T W v (models/ITA_single_layer_upsample_shuffle/model.py:81,84).
    The reference's own strict=False load silently drops these keys; fold them explicitly."""
    if name + ".weight" in sd:
        return np.asarray(sd[name + ".weight"].detach().cpu().numpy(), np.float32)
    w = np.asarray(sd[name + ".weight_orig"].detach().cpu().numpy(), np.float64)
    u = np.asarray(sd[name + ".weight_u"].detach().cpu().numpy(), np.float64)
    v = np.asarray(sd[name + ".weight_v"].detach().cpu().numpy(), np.float64)
    sigma = float(u @ (w.reshape(w.shape[0], -1) @ v))
    return (w / sigma).astype(np.float32)


def float_params_from_state_dict(sd: dict, num_layers: int = 1) -> dict:
    """non-quantised layers under the reference's state_dict names (float or converted checkpoint)"""
    g = lambda k: np.asarray(sd[k].detach().cpu().numpy(), np.float32)
    fp = {k: g(k) for k in ("tokenizer.conv.weight", "tokenizer.conv.bias", "tokenizer.norm.weight",
                            "tokenizer.norm.bias", "down_sample.weight", "down_sample.bias", "decoder.bias",
                            "nn_fc2.bias")}
    fp["decoder.weight"] = fold_spectral_norm(sd, "decoder")
    fp["nn_fc2.weight"] = fold_spectral_norm(sd, "nn_fc2")
    for i in range(num_layers):
        for nm in (f"norms1.{i}", f"norms2.{i}"):
            fp[nm + ".weight"], fp[nm + ".bias"] = g(nm + ".weight"), g(nm + ".bias")
    for l in range(3):
        for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            fp[f"lstm.{nm}_l{l}"] = g(f"lstm.{nm}_l{l}")
    return fp


def blob_from_state_dict(sd: dict, num_layers: int = 1) -> bytes:
    fp = float_params_from_state_dict(sd, num_layers)
    E = fp["tokenizer.conv.weight"].shape[0]
    return blob_from_record(record_from_state_dict(sd, num_layers), fp, E=E, num_layers=num_layers)
