"""ctypes binding of the TEST ORACLE (oracle/libita_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libita_oracle.so")
    src = os.path.join(_HERE, "ita_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libita_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libita_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.ita_oracle_expf.restype = C.c_float
        _LIB.ita_oracle_expf.argtypes = [C.c_float]
        _LIB.ita_oracle_softmax_inverse.restype = C.c_int32
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def tokenizer(img, cw, cb, lnw, lnb):
    """f32 frames: the float blend; uint8 wire frames: the exact integer blend (ita_oracle_tokenizer_u8)"""
    img = np.asarray(img)
    if img.dtype == np.uint8:
        img = np.ascontiguousarray(img).reshape(-1, 60, 90)
        B, E = img.shape[0], cw.shape[0]
        out = np.empty((B, 128, E), np.float32)
        rc = lib().ita_oracle_tokenizer_u8(_p(img), B, E, _p(_c(cw, np.float32)), _p(_c(cb, np.float32)),
                                           _p(_c(lnw, np.float32)), _p(_c(lnb, np.float32)), _p(out))
        if rc:
            raise RuntimeError(f"ita_oracle_tokenizer_u8 rc={rc}")
        return out
    img = _c(img, np.float32).reshape(-1, 60, 90)
    B, E = img.shape[0], cw.shape[0]
    out = np.empty((B, 128, E), np.float32)
    lib().ita_oracle_tokenizer(_p(img), B, E, _p(_c(cw, np.float32)), _p(_c(cb, np.float32)),
                               _p(_c(lnw, np.float32)), _p(_c(lnb, np.float32)), _p(out))
    return out


def quantize(x, inv_scale):
    x = _c(x, np.float32)
    q = np.empty(x.shape, np.int8)
    lib().ita_oracle_quantize(_p(x), C.c_size_t(x.size), C.c_float(inv_scale), _p(q))
    return q


def linear_q(x, w, bq, mult, relu=False):
    x, w = _c(x, np.int8), _c(w, np.int8)
    rows, K, N = x.size // x.shape[-1], x.shape[-1], w.shape[0]
    y = np.empty(x.shape[:-1] + (N,), np.int8)
    lib().ita_oracle_linear_q(_p(x), rows, K, N, _p(w), _p(_c(bq, np.int32)), C.c_float(mult), int(relu), _p(y))
    return y


def softmax(logits):
    x = _c(logits, np.int8)
    y = np.empty(x.shape, np.uint8)
    lib().ita_oracle_softmax(_p(x), x.size // x.shape[-1], x.shape[-1], _p(y))
    return y


def mha(x, t, i=0, taps=False):
    """t: blob-name keyed tensors (params.attention_tensors).  Returns out_f [, dict of taps]."""
    x = _c(x, np.float32)
    B, S, E = x.shape
    P = t[f"attn{i}.wq"].shape[0]
    out = np.empty_like(x)
    tp = {}
    if taps:
        tp = dict(x_q=np.empty((B, S, E), np.int8), Q=np.empty((B, S, P), np.int8), K=np.empty((B, S, P), np.int8),
                  V=np.empty((B, S, P), np.int8), logits=np.empty((B, S, S), np.int8),
                  probs=np.empty((B, S, S), np.uint8), ctx=np.empty((B, S, P), np.int8),
                  out_q=np.empty((B, S, E), np.int8))
    order = ("x_q", "Q", "K", "V", "logits", "probs", "ctx", "out_q")
    lib().ita_oracle_mha(_p(x), B, S, E, P, _p(t[f"attn{i}.wq"]), _p(t[f"attn{i}.bq"]), _p(t[f"attn{i}.wk"]),
                         _p(t[f"attn{i}.bk"]), _p(t[f"attn{i}.wv"]), _p(t[f"attn{i}.bv"]), _p(t[f"attn{i}.wo"]),
                         _p(t[f"attn{i}.bo"]), _p(t[f"attn{i}.scal"]), _p(out), *[_p(tp.get(k)) for k in order])
    return (out, tp) if taps else out


def mha_q8(x_q, t, i=0):
    """the attention block on int8 codes: x_q (B,S,E) int8 -> out_proj codes (B,S,E) int8 (ita_oracle_mha_q8)"""
    x_q = _c(x_q, np.int8)
    B, S, E = x_q.shape
    P = t[f"attn{i}.wq"].shape[0]
    out = np.empty_like(x_q)
    lib().ita_oracle_mha_q8(_p(x_q), B, S, E, P, _p(t[f"attn{i}.wq"]), _p(t[f"attn{i}.bq"]), _p(t[f"attn{i}.wk"]),
                            _p(t[f"attn{i}.bk"]), _p(t[f"attn{i}.wv"]), _p(t[f"attn{i}.bv"]), _p(t[f"attn{i}.wo"]),
                            _p(t[f"attn{i}.bo"]), _p(t[f"attn{i}.scal"]), _p(out))
    return out


def mha_q8_rows(x_q, t, rows, i=0):
    """rows `rows` of mha_q8's result for ONE frame x_q (S,E) int8 of any length S (ita_oracle_mha_q8_rows): (n,E) int8"""
    x_q = _c(x_q, np.int8)
    S, E = x_q.shape
    P = t[f"attn{i}.wq"].shape[0]
    rows = _c(rows, np.int32)
    out = np.empty((len(rows), E), np.int8)
    rc = lib().ita_oracle_mha_q8_rows(_p(x_q), S, E, P, _p(t[f"attn{i}.wq"]), _p(t[f"attn{i}.bq"]), _p(t[f"attn{i}.wk"]),
                                      _p(t[f"attn{i}.bk"]), _p(t[f"attn{i}.wv"]), _p(t[f"attn{i}.bv"]), _p(t[f"attn{i}.wo"]),
                                      _p(t[f"attn{i}.bo"]), _p(t[f"attn{i}.scal"]), _p(rows), len(rows), _p(out))
    if rc:
        raise RuntimeError(f"ita_oracle_mha_q8_rows rc={rc}")
    return out


def ffn(x, t, i=0, taps=False):
    x = _c(x, np.float32)
    B, S, E = x.shape
    F = t[f"ffn{i}.w1"].shape[0]
    out = np.empty_like(x)
    tp = {}
    if taps:
        tp = dict(x_q=np.empty((B, S, E), np.int8), h=np.empty((B, S, F), np.int8), out_q=np.empty((B, S, E), np.int8))
    lib().ita_oracle_ffn(_p(x), B, S, E, F, _p(t[f"ffn{i}.w1"]), _p(t[f"ffn{i}.b1"]), _p(t[f"ffn{i}.w2"]),
                         _p(t[f"ffn{i}.b2"]), _p(t[f"ffn{i}.scal"]), _p(out),
                         *[_p(tp.get(k)) for k in ("x_q", "h", "out_q")])
    return (out, tp) if taps else out


def add_ln(x, y, w, b):
    x, y = _c(x, np.float32), _c(y, np.float32)
    E = x.shape[-1]
    out = np.empty_like(x)
    lib().ita_oracle_add_ln(_p(x), _p(y), x.size // E, E, _p(_c(w, np.float32)), _p(_c(b, np.float32)), _p(out))
    return out


def tail(x2, cw, cb, want_fused=False):
    x2 = _c(x2, np.float32)
    B, _, E = x2.shape
    feat = np.empty((B, 9 * 16 * 32), np.float32)
    fused = np.empty((B, E // 4 + E, 16, 32), np.float32) if want_fused else None
    lib().ita_oracle_tail(_p(x2), B, E, _p(_c(cw, np.float32)), _p(_c(cb, np.float32)), _p(feat), _p(fused))
    return (feat, fused) if want_fused else feat


def tail_general(x2, tok_h, tok_w, cw, cb, want_fused=False):
    """fusion tail on a tok_h x tok_w token grid, cw (CO, 5E/4, 3, 3) -> (B, CO, 2 tok_h, 2 tok_w)  (BASELINE config 5)"""
    x2 = _c(x2, np.float32)
    B, T, E = x2.shape
    assert T == tok_h * tok_w
    cw = _c(cw, np.float32)
    CO = cw.shape[0]
    out = np.empty((B, CO, 2 * tok_h, 2 * tok_w), np.float32)
    fused = np.empty((B, E // 4 + E, 2 * tok_h, 2 * tok_w), np.float32) if want_fused else None
    lib().ita_oracle_tail_general(_p(x2), B, E, tok_h, tok_w, CO, _p(cw), _p(_c(cb, np.float32)), _p(out), _p(fused))
    return (out, fused) if want_fused else out


def tail_general_at(x2, tok_h, tok_w, cw, cb, pts):
    """the same map sampled at pts (n, 4) int32 rows (b, o, y, x)"""
    x2, cw = _c(x2, np.float32), _c(cw, np.float32)
    pts = _c(pts, np.int32)
    vals = np.empty(len(pts), np.float32)
    lib().ita_oracle_tail_general_at(_p(x2), x2.shape[-1], tok_h, tok_w, cw.shape[0], _p(cw), _p(_c(cb, np.float32)),
                                     _p(pts), len(pts), _p(vals))
    return vals


def linear_f32(x, w, b):
    x, w = _c(x, np.float32), _c(w, np.float32)
    rows, K, N = x.size // x.shape[-1], x.shape[-1], w.shape[0]
    y = np.empty(x.shape[:-1] + (N,), np.float32)
    lib().ita_oracle_linear_f32(_p(x), rows, K, N, _p(w), _p(_c(b, np.float32)), _p(y))
    return y


def lstm_cell(x, w_ih, w_hh, b_ih, b_hh, h_in, c_in):
    """one nn.LSTM layer, seq_len 1 (QAT/model.py:128-129): x (B,In), h_in/c_in (B,128) -> (h, c)"""
    x, h_in, c_in = _c(x, np.float32), _c(h_in, np.float32), _c(c_in, np.float32)
    B, In = x.shape
    h, c = np.empty((B, 128), np.float32), np.empty((B, 128), np.float32)
    lib().ita_oracle_lstm_cell(_p(x), B, In, _p(_c(w_ih, np.float32)), _p(_c(w_hh, np.float32)), _p(_c(b_ih, np.float32)),
                               _p(_c(b_hh, np.float32)), _p(h_in), _p(c_in), _p(h), _p(c))
    return h, c


def head_from_dec(dec, desvel, quat, fp, h_in=None, c_in=None):
    """the recurrent head after the decoder (QAT/model.py:126-130): cat([dec, desvel/10, quat]) -> 3 LSTM layers ->
    fc.  fp: float parameters under the reference's state_dict names.  Returns (vel, h (3,B,128), c)."""
    dec = _c(dec, np.float32)
    B = dec.shape[0]
    cat = np.concatenate([dec, (_c(desvel, np.float32).reshape(B, 1) / np.float32(10.0)), _c(quat, np.float32).reshape(B, 4)], 1)
    h_in = np.zeros((3, B, 128), np.float32) if h_in is None else _c(h_in, np.float32)
    c_in = np.zeros((3, B, 128), np.float32) if c_in is None else _c(c_in, np.float32)
    h, c, x = np.empty_like(h_in), np.empty_like(c_in), cat
    for l in range(3):
        h[l], c[l] = lstm_cell(x, fp[f"lstm.weight_ih_l{l}"], fp[f"lstm.weight_hh_l{l}"], fp[f"lstm.bias_ih_l{l}"],
                               fp[f"lstm.bias_hh_l{l}"], h_in[l], c_in[l])
        x = h[l]
    return linear_f32(x, fp["nn_fc2.weight"], fp["nn_fc2.bias"]), h, c


def forward_from_tokens(tokens, block_tensors, fp, num_layers, desvel, quat, h_in=None, c_in=None):
    """the graph BEHIND the tokenizer, composed from the block entry points above (QAT/model.py:100-130): encoder layers,
    fusion tail (when fp holds `down_sample.*`) or the flattened tokens, decoder, LSTM head.  block_tensors: blob-name keyed
    int8 tensors of every layer (params.attention_tensors / ffn_tensors), fp: float parameters.  Lets a test start from
    the REFERENCE's token tensor, so that what follows is free of the tokenizer's float noise.
    Returns (vel, h, c, {"x1": last layer's LayerNorm1 output, "x2", "dec", "x_q0": layer 0's input codes})."""
    x = _c(tokens, np.float32)
    B = x.shape[0]
    x1, xq0 = None, None
    for l in range(num_layers):
        a, tp = mha(x, block_tensors, l, taps=True)
        if l == 0:
            xq0 = tp["x_q"]
        x1 = add_ln(x, a, fp[f"norms1.{l}.weight"], fp[f"norms1.{l}.bias"])
        x = add_ln(x1, ffn(x1, block_tensors, l), fp[f"norms2.{l}.weight"], fp[f"norms2.{l}.bias"])
    if "down_sample.weight" in fp:
        dec = linear_f32(tail(x, fp["down_sample.weight"], fp["down_sample.bias"]), fp["decoder.weight"], fp["decoder.bias"])
    else:
        dec = linear_f32(x.reshape(B, -1), fp["decoder.weight"], fp["decoder.bias"])
    vel, h, c = head_from_dec(dec, desvel, quat, fp, h_in, c_in)
    return vel, h, c, {"x1": x1, "x2": x, "dec": dec, "x_q0": xq0}


def forward(blob: bytes, image, desvel, quat, h_in=None, c_in=None, taps=False):
    """module.main_graph semantics with leading batch (QAT/model.py:93-132)."""
    image = np.ascontiguousarray(image)
    is_u8 = image.dtype == np.uint8
    if not is_u8:
        image = image.astype(np.float32)
    B = image.reshape(-1, 60, 90).shape[0]
    E = np.frombuffer(blob[12:16], np.int32)[0]
    desvel, quat = _c(desvel, np.float32).reshape(B), _c(quat, np.float32).reshape(B, 4)
    h_in = np.zeros((3, B, 128), np.float32) if h_in is None else _c(h_in, np.float32)
    c_in = np.zeros((3, B, 128), np.float32) if c_in is None else _c(c_in, np.float32)
    vel = np.empty((B, 3), np.float32)
    h_out, c_out = np.empty((3, B, 128), np.float32), np.empty((3, B, 128), np.float32)
    tp = {}
    if taps:
        tp = dict(tokens=np.empty((B, 128, E), np.float32), x1=np.empty((B, 128, E), np.float32),
                  x2=np.empty((B, 128, E), np.float32), feat=np.empty((B, 4608), np.float32),
                  dec=np.empty((B, 512), np.float32))
    buf = C.create_string_buffer(blob, len(blob))
    rc = lib().ita_oracle_forward(buf, C.c_size_t(len(blob)), _p(image), int(is_u8), _p(desvel), _p(quat), _p(h_in),
                                  _p(c_in), B, _p(vel), _p(h_out), _p(c_out),
                                  *[_p(tp.get(k)) for k in ("tokens", "x1", "x2", "feat", "dec")])
    if rc != 0:
        raise RuntimeError(f"ita_oracle_forward rc={rc}")
    return (vel, h_out, c_out, tp) if taps else (vel, h_out, c_out)


def unpack_packet(packet: bytes, ref_bug: bool = False):
    out = np.zeros(6, np.float32)
    buf = C.create_string_buffer(packet, len(packet))
    rc = lib().ita_oracle_unpack_packet(buf, C.c_size_t(len(packet)), int(ref_bug), _p(out))
    return None if rc else out


def final_velocity(raw, desired_vel, pos_x):
    raw = _c(raw, np.float32)
    out = np.zeros(3, np.float32)
    lib().ita_oracle_final_velocity(_p(raw), C.c_float(desired_vel), C.c_float(pos_x), _p(out))
    return out
