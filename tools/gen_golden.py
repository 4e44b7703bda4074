#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE (PyTorch int8 path) in the build container.

Runs only where /root/reference exists (never on the GPU box).  It imports the
reference's own modules
    models/ITA_single_layer_upsample_shuffle/QAT/model.py   (ITALSTMNetVIT_QAT, E=64)
    models/ITA/QAT/layers.py                                (ITASelfAttention_QAT, ITAFeedForward_QAT)
    models/ITA/QAT/ITA_softmax.py                           (IntegerApproximatedSoftmax)
loads this repo's seeded synthetic float parameters into them, does the reference's
QAT flow (qconfig on attention_blocks/ffn_blocks -> prepare_qat -> calibration
forwards -> convert; training/qa_train.py:60-95, tests/export_and_validation_W_B.py:359-382),
patches matmul2 exactly like the reference's validation script does
(tests/export_and_validation_W_B.py:120-151,409-422 -- restated here, not copied:
int32 accumulate of (a-zp_a)(b-zp_b), fp32 multiply by (s_a*s_b)/s_out, round, clamp),
and dumps inputs + per-stage tensors as compressed .npz under tests/golden/.

Only DATA is written: int8 weights / scales produced by the reference's convert(),
inputs, per-stage outputs.  No reference source is stored.

Usage:  python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("ITA_REFERENCE_ROOT", "/root/reference")


def _load_synth():
    spec = importlib.util.spec_from_file_location(
        "ita_synth", os.path.join(REPO, "drone-oa-iree-vit-accelerator_amd", "synth.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


synth = _load_synth()
sys.path.insert(0, REF)
from models.ITA.QAT.layers import (ITAFeedForward_QAT, ITASelfAttention_QAT,  # noqa: E402
                                   ita_symmetric_qconfig)
from models.ITA_single_layer_upsample_shuffle.QAT.model import ITALSTMNetVIT_QAT  # noqa: E402

torch.backends.quantized.engine = "qnnpack"   # tests/export_and_validation_W_B.py:360
torch.set_num_threads(1)


def patched_matmul2(scale, zero_point):
    """mixed quint8 x qint8 matmul the way the reference's validation harness defines it."""
    def f(q_a, q_b):
        a = q_a.int_repr().to(torch.int32) - q_a.q_zero_point()
        b = q_b.int_repr().to(torch.int32) - q_b.q_zero_point()
        acc = torch.matmul(a, b)
        mult = (q_a.q_scale() * q_b.q_scale()) / scale
        out = (acc.float() * mult + zero_point).round().clamp(-128, 127).to(torch.int8)
        return torch._make_per_tensor_quantized_tensor(out, scale=scale, zero_point=zero_point)
    return f


class Tap:
    """forward hooks that record int_repr / float outputs (and first input) per module."""

    def __init__(self):
        self.t = {}

    @staticmethod
    def _np(x):
        if isinstance(x, (tuple, list)):
            x = x[0]
        if x.is_quantized:
            return x.int_repr().detach().numpy().copy()
        return x.detach().numpy().copy()

    def add(self, module, name, with_input=False):
        def hook(mod, inp, out):
            self.t[name] = self._np(out)
            if with_input:
                self.t[name + ".in"] = self._np(inp[0])
        module.register_forward_hook(hook)


def block_quant_record(prefix, attn=None, ffn=None):
    """int8 weights, float biases and every scale the converted reference modules hold."""
    r = {}
    if attn is not None:
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin = getattr(attn, nm)
            w = lin.weight()
            assert w.q_zero_point() == 0 and lin.zero_point == 0
            r[f"{prefix}{nm}.w_q"] = w.int_repr().numpy().copy()
            r[f"{prefix}{nm}.w_scale"] = np.float64(w.q_scale())
            r[f"{prefix}{nm}.bias"] = lin.bias().detach().numpy().copy()
            r[f"{prefix}{nm}.out_scale"] = np.float64(lin.scale)
        r[f"{prefix}quant.scale"] = np.float64(float(attn.quant.scale))
        assert int(attn.quant.zero_point) == 0
        r[f"{prefix}matmul1.scale"] = np.float64(attn.matmul1.scale)
        r[f"{prefix}matmul2.scale"] = np.float64(attn.matmul2.scale)
        assert attn.matmul1.zero_point == 0 and attn.matmul2.zero_point == 0
    if ffn is not None:
        for nm in ("fc1", "fc2"):
            lin = getattr(ffn, nm)
            w = lin.weight()
            assert w.q_zero_point() == 0 and lin.zero_point == 0
            r[f"{prefix}{nm}.w_q"] = w.int_repr().numpy().copy()
            r[f"{prefix}{nm}.w_scale"] = np.float64(w.q_scale())
            r[f"{prefix}{nm}.bias"] = lin.bias().detach().numpy().copy()
            r[f"{prefix}{nm}.out_scale"] = np.float64(lin.scale)
        r[f"{prefix}quant.scale"] = np.float64(float(ffn.quant.scale))
        assert int(ffn.quant.zero_point) == 0
    return r


def tap_attention(tap, attn, prefix):
    tap.add(attn.quant, prefix + "x_q", with_input=True)       # .in = float block input
    tap.add(attn.q_proj, prefix + "Q")
    tap.add(attn.k_proj, prefix + "K")
    tap.add(attn.v_proj, prefix + "V")
    tap.add(attn.custom_softmax, prefix + "probs", with_input=True)   # .in = int8 logits
    tap.add(attn.out_proj, prefix + "out_q", with_input=True)         # .in = int8 context
    tap.add(attn.dequant, prefix + "out_f")


def tap_ffn(tap, ffn, prefix):
    tap.add(ffn.quant, prefix + "x_q", with_input=True)
    tap.add(ffn.fc1, prefix + "h1")
    tap.add(ffn.activation, prefix + "h1_relu")
    tap.add(ffn.fc2, prefix + "out_q")
    tap.add(ffn.dequant, prefix + "out_f")


def to_X(fr, hidden=None):
    img = torch.from_numpy(fr["img_u8"].astype(np.float32) / np.float32(255.0)).unsqueeze(1)
    X = [img, torch.from_numpy(fr["desvel"]), torch.from_numpy(fr["quat"])]
    X.append(hidden)
    return X


def gen_vitlstm(seed, B, out_dir):
    """Full ITAViTLSTM (E=64) int8 model: per-stage tensors for step 0, and a second
    time step that carries (h, c) like the reference host does
    (samples/inference_udp_FPGA_custom_dispatch/main.cpp:143-148,217-221)."""
    fp = synth.float_params(seed, E=64)
    model = ITALSTMNetVIT_QAT(num_layers=1)
    sd = model.state_dict()
    for k, v in fp.items():
        assert k in sd and tuple(sd[k].shape) == v.shape, k
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fp.items()}, strict=True)
    model.attention_blocks.qconfig = ita_symmetric_qconfig     # training/qa_train.py:67-68
    model.ffn_blocks.qconfig = ita_symmetric_qconfig
    prepared = torch.ao.quantization.prepare_qat(model.train())
    prepared.lstm.dropout = 0.0
    with torch.no_grad():
        for it in range(4):                                    # observer calibration, dimmer frames
            prepared(to_X(synth.frames(100 * seed + 50 + it, 8, gain=0.8)))
    conv = torch.ao.quantization.convert(prepared.eval())
    for blk in conv.attention_blocks:
        blk.matmul2.matmul = patched_matmul2(blk.matmul2.scale, blk.matmul2.zero_point)

    tap = Tap()
    tap.add(conv.tokenizer.conv, "tok.conv")
    tap.add(conv.tokenizer, "tok.out")
    tap_attention(tap, conv.attention_blocks[0], "attn0.")
    tap_ffn(tap, conv.ffn_blocks[0], "ffn0.")
    tap.add(conv.norms1[0], "x1")
    tap.add(conv.norms2[0], "x2")
    tap.add(conv.pxShuffle, "tail.shuffled")
    tap.add(conv.up_sample, "tail.upsampled")
    tap.add(conv.down_sample, "tail.conv")
    tap.add(conv.decoder, "dec")

    fr0 = synth.frames(10 * seed, B)
    fr1 = synth.frames(10 * seed + 1, B)
    with torch.no_grad():
        vel0, (h0, c0) = conv(to_X(fr0, None))
        stage = dict(tap.t)
        vel1, (h1, c1) = conv(to_X(fr1, (h0, c0)))

    rec = {"meta.seed": np.int64(seed), "meta.B": np.int64(B), "meta.E": np.int64(64),
           "meta.params_sha256": np.array(synth.digest(fp)),
           "meta.torch": np.array(torch.__version__), "meta.engine": np.array("qnnpack")}
    rec.update(block_quant_record("attn0.", attn=conv.attention_blocks[0]))
    rec.update(block_quant_record("ffn0.", ffn=conv.ffn_blocks[0]))
    for k, v in fr0.items():
        rec["in0." + k] = v
    for k, v in fr1.items():
        rec["in1." + k] = v
    for k, v in stage.items():
        rec["s0." + k] = v
    rec["s0.vel"] = vel0.numpy(); rec["s0.h"] = h0.numpy(); rec["s0.c"] = c0.numpy()
    rec["s1.vel"] = vel1.numpy(); rec["s1.h"] = h1.numpy(); rec["s1.c"] = c1.numpy()
    # float32 intermediates that only feed later stages are stored as float32; the big
    # conv map (B,64,30,45) is dropped (the bilinear-sampled tokens pin it).
    rec.pop("s0.tok.conv", None)
    path = os.path.join(out_dir, f"vitlstm_E64_seed{seed}_B{B}.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_vit2l(seed, B, out_dir, upsample_shuffle_file=False):
    """The graph family WITHOUT the fusion tail: models/ITA/QAT/model.py:22-87 -- E = 128, two encoder layers, decoder
    Linear(E*S -> 512) on the flattened tokens, then the same LSTM head.  Same QAT flow as gen_vitlstm (qconfig on the
    attention / FFN blocks only, calibration, convert, the validation harness's matmul2).  This repo's synthetic
    parameters use the names norms1.i / norms2.i; the reference module calls them norm1_layers.i / norm2_layers.i."""
    # upsample_shuffle_file: the same graph as models/ITA_upsample_shuffle/QAT/model.py:36-115 declares it -- that file's forward
    # never calls its fusion layers (up_sample / pxShuffle / down_sample are constructed and unused), so it computes what
    # models/ITA/QAT/model.py computes, with decoder and nn_fc2 under spectral_norm (removed on the instance, see gen_float_twin)
    if upsample_shuffle_file:
        from models.ITA_upsample_shuffle.QAT.model import ITALSTMNetVIT_QAT as ITAViT2L
    else:
        from models.ITA.QAT.model import ITALSTMNetVIT_QAT as ITAViT2L
    fp = synth.float_params(seed, E=128, num_layers=2, tail=False)
    model = ITAViT2L()
    if upsample_shuffle_file:
        for lin in (model.decoder, model.nn_fc2):
            torch.nn.utils.remove_spectral_norm(lin)
            lin._load_state_dict_pre_hooks.clear()
    ren = lambda k: k.replace("norms1.", "norm1_layers.").replace("norms2.", "norm2_layers.")
    sd = model.state_dict()
    for k, v in fp.items():
        assert ren(k) in sd and tuple(sd[ren(k)].shape) == v.shape, k
    res = model.load_state_dict({ren(k): torch.from_numpy(v) for k, v in fp.items()}, strict=not upsample_shuffle_file)
    if upsample_shuffle_file:   # only the unused fusion conv may be missing from the synthetic parameters
        assert not res.unexpected_keys and all(k.startswith("down_sample.") for k in res.missing_keys), res
    model.attention_blocks.qconfig = ita_symmetric_qconfig     # training/qa_train.py:67-68
    model.ffn_blocks.qconfig = ita_symmetric_qconfig
    prepared = torch.ao.quantization.prepare_qat(model.train())
    prepared.lstm.dropout = 0.0
    with torch.no_grad():
        for it in range(4):
            prepared(to_X(synth.frames(100 * seed + 70 + it, 8, gain=0.8)))
    conv = torch.ao.quantization.convert(prepared.eval())
    for blk in conv.attention_blocks:
        blk.matmul2.matmul = patched_matmul2(blk.matmul2.scale, blk.matmul2.zero_point)
    tap = Tap()
    tap.add(conv.tokenizer, "tok.out")
    for i in range(2):
        tap_attention(tap, conv.attention_blocks[i], f"attn{i}.")
        tap_ffn(tap, conv.ffn_blocks[i], f"ffn{i}.")
        tap.add(conv.norm1_layers[i], f"x1_{i}")
        tap.add(conv.norm2_layers[i], f"x2_{i}")
    tap.add(conv.decoder, "dec")
    fr0, fr1 = synth.frames(10 * seed + 5, B), synth.frames(10 * seed + 6, B)
    with torch.no_grad():
        vel0, (h0, c0) = conv(to_X(fr0, None))
        stage = dict(tap.t)
        vel1, (h1, c1) = conv(to_X(fr1, (h0, c0)))
    rec = {"meta.seed": np.int64(seed), "meta.B": np.int64(B), "meta.E": np.int64(128), "meta.num_layers": np.int64(2),
           "meta.params_sha256": np.array(synth.digest(fp)), "meta.torch": np.array(torch.__version__),
           "meta.engine": np.array("qnnpack")}
    for i in range(2):
        rec.update(block_quant_record(f"attn{i}.", attn=conv.attention_blocks[i]))
        rec.update(block_quant_record(f"ffn{i}.", ffn=conv.ffn_blocks[i]))
    for k, v in fr0.items():
        rec["in0." + k] = v
    for k, v in fr1.items():
        rec["in1." + k] = v
    keep = ("tok.out", "x1_0", "x2_0", "x1_1", "x2_1", "dec", "attn0.x_q", "attn0.out_q", "ffn0.out_q", "attn1.x_q", "attn1.probs",
            "attn1.out_q", "ffn1.h1_relu", "ffn1.out_q")
    for k in keep:      # a selection: the per-stage int8 tensors of an E = 128 block are pinned by blocks_E128_*.npz already
        rec["s0." + k] = stage[k]
    rec["s0.vel"] = vel0.numpy(); rec["s0.h"] = h0.numpy(); rec["s0.c"] = c0.numpy()
    rec["s1.vel"] = vel1.numpy(); rec["s1.h"] = h1.numpy(); rec["s1.c"] = c1.numpy()
    path = os.path.join(out_dir, f"vit2l_{'us_' if upsample_shuffle_file else ''}E128_s{seed}_B{B}.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_vit1l(seed, B, out_dir):
    """models/ITA_single_layer/QAT/model.py:30-101 -- the single-layer member of the no-tail family: E = 128, ONE encoder
    layer (attributes attention_block / ffn_block / norm1 / norm2, no lists), decoder Linear(E*S -> 512) and nn_fc2 under
    spectral_norm (removed on the instance, as in gen_float_twin, so that the synthetic weights are the effective ones).
    Same QAT flow and the same record layout as gen_vit2l with num_layers = 1."""
    from models.ITA_single_layer.QAT.model import ITALSTMNetVIT_QAT as ITAViT1L
    fp = synth.float_params(seed, E=128, num_layers=1, tail=False)
    model = ITAViT1L()
    for lin in (model.decoder, model.nn_fc2):
        torch.nn.utils.remove_spectral_norm(lin)
        lin._load_state_dict_pre_hooks.clear()   # the removal leaves spectral_norm's load hook behind (it demands weight_orig)
    ren = lambda k: (k.replace("attention_blocks.0.", "attention_block.").replace("ffn_blocks.0.", "ffn_block.")
                      .replace("norms1.0.", "norm1.").replace("norms2.0.", "norm2."))
    sd = model.state_dict()
    for k, v in fp.items():
        assert ren(k) in sd and tuple(sd[ren(k)].shape) == v.shape, k
    model.load_state_dict({ren(k): torch.from_numpy(v) for k, v in fp.items()}, strict=True)
    model.attention_block.qconfig = ita_symmetric_qconfig     # training/qa_train.py:67-68
    model.ffn_block.qconfig = ita_symmetric_qconfig
    prepared = torch.ao.quantization.prepare_qat(model.train())
    prepared.lstm.dropout = 0.0
    with torch.no_grad():
        for it in range(4):
            prepared(to_X(synth.frames(100 * seed + 80 + it, 8, gain=0.8)))
    conv = torch.ao.quantization.convert(prepared.eval())
    conv.attention_block.matmul2.matmul = patched_matmul2(conv.attention_block.matmul2.scale, conv.attention_block.matmul2.zero_point)
    tap = Tap()
    tap.add(conv.tokenizer, "tok.out")
    tap_attention(tap, conv.attention_block, "attn0.")
    tap_ffn(tap, conv.ffn_block, "ffn0.")
    tap.add(conv.norm1, "x1_0")
    tap.add(conv.norm2, "x2_0")
    tap.add(conv.decoder, "dec")
    fr0, fr1 = synth.frames(10 * seed + 7, B), synth.frames(10 * seed + 8, B)
    with torch.no_grad():
        vel0, (h0, c0) = conv(to_X(fr0, None))
        stage = dict(tap.t)
        vel1, (h1, c1) = conv(to_X(fr1, (h0, c0)))
    rec = {"meta.seed": np.int64(seed), "meta.B": np.int64(B), "meta.E": np.int64(128), "meta.num_layers": np.int64(1),
           "meta.params_sha256": np.array(synth.digest(fp)), "meta.torch": np.array(torch.__version__),
           "meta.engine": np.array("qnnpack")}
    rec.update(block_quant_record("attn0.", attn=conv.attention_block))
    rec.update(block_quant_record("ffn0.", ffn=conv.ffn_block))
    for k, v in fr0.items():
        rec["in0." + k] = v
    for k, v in fr1.items():
        rec["in1." + k] = v
    for k in ("tok.out", "x1_0", "x2_0", "dec", "attn0.x_q", "attn0.probs", "attn0.out_q", "ffn0.h1_relu", "ffn0.out_q"):
        rec["s0." + k] = stage[k]
    rec["s0.vel"] = vel0.numpy(); rec["s0.h"] = h0.numpy(); rec["s0.c"] = c0.numpy()
    rec["s1.vel"] = vel1.numpy(); rec["s1.h"] = h1.numpy(); rec["s1.c"] = c1.numpy()
    path = os.path.join(out_dir, f"vit1l_E128_s{seed}_B{B}.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


class _Blocks(torch.nn.Module):
    """container so that prepare_qat/convert see the reference blocks as children"""

    def __init__(self, E, P, F):
        super().__init__()
        self.attn = ITASelfAttention_QAT(E, P, 1)
        self.ffn = ITAFeedForward_QAT(E, F)

    def forward(self, x):
        return self.attn(x), self.ffn(x)


def gen_blocks(seed, E, B, out_dir, gain_qk=3.0):
    """MHA / FFN blocks alone (BASELINE config 2: E=128 'ITA_single_layer' MHA bring-up)."""
    P, F, S = 192, 256, 128
    fp = synth.float_params(seed, E=E, gain_qk=gain_qk)
    m = _Blocks(E, P, F)
    sd = {}
    for k, v in fp.items():
        if k.startswith("attention_blocks.0."):
            sd["attn." + k[len("attention_blocks.0."):]] = torch.from_numpy(v)
        if k.startswith("ffn_blocks.0."):
            sd["ffn." + k[len("ffn_blocks.0."):]] = torch.from_numpy(v)
    m.load_state_dict(sd, strict=True)
    m.qconfig = ita_symmetric_qconfig
    prepared = torch.ao.quantization.prepare_qat(m.train())
    rs = np.random.RandomState(777 + seed)
    with torch.no_grad():
        for _ in range(4):
            prepared(torch.from_numpy((0.8 * rs.standard_normal((4, S, E))).astype(np.float32)))
    conv = torch.ao.quantization.convert(prepared.eval())
    conv.attn.matmul2.matmul = patched_matmul2(conv.attn.matmul2.scale, conv.attn.matmul2.zero_point)
    tap = Tap()
    tap_attention(tap, conv.attn, "attn0.")
    tap_ffn(tap, conv.ffn, "ffn0.")
    x = rs.standard_normal((B, S, E)).astype(np.float32)
    # a few outliers so the input quantiser and the requant stages saturate
    x[:, ::17, ::5] *= 4.0
    with torch.no_grad():
        conv(torch.from_numpy(x))
    rec = {"meta.seed": np.int64(seed), "meta.B": np.int64(B), "meta.E": np.int64(E),
           "meta.torch": np.array(torch.__version__), "meta.engine": np.array("qnnpack")}
    rec.update(block_quant_record("attn0.", attn=conv.attn))
    rec.update(block_quant_record("ffn0.", ffn=conv.ffn))
    rec["in0.x"] = x
    for k, v in tap.t.items():
        rec["s0." + k] = v
    path = os.path.join(out_dir, f"blocks_E{E}_seed{seed}_B{B}.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def gen_softmax(out_dir):
    """The reference's IntegerApproximatedSoftmax on adversarial int8 rows
    (models/ITA/QAT/ITA_softmax.py:19-73, quantized branch)."""
    from models.ITA.QAT.ITA_softmax import IntegerApproximatedSoftmax
    rs = np.random.RandomState(42)
    rows = []
    rows.append(np.zeros(128, np.int8))                       # all equal
    rows.append(np.full(128, -128, np.int8))
    rows.append(np.full(128, 127, np.int8))
    r = np.full(128, -128, np.int8); r[5] = 127; rows.append(r)   # one-hot max
    for d in (1, 7, 8, 9, 10, 31, 32, 33, 254, 255):              # shift boundaries (256>>8=1, >>9=0)
        r = np.full(128, 127 - d, np.int16); r[0] = 127; rows.append(r.astype(np.int8))
        r = np.full(128, 127, np.int16); r[3] = 127 - d; rows.append(r.astype(np.int8))
    rows.append(np.arange(-64, 64).astype(np.int8))
    rows.append(np.arange(127, -1, -1).astype(np.int8))
    for _ in range(96):
        rows.append(rs.randint(-128, 128, size=128).astype(np.int8))
    for sc in (1, 2, 3, 5, 8, 20):
        for _ in range(8):
            rows.append(np.clip(np.round(rs.standard_normal(128) * sc), -128, 127).astype(np.int8))
    x = np.stack(rows).astype(np.int8)
    sm = IntegerApproximatedSoftmax(dim=-1)
    q = torch._make_per_tensor_quantized_tensor(torch.from_numpy(x).reshape(1, 1, -1, 128), scale=0.1, zero_point=0)
    y = sm(q)
    assert y.dtype == torch.quint8 and y.q_zero_point() == 0
    np.savez_compressed(os.path.join(out_dir, "softmax_rows.npz"),
                        x=x, y=y.int_repr().numpy().reshape(-1, 128), out_scale=np.float64(y.q_scale()))
    # every possible denominator: sum in [1, 32768]  (ITA_softmax.py:56-61)
    sums = torch.arange(1, 32769, dtype=torch.int32)
    inv = torch.floor(((2 ** 8 - 1) * (2 ** 16)) / sums).to(torch.int32)
    np.savez_compressed(os.path.join(out_dir, "softmax_inverse.npz"), inv=inv.numpy())
    print("wrote softmax fixtures", x.shape)


def gen_tail_large(out_dir):
    """BASELINE config 5 / SURVEY.md section 8(d): the fusion-tail layers on a larger token grid with E = 128 and
    48 conv outputs -- nn.PixelShuffle(2), nn.Upsample(bilinear, align_corners=True) and nn.Conv2d(160, 48, 3,
    padding=1) exactly as the reference declares them (models/ITA_upsample_shuffle/model.py:70-78) and wires them
    (models/ITA_single_layer_upsample_shuffle/QAT/model.py:116-121); seeded inputs from synth.tail_large_case, only
    the expected output is stored."""
    synth = _load_synth()
    for (seed, E, th, tw, co, B) in ((0, 128, 4, 16, 48, 2), (1, 64, 8, 16, 9, 1)):
        c = synth.tail_large_case(seed, E, th, tw, co, B)
        x = torch.from_numpy(c["x"])
        x2d = x.transpose(1, 2).reshape(B, E, th, tw)
        up = torch.nn.Upsample(size=(2 * th, 2 * tw), mode="bilinear", align_corners=True)
        conv = torch.nn.Conv2d(E // 4 + E, co, 3, padding=1)
        with torch.no_grad():
            conv.weight.copy_(torch.from_numpy(c["conv_w"]))
            conv.bias.copy_(torch.from_numpy(c["conv_b"]))
            out = conv(torch.cat([torch.nn.PixelShuffle(2)(x2d), up(x2d)], dim=1))
        name = f"tail_large_E{E}_T{th}x{tw}_CO{co}_s{seed}.npz"   # (not "*_seed*": that glob selects the int8 block fixtures)
        np.savez_compressed(os.path.join(out_dir, name), out=out.numpy(), seed=seed, E=E, tok_h=th, tok_w=tw,
                            out_ch=co, B=B, inputs_sha256=synth.digest(c))
        print("wrote", name, tuple(out.shape))


def gen_float_twin(seed, B, out_dir):
    """The reference's FLOAT model (models/ITA_single_layer_upsample_shuffle/model.py:35-140 with the float blocks of
    models/ITA/layers.py: nn.Softmax attention) -- the graph its CPU .vmfb holds.  decoder and
    nn_fc2 are declared under spectral_norm (model.py:81,84); the parametrisation is removed on the instance
    (torch.nn.utils.remove_spectral_norm leaves a plain weight) so that the synthetic weights ARE the effective ones --
    in eval mode the module then computes exactly F.linear(x, weight, bias).  Two time steps, state carried."""
    from models.ITA_single_layer_upsample_shuffle.model import ITALSTMNetVIT
    fp = synth.float_params(seed, E=64)
    model = ITALSTMNetVIT(num_layers=1)
    for lin in (model.decoder, model.nn_fc2):
        torch.nn.utils.remove_spectral_norm(lin)
        lin._load_state_dict_pre_hooks.clear()   # the removal leaves spectral_norm's load hook behind (it demands weight_orig)
    sd = model.state_dict()
    for k, v in fp.items():
        assert k in sd and tuple(sd[k].shape) == v.shape, k
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fp.items()}, strict=True)
    model.eval()
    tap = Tap()
    tap.add(model.tokenizer, "tok.out")
    tap.add(model.norms1[0], "x1")
    tap.add(model.norms2[0], "x2")
    tap.add(model.decoder, "dec")
    fr0, fr1 = synth.frames(10 * seed, B), synth.frames(10 * seed + 1, B)
    with torch.no_grad():
        vel0, (h0, c0) = model(to_X(fr0, None))
        stage = dict(tap.t)
        vel1, (h1, c1) = model(to_X(fr1, (h0, c0)))
    rec = {"meta.seed": np.int64(seed), "meta.B": np.int64(B), "meta.params_sha256": np.array(synth.digest(fp)),
           "meta.torch": np.array(torch.__version__)}
    for k, v in fr0.items():
        rec["in0." + k] = v
    for k, v in fr1.items():
        rec["in1." + k] = v
    for k, v in stage.items():
        rec["s0." + k] = v
    rec["s0.vel"] = vel0.numpy(); rec["s0.h"] = h0.numpy(); rec["s0.c"] = c0.numpy()
    rec["s1.vel"] = vel1.numpy(); rec["s1.h"] = h1.numpy(); rec["s1.c"] = c1.numpy()
    path = os.path.join(out_dir, f"floattwin_E64_s{seed}_B{B}.npz")    # (not "*_seed*": that glob selects the int8 fixtures)
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default="", help="comma list of: softmax, vitlstm, blocks, tail_large, float_twin, vit2l, vit2l_us, vit1l (default: all)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    only = set(filter(None, a.only.split(",")))
    if only:
        if "softmax" in only: gen_softmax(a.out)
        if "vitlstm" in only:
            for seed in (0, 1, 2): gen_vitlstm(seed, 2, a.out)
        if "blocks" in only:
            for seed in (0, 1): gen_blocks(seed, 128, 1, a.out)
            gen_blocks(2, 64, 1, a.out, gain_qk=6.0)
        if "tail_large" in only: gen_tail_large(a.out)
        if "float_twin" in only: gen_float_twin(0, 2, a.out)
        if "vit2l" in only: gen_vit2l(0, 2, a.out)
        if "vit2l_us" in only: gen_vit2l(1, 2, a.out, upsample_shuffle_file=True)
        if "vit1l" in only: gen_vit1l(0, 2, a.out)
        return
    gen_float_twin(0, 2, a.out)
    gen_vit2l(0, 2, a.out)
    gen_vit2l(1, 2, a.out, upsample_shuffle_file=True)
    gen_vit1l(0, 2, a.out)
    gen_softmax(a.out)
    for seed in (0, 1, 2):
        gen_vitlstm(seed, 2, a.out)
    for seed in (0, 1):
        gen_blocks(seed, 128, 1, a.out)
    gen_blocks(2, 64, 1, a.out, gain_qk=6.0)
    gen_tail_large(a.out)


if __name__ == "__main__":
    main()
