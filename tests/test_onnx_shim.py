"""ONNX entry shim.  No `onnx` package and no file from the reference's exporter
exist here, so the marker graph is BUILT below with a minimal protobuf writer, following the graph that
tests/export_onnx_for_FPGA.py:71-80 exports from models/ITA_single_layer_upsample_shuffle/export/model.py
(op lowering of the TorchScript exporter at opset 17; ONNX LSTM gate order i, o, f, c), and the shim must recognise
it and hand back exactly the float parameters it was built from -- parity with a real export is unpinned."""
import os
import struct

import numpy as np
import pytest

from drone_oa_iree_vit_accelerator_amd import onnx_shim, params, synth


# ------------------------------------------------------------------ minimal protobuf writer (test side only)
def _vi(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(fno, payload):
    return _vi(fno << 3 | 2) + _vi(len(payload)) + payload


def _iv(fno, v):
    return _vi(fno << 3 | 0) + _vi(v)


def _tensor(name, arr, raw=True):
    arr = np.ascontiguousarray(arr)
    dt = {np.dtype(np.float32): 1, np.dtype(np.int64): 7}[arr.dtype]
    b = b"".join(_iv(1, d) for d in arr.shape) + _iv(2, dt)
    if raw:
        b += _ld(9, arr.tobytes())
    elif dt == 1:
        b += _ld(4, arr.astype("<f4").tobytes())            # packed float_data
    else:
        b += _ld(7, b"".join(_vi(int(x)) for x in arr.ravel()))
    return b + _ld(8, name.encode())


def _attr(name, v):
    b = _ld(1, name.encode())
    if isinstance(v, int):
        return b + _iv(3, v) + _iv(20, 2)
    if isinstance(v, float):
        return b + _vi(2 << 3 | 5) + struct.pack("<f", v) + _iv(20, 1)
    if isinstance(v, (bytes, str)):
        return b + _ld(4, v if isinstance(v, bytes) else v.encode()) + _iv(20, 3)
    return b + _ld(8, b"".join(_vi(int(x)) for x in v)) + _iv(20, 7)   # ints, packed


def _node(op, ins, outs, **attrs):
    b = b"".join(_ld(1, i.encode()) for i in ins) + b"".join(_ld(2, o.encode()) for o in outs)
    b += _ld(4, op.encode()) + b"".join(_ld(5, _attr(k, v)) for k, v in attrs.items())
    return b


def _vinfo(name, shape):
    dims = b"".join(_ld(1, _iv(1, d)) for d in shape)
    return _ld(1, name.encode()) + _ld(2, _ld(1, _iv(1, 1) + _ld(2, dims)))


def _onnx_lstm_order(w):          # PyTorch i, f, g, o -> ONNX i, o, f, c
    H = w.shape[0] // 4
    i, f, g, o = (w[k * H:(k + 1) * H] for k in range(4))
    return np.concatenate([i, o, f, g], axis=0)


def build_marker_onnx(fp, E=64, marker="addself", opset=17, inputs=onnx_shim.IO_INPUTS, raw=True, drop_marker=False,
                      num_layers=1, tail=True, swap_markers=False):
    """tail=False, E=128, marker="negabs": the graph of models/ITA/export/ITA_ONNX.py:88-115 (no fusion tail, the decoder
    reads the flattened tokens)"""
    init, nodes = [], []
    t = lambda name, arr: (init.append(_tensor(name, arr, raw)), name)[1]
    N = lambda *a, **k: nodes.append(_node(*a, **k))
    N("Conv", ["image", t("tokenizer.conv.weight", fp["tokenizer.conv.weight"]), t("tokenizer.conv.bias", fp["tokenizer.conv.bias"])],
      ["conv0"], dilations=[1, 1], group=1, kernel_shape=[7, 7], pads=[3, 3, 3, 3], strides=[2, 2])
    N("Resize", ["conv0", "", "", t("onnx::Resize_9", np.array([1, E, 8, 16], np.int64))], ["res0"], mode="linear",
      coordinate_transformation_mode="pytorch_half_pixel")
    N("Reshape", ["res0", t("onnx::Reshape_12", np.array([1, E, 128], np.int64))], ["flat0"])
    N("Transpose", ["flat0"], ["tok0"], perm=[0, 2, 1])
    N("LayerNormalization", ["tok0", t("tokenizer.norm.weight", fp["tokenizer.norm.weight"]), t("tokenizer.norm.bias", fp["tokenizer.norm.bias"])],
      ["x0"], axis=-1, epsilon=1e-5)
    x = "x0"
    for l in range(num_layers):
        for blk, key in ((f"attn{l}", f"norms1.{l}"), (f"ffn{l}", f"norms2.{l}")):
            is_attn = blk.startswith("attn")
            if not (drop_marker and not is_attn):
                if marker == "addself":
                    N("Add", [x, x], [blk + "_out"])
                else:
                    N("Neg" if is_attn != swap_markers else "Abs", [x], [blk + "_out"])
            else:
                N("Identity", [x], [blk + "_out"])
            N("Add", [x, blk + "_out"], [blk + "_res"])
            N("LayerNormalization", [blk + "_res", t(key + ".weight", fp[key + ".weight"]), t(key + ".bias", fp[key + ".bias"])],
              [blk + "_ln"], axis=-1, epsilon=1e-5)
            x = blk + "_ln"
    if tail:
        N("Transpose", [x], ["xt"], perm=[0, 2, 1])
        N("Reshape", ["xt", t("onnx::Reshape_40", np.array([1, E, 8, 16], np.int64))], ["x2d"])
        N("DepthToSpace", ["x2d"], ["shuf"], blocksize=2, mode="CRD")
        N("Resize", ["x2d", "", "", t("onnx::Resize_44", np.array([1, E, 16, 32], np.int64))], ["ups"], mode="linear",
          coordinate_transformation_mode="align_corners")
        N("Concat", ["shuf", "ups"], ["fused"], axis=1)
        N("Conv", ["fused", t("down_sample.weight", fp["down_sample.weight"]), t("down_sample.bias", fp["down_sample.bias"])], ["down"],
          dilations=[1, 1], group=1, kernel_shape=[3, 3], pads=[1, 1, 1, 1], strides=[1, 1])
        N("Flatten", ["down"], ["feat"], axis=1)
    else:
        N("Flatten", [x], ["feat"], axis=1)          # x.flatten(1): (1, 128 * E), ITA_ONNX.py:109
    N("Gemm", ["feat", t("decoder.weight", fp["decoder.weight"]), t("decoder.bias", fp["decoder.bias"])], ["dec"], alpha=1.0,
      beta=1.0, transB=1)
    N("Div", ["additional_data", t("onnx::Div_50", np.array([10.0], np.float32))], ["dv"])
    N("Concat", ["dec", "dv", "quat_data"], ["cat"], axis=1)
    N("Unsqueeze", ["cat", t("onnx::Unsqueeze_53", np.array([0], np.int64))], ["seq0"])
    seq = "seq0"
    for l in range(3):
        W = _onnx_lstm_order(fp[f"lstm.weight_ih_l{l}"])[None]
        R = _onnx_lstm_order(fp[f"lstm.weight_hh_l{l}"])[None]
        B = np.concatenate([_onnx_lstm_order(fp[f"lstm.bias_ih_l{l}"]), _onnx_lstm_order(fp[f"lstm.bias_hh_l{l}"])])[None]
        N("LSTM", [seq, t(f"onnx::LSTM_{100 + 3 * l}", W), t(f"onnx::LSTM_{101 + 3 * l}", R), t(f"onnx::LSTM_{102 + 3 * l}", B), "",
                   f"h_in{l}", f"c_in{l}"], [f"y{l}", f"h{l}", f"c{l}"], hidden_size=128)
        N("Squeeze", [f"y{l}", t(f"onnx::Squeeze_{200 + l}", np.array([1], np.int64))], [f"seq{l + 1}"])
        seq = f"seq{l + 1}"
    N("Squeeze", [seq, t("onnx::Squeeze_210", np.array([0], np.int64))], ["last"])
    if raw:
        N("Gemm", ["last", t("nn_fc2.weight", fp["nn_fc2.weight"]), t("nn_fc2.bias", fp["nn_fc2.bias"])], ["output"], alpha=1.0,
          beta=1.0, transB=1)
    else:       # the other lowering of nn.Linear: MatMul with the transposed weight, then Add
        N("MatMul", ["last", t("onnx::MatMul_220", np.ascontiguousarray(fp["nn_fc2.weight"].T))], ["mm"])
        N("Add", [t("nn_fc2.bias", fp["nn_fc2.bias"]), "mm"], ["output"])
    N("Concat", ["h0", "h1", "h2"], ["hidden_out_h"], axis=0)
    N("Concat", ["c0", "c1", "c2"], ["hidden_out_c"], axis=0)
    shapes = {"image": [1, 1, 60, 90], "additional_data": [1, 1], "quat_data": [1, 4], "hidden_in_h": [3, 1, 128],
              "hidden_in_c": [3, 1, 128]}
    g = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"main_graph") + b"".join(_ld(5, i) for i in init)
    g += b"".join(_ld(11, _vinfo(n, shapes.get(n, [1]))) for n in inputs)
    g += b"".join(_ld(12, _vinfo(n, s)) for n, s in (("output", [1, 3]), ("hidden_out_h", [3, 1, 128]), ("hidden_out_c", [3, 1, 128])))
    return _iv(1, 8) + _ld(2, b"pytorch") + _ld(7, g) + _ld(8, _ld(1, b"") + _iv(2, opset))


# ------------------------------------------------------------------ tests
FLOAT_KEYS = [k for k in synth.float_params(0, E=64) if not k.startswith(("attention_blocks", "ffn_blocks"))]


@pytest.mark.parametrize("raw", [True, False], ids=["raw_data_gemm", "typed_data_matmul"])
def test_marker_graph_recognised_and_weights_recovered(raw):
    fp = synth.float_params(3, E=64)
    info = onnx_shim.match_itavitlstm(onnx_shim.parse_model(build_marker_onnx(fp, raw=raw)))
    assert info["E"] == 64 and info["num_layers"] == 1 and len(info["markers"]) == 2
    assert sorted(info["float_params"]) == sorted(FLOAT_KEYS)
    for k in FLOAT_KEYS:
        assert np.array_equal(info["float_params"][k], fp[k]), k       # incl. the LSTM gate re-ordering


def test_neg_abs_markers_and_blob():
    """the E = 128 block exports mark attention with neg and feed-forward with abs (ITA_ONNX.py:26,38); the recovered
    float half plus the int8 record of a converted checkpoint gives the same blob as the original parameters"""
    fp = synth.float_params(0, E=64)
    info = onnx_shim.match_itavitlstm(onnx_shim.parse_model(build_marker_onnx(fp, marker="negabs")))
    assert len(info["markers"]) == 2
    fx = params.load_fixture(os.path.join(os.path.dirname(__file__), "golden", "vitlstm_E64_seed0_B2.npz"))
    assert params.blob_from_record(fx, info["float_params"], E=64) == params.blob_from_record(fx, fp, E=64)


def test_violations_are_reported():
    fp = synth.float_params(1, E=64)
    with pytest.raises(onnx_shim.OnnxShimError, match="inputs"):
        onnx_shim.match_itavitlstm(onnx_shim.parse_model(build_marker_onnx(
            fp, inputs=("img", "additional_data", "quat_data", "hidden_in_h", "hidden_in_c"))))
    with pytest.raises(onnx_shim.OnnxShimError, match="opset"):
        onnx_shim.match_itavitlstm(onnx_shim.parse_model(build_marker_onnx(fp, opset=13)))
    with pytest.raises(onnx_shim.OnnxShimError, match="marker"):
        onnx_shim.match_itavitlstm(onnx_shim.parse_model(build_marker_onnx(fp, drop_marker=True)))
    with pytest.raises(onnx_shim.OnnxShimError):
        onnx_shim.parse_model(b"\x0a\xff\xff")                        # truncated field
    with pytest.raises(onnx_shim.OnnxShimError, match="no graph"):
        onnx_shim.parse_model(_iv(1, 8))


def _no_tail_case():
    fx = params.load_fixture(os.path.join(os.path.dirname(__file__), "golden", "vit2l_E128_s0_B2.npz"))
    fp = synth.float_params(int(fx["meta.seed"]), E=128, num_layers=2, tail=False)
    return fx, fp


def test_no_tail_family_recognised():
    """models/ITA/export/ITA_ONNX.py: E = 128, two layers, neg / abs markers, no 3x3 conv, decoder on E * 128 inputs --
    the graph whose `negf 1x128x128` marker ITA_spec.mlir:69-85 matches.  Float half recovered exactly; with the int8
    record of the converted checkpoint it packs to the same blob as the original parameters."""
    fx, fp = _no_tail_case()
    data = build_marker_onnx(fp, E=128, marker="negabs", num_layers=2, tail=False)
    info = onnx_shim.match_itavitlstm(onnx_shim.parse_model(data))
    assert info["E"] == 128 and info["num_layers"] == 2 and info["has_tail"] is False
    assert info["marker_kinds"] == ["attention", "ffn", "attention", "ffn"]
    keys = [k for k in fp if not k.startswith(("attention_blocks", "ffn_blocks"))]
    assert sorted(info["float_params"]) == sorted(keys)
    for k in keys:
        assert np.array_equal(info["float_params"][k], fp[k]), k
    assert info["float_params"]["decoder.weight"].shape == (512, 128 * 128)
    assert (params.blob_from_record(fx, info["float_params"], E=128, num_layers=2)
            == params.blob_from_record(fx, fp, E=128, num_layers=2))
    # the single-layer member of the family, and the ITAViTLSTM graph still reports its tail
    fp1 = synth.float_params(0, E=128, num_layers=1, tail=False)
    i1 = onnx_shim.match_itavitlstm(onnx_shim.parse_model(build_marker_onnx(fp1, E=128, marker="negabs", tail=False)))
    assert (i1["E"], i1["num_layers"], i1["has_tail"]) == (128, 1, False)
    i64 = onnx_shim.match_itavitlstm(onnx_shim.parse_model(build_marker_onnx(synth.float_params(0, E=64))))
    assert i64["has_tail"] is True and i64["marker_kinds"] == ["attention", "ffn"]
    with pytest.raises(onnx_shim.OnnxShimError, match="position"):       # abs where the attention marker belongs
        onnx_shim.match_itavitlstm(onnx_shim.parse_model(build_marker_onnx(fp, E=128, marker="negabs", num_layers=2,
                                                                         tail=False, swap_markers=True)))
    with pytest.raises(onnx_shim.OnnxShimError, match="decoder"):        # a tail-sized decoder in a no-tail graph
        bad = dict(fp)
        bad["decoder.weight"] = np.zeros((512, 4608), np.float32)
        onnx_shim.match_itavitlstm(onnx_shim.parse_model(build_marker_onnx(bad, E=128, marker="negabs", num_layers=2, tail=False)))


MLIR_ONNX_FORM = """
module {
  func.func @main_graph(%arg0: !torch.vtensor<[1,1,60,90],f32>, %arg1: !torch.vtensor<[1,1],f32>, %arg2: !torch.vtensor<[1,4],f32>, %arg3: !torch.vtensor<[3,1,128],f32>, %arg4: !torch.vtensor<[3,1,128],f32>) -> (!torch.vtensor<[1,3],f32>, !torch.vtensor<[3,1,128],f32>, !torch.vtensor<[3,1,128],f32>) attributes {torch.onnx_meta.opset_version = 17 : si64} {
    %12 = torch.operator "onnx.LayerNormalization"(%11, %0, %1) {torch.onnx.axis = -1 : si64, torch.onnx.epsilon = 9.99999974E-6 : f32} : (!torch.vtensor<[1,128,128],f32>, !torch.vtensor<[128],f32>, !torch.vtensor<[128],f32>) -> !torch.vtensor<[1,128,128],f32>
    %13 = torch.operator "onnx.Neg"(%12) : (!torch.vtensor<[1,128,128],f32>) -> !torch.vtensor<[1,128,128],f32>
    %14 = torch.operator "onnx.Add"(%12, %13) : (!torch.vtensor<[1,128,128],f32>, !torch.vtensor<[1,128,128],f32>) -> !torch.vtensor<[1,128,128],f32>
    %16 = torch.operator "onnx.Abs"(%15) : (!torch.vtensor<[1,128,128],f32>) -> !torch.vtensor<[1,128,128],f32>
    %19 = torch.operator "onnx.Neg"(%18) : (!torch.vtensor<[1,128,128],f32>) -> !torch.vtensor<[1,128,128],f32>
    %22 = torch.operator "onnx.Abs"(%21) : (!torch.vtensor<[1,128,128],f32>) -> !torch.vtensor<[1,128,128],f32>
    return %40, %41, %42 : !torch.vtensor<[1,3],f32>, !torch.vtensor<[3,1,128],f32>, !torch.vtensor<[3,1,128],f32>
  }
}
"""

MLIR_LINALG_FORM = """
#map1 = affine_map<(d0, d1, d2) -> (d0, d1, d2)>
util.func public @main_graph(%arg0: tensor<1x1x60x90xf32>) -> tensor<1x3xf32> {
  %7 = tensor.empty() : tensor<1x128x128xf32>
  %8 = linalg.generic {indexing_maps = [#map1, #map1], iterator_types = ["parallel", "parallel", "parallel"]} ins(%6 : tensor<1x128x128xf32>) outs(%7 : tensor<1x128x128xf32>) {
  ^bb0(%in: f32, %out: f32):
    %res = arith.negf %in : f32
    linalg.yield %res : f32
  } -> tensor<1x128x128xf32>
  %9 = linalg.generic {indexing_maps = [#map1, #map1, #map1], iterator_types = ["parallel", "parallel", "parallel"]} ins(%6, %8 : tensor<1x128x128xf32>, tensor<1x128x128xf32>) outs(%7 : tensor<1x128x128xf32>) {
  ^bb0(%in: f32, %in_0: f32, %out: f32):
    %r = arith.addf %in, %in_0 : f32
    linalg.yield %r : f32
  } -> tensor<1x128x128xf32>
  %11 = linalg.generic {indexing_maps = [#map1, #map1], iterator_types = ["parallel", "parallel", "parallel"]} ins(%10 : tensor<1x128x128xf32>) outs(%7 : tensor<1x128x128xf32>) {
  ^bb0(%in: f32, %out: f32):
    %res = math.absf %in : f32
    linalg.yield %res : f32
  } -> tensor<1x128x128xf32>
  util.return %99 : tensor<1x3xf32>
}
"""


def test_mlir_marker_reader():
    """the marker graph as MLIR text (iree-import-onnx output, tests/export_onnx_for_FPGA.py:86-89, and the linalg form
    the transform spec matches, ITA_spec.mlir:69-85): markers, shapes and the plugin symbol each is routed to"""
    m = onnx_shim.match_mlir(MLIR_ONNX_FORM)
    assert (m["E"], m["S"], m["num_layers"]) == (128, 128, 2)
    assert [x["kind"] for x in m["markers"]] == ["attention", "ffn", "attention", "ffn"]
    assert all(x["form"] == "onnx" and x["shape"] == (1, 128, 128) for x in m["markers"])
    assert m["dispatch"] == ["ITASelfAttention_workgroup", "ITAFeedForward_workgroup"] * 2
    l = onnx_shim.match_mlir(MLIR_LINALG_FORM)
    assert (l["E"], l["num_layers"]) == (128, 1) and [x["op"] for x in l["markers"]] == ["arith.negf", "math.absf"]
    # the E = 64 export's x + x markers in the linalg form: `arith.addf %in, %in` on ONE input
    e64 = MLIR_LINALG_FORM.replace("128x128xf32", "128x64xf32").replace("arith.negf %in", "arith.addf %in, %in") \
                          .replace("math.absf %in", "arith.addf %in, %in")
    k = onnx_shim.match_mlir(e64)
    assert k["E"] == 64 and [x["kind"] for x in k["markers"]] == ["block", "block"]
    # this repo's own transform spec carries the negf pattern it matches on (plugin/ita_mi355x_spec.mlir)
    spec = open(os.path.join(os.path.dirname(onnx_shim.__file__), "plugin", "ita_mi355x_spec.mlir")).read()
    assert any(x["kind"] == "attention" and x["shape"][-2:] == (128, 128) for x in onnx_shim.scan_mlir_markers(spec))
    with pytest.raises(onnx_shim.OnnxShimError, match="marker"):
        onnx_shim.match_mlir("module { }")
    with pytest.raises(onnx_shim.OnnxShimError, match="position"):
        onnx_shim.match_mlir(MLIR_ONNX_FORM.replace('"onnx.Neg"(%12)', '"onnx.Abs"(%12)'))
    with pytest.raises(onnx_shim.OnnxShimError, match="five"):
        onnx_shim.match_mlir(MLIR_ONNX_FORM.replace(", %arg4: !torch.vtensor<[3,1,128],f32>", ""))


@pytest.mark.gpu
def test_no_tail_shim_blob_runs_on_the_engine_and_equals_oracle(oracle):
    """row n3 for the family the MLIR spec matches: neg/abs marker ONNX bytes (E = 128, two layers, no fusion tail) ->
    onnx_shim -> blob with the int8 record of the vit2l fixture -> Engine.forward, against oracle.forward on the blob
    packed from the original parameters, and the reference's fixture within the end-to-end int8 bound."""
    import torch
    from drone_oa_iree_vit_accelerator_amd import host
    fx, fp = _no_tail_case()
    info = onnx_shim.match_itavitlstm(onnx_shim.parse_model(build_marker_onnx(fp, E=128, marker="negabs", num_layers=2,
                                                                              tail=False, raw=False)))
    blob = params.blob_from_record(fx, info["float_params"], E=info["E"], num_layers=info["num_layers"])
    ref_blob = params.blob_from_record(fx, fp, E=128, num_layers=2)
    eng = host.Engine(blob, device=0)
    cu = lambda a: torch.from_numpy(a).cuda()
    vel, (h, c), tp = eng.forward(cu(fx["in0.img_u8"]), cu(fx["in0.desvel"]), cu(fx["in0.quat"]), taps=True)
    ovel, oh, oc, otp = oracle.forward(ref_blob, fx["in0.img_u8"], fx["in0.desvel"], fx["in0.quat"], taps=True)
    for k in ("tokens", "x1", "x2"):
        np.testing.assert_array_equal(tp[k].cpu().numpy(), otp[k], err_msg=k)
    np.testing.assert_allclose(vel.cpu().numpy(), ovel, atol=2e-5, rtol=0)
    np.testing.assert_allclose(h.cpu().numpy(), oh, atol=2e-5, rtol=0)
    np.testing.assert_allclose(vel.cpu().numpy(), fx["s0.vel"], atol=5e-4, rtol=0)
    eng.close()


@pytest.mark.gpu
def test_shim_blob_runs_on_the_engine_and_equals_oracle(oracle):
    """row n3 end to end: marker-op ONNX bytes -> onnx_shim -> float half of the blob (+ the int8 record of a converted
    checkpoint) -> Engine.forward on the GPU, against oracle.forward on a blob packed from the ORIGINAL parameters.
    (The ONNX bytes come from this test's own writer: a real export cannot be produced here -- `onnx` is not in the
    image -- so parity with the reference exporter's file stays unpinned.)"""
    import torch
    from drone_oa_iree_vit_accelerator_amd import host
    fp = synth.float_params(2, E=64)
    fx = params.load_fixture(os.path.join(os.path.dirname(__file__), "golden", "vitlstm_E64_seed2_B2.npz"))
    info = onnx_shim.match_itavitlstm(onnx_shim.parse_model(build_marker_onnx(fp, raw=False)))
    blob = params.blob_from_record(fx, info["float_params"], E=64)
    ref_blob = params.blob_from_record(fx, fp, E=64)
    eng = host.Engine(blob, device=0)
    cu = lambda a: torch.from_numpy(a).cuda()
    vel, (h, c), tp = eng.forward(cu(fx["in0.img_u8"]), cu(fx["in0.desvel"]), cu(fx["in0.quat"]), taps=True)
    ovel, oh, oc, otp = oracle.forward(ref_blob, fx["in0.img_u8"], fx["in0.desvel"], fx["in0.quat"], taps=True)
    for k in ("tokens", "x1", "x2"):
        np.testing.assert_array_equal(tp[k].cpu().numpy(), otp[k], err_msg=k)
    np.testing.assert_allclose(vel.cpu().numpy(), ovel, atol=2e-5, rtol=0)
    np.testing.assert_allclose(h.cpu().numpy(), oh, atol=2e-5, rtol=0)
    np.testing.assert_allclose(c.cpu().numpy(), oc, atol=2e-5, rtol=0)
    np.testing.assert_allclose(vel.cpu().numpy(), fx["s0.vel"], atol=5e-4, rtol=0)     # and the reference fixture
    eng.close()
