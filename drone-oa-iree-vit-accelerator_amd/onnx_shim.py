"""ONNX entry shim: accept the reference's marker-op ONNX export and route it to
the MI355X engine.

The reference exports a float graph in which the two ITA blocks are placeholders -- `x + x` in the E = 64 model
(models/ITA_single_layer_upsample_shuffle/export/model.py:12-29,108-117), `neg` / `abs` in the E = 128 blocks
(models/ITA/export/ITA_ONNX.py:26,38) -- with opset 17 and fixed I/O names
(tests/export_onnx_for_FPGA.py:71-80: inputs image, additional_data, quat_data, hidden_in_h, hidden_in_c; outputs
output, hidden_out_h, hidden_out_c).  IREE then replaces the markers by the custom dispatch.  Here the graph is
not executed: it is RECOGNISED -- I/O contract, the marker ops at the two block positions of every layer, the float
layers around them -- and its initializers become the float half of the weight blob; the int8 half comes from the
converted checkpoint (params.record_from_state_dict), because a marker graph carries no attention / FFN weights.

    model  = onnx_shim.parse_model(open("ITAViTLSTM_float.onnx", "rb").read())
    info   = onnx_shim.match_itavitlstm(model)          # raises OnnxShimError with the first violated expectation
    blob   = params.blob_from_record(int8_record, info["float_params"], E=info["E"])

The `onnx` package is not part of the image, so the ModelProto is read by the small protobuf wire-format reader
below (field numbers from onnx.proto3: ModelProto.graph = 7, GraphProto.node = 1 / initializer = 5 / input = 11 /
output = 12, NodeProto.input = 1 / output = 2 / op_type = 4 / attribute = 5, TensorProto.dims = 1 / data_type = 2 /
float_data = 4 / name = 8 / raw_data = 9, ...).  No file produced by the reference's exporter is available in this
environment (torch.onnx.export itself needs `onnx`): the matcher is written against the exporter's documented
lowering of the layers (Conv, LayerNormalization, Gemm / MatMul + Add, LSTM with W [1, 4H, in] in gate order
i, o, f, c) and identifies tensors by graph structure and shape, not by initializer name; it is tested on graphs
built by tests/test_onnx_shim.py.  Parity with a real export is UNPINNED.
"""
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

IO_INPUTS = ("image", "additional_data", "quat_data", "hidden_in_h", "hidden_in_c")     # export_onnx_for_FPGA.py:78
IO_OUTPUTS = ("output", "hidden_out_h", "hidden_out_c")                                  # :79


class OnnxShimError(ValueError):
    pass


# --------------------------------------------------------------------------------- protobuf wire format
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        if pos >= len(buf):
            raise OnnxShimError("truncated varint")
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise OnnxShimError("varint too long")


def _fields(buf: bytes):
    """yields (field number, wire type, value): value is int for varint / fixed, bytes for length-delimited"""
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            if pos + n > len(buf):
                raise OnnxShimError("truncated length-delimited field")
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<i", buf, pos)[0]
            pos += 4
        else:
            raise OnnxShimError(f"unsupported wire type {wt}")
        yield fno, wt, v


def _signed(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


def _packed_varints(v) -> List[int]:
    if isinstance(v, int):
        return [_signed(v)]
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed(x))
    return out


# --------------------------------------------------------------------------------- ONNX messages
_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 6: np.int32, 7: np.int64, 10: np.float16, 11: np.float64}


def _tensor(buf: bytes) -> Tuple[str, np.ndarray]:
    dims, dtype, name, raw = [], 1, "", None
    floats, int32s, int64s = [], [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _packed_varints(v)
        elif fno == 2:
            dtype = v
        elif fno == 4:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", struct.pack("<i", v))[0]]
        elif fno == 5:
            int32s += _packed_varints(v)
        elif fno == 7:
            int64s += _packed_varints(v)
        elif fno == 8:
            name = v.decode()
        elif fno == 9:
            raw = v
        elif fno == 13 or (fno == 14 and v == 1):
            raise OnnxShimError("external tensor data is not supported")
    if dtype not in _DTYPES:
        raise OnnxShimError(f"tensor {name!r}: unsupported data type {dtype}")
    if raw is not None:
        arr = np.frombuffer(raw, dtype=_DTYPES[dtype]).copy()
    elif dtype == 1:
        arr = np.asarray(floats, np.float32)
    elif dtype == 7:
        arr = np.asarray(int64s, np.int64)
    else:
        arr = np.asarray(int32s).astype(_DTYPES[dtype])
    n = int(np.prod(dims)) if dims else arr.size
    if arr.size != n:
        raise OnnxShimError(f"tensor {name!r}: {arr.size} elements for dims {dims}")
    return name, arr.reshape(dims)


def _value_info(buf: bytes) -> Tuple[str, List]:
    name, shape = "", None
    for fno, _, v in _fields(buf):
        if fno == 1:
            name = v.decode()
        elif fno == 2:                                   # TypeProto
            for f2, _, v2 in _fields(v):
                if f2 == 1:                              # Tensor
                    for f3, _, v3 in _fields(v2):
                        if f3 == 2:                      # TensorShapeProto
                            shape = []
                            for f4, _, v4 in _fields(v3):
                                if f4 == 1:              # Dimension
                                    d = None
                                    for f5, _, v5 in _fields(v4):
                                        if f5 == 1:
                                            d = _signed(v5)
                                        elif f5 == 2:
                                            d = v5.decode()
                                    shape.append(d)
    return name, shape


@dataclass
class Node:
    op_type: str
    inputs: List[str]
    outputs: List[str]
    attrs: Dict[str, object] = field(default_factory=dict)
    name: str = ""


def _attribute(buf: bytes):
    name, val, ints, floats = "", None, [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = v.decode()
        elif fno == 2:
            val = struct.unpack("<f", struct.pack("<i", v))[0]
        elif fno == 3:
            val = _signed(v)
        elif fno == 4:
            val = v
        elif fno == 5:
            val = _tensor(v)[1]
        elif fno == 7:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", struct.pack("<i", v))[0]]
        elif fno == 8:
            ints += _packed_varints(v)
    if ints:
        val = ints
    elif floats:
        val = floats
    return name, val


def _node(buf: bytes) -> Node:
    n = Node("", [], [])
    for fno, _, v in _fields(buf):
        if fno == 1:
            n.inputs.append(v.decode())
        elif fno == 2:
            n.outputs.append(v.decode())
        elif fno == 3:
            n.name = v.decode()
        elif fno == 4:
            n.op_type = v.decode()
        elif fno == 5:
            k, a = _attribute(v)
            n.attrs[k] = a
    return n


@dataclass
class Model:
    opset: int
    inputs: List[Tuple[str, List]]
    outputs: List[Tuple[str, List]]
    nodes: List[Node]
    initializers: Dict[str, np.ndarray]


def parse_model(data: bytes) -> Model:
    graph, opset = None, 0
    for fno, _, v in _fields(data):
        if fno == 7:
            graph = v
        elif fno == 8:                                   # OperatorSetIdProto: domain = 1, version = 2
            dom, ver = "", 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    dom = v2.decode()
                elif f2 == 2:
                    ver = v2
            if dom in ("", "ai.onnx"):
                opset = ver
    if graph is None:
        raise OnnxShimError("not an ONNX ModelProto: no graph")
    m = Model(opset, [], [], [], {})
    for fno, _, v in _fields(graph):
        if fno == 1:
            m.nodes.append(_node(v))
        elif fno == 5:
            k, t = _tensor(v)
            m.initializers[k] = t
        elif fno == 11:
            m.inputs.append(_value_info(v))
        elif fno == 12:
            m.outputs.append(_value_info(v))
    m.inputs = [(n, s) for n, s in m.inputs if n not in m.initializers]      # old exporters list weights as inputs too
    return m


# --------------------------------------------------------------------------------- the ITAViTLSTM marker graph
def _const_of(model: Model, name: str):
    if name in model.initializers:
        return model.initializers[name]
    for n in model.nodes:                                # Constant nodes (do_constant_folding leaves some)
        if n.op_type == "Constant" and n.outputs and n.outputs[0] == name and "value" in n.attrs:
            return n.attrs["value"]
    return None


def _lstm_reorder(w: np.ndarray) -> np.ndarray:
    """ONNX gate order i, o, f, c (rows in four blocks of H)  ->  PyTorch i, f, g, o"""
    H = w.shape[0] // 4
    i, o, f, c = (w[k * H:(k + 1) * H] for k in range(4))
    return np.concatenate([i, f, c, o], axis=0)


def match_itavitlstm(model: Model) -> dict:
    """Checks that `model` is one of the reference's marker exports and returns
    {"E", "num_layers", "has_tail", "markers": [(node index, tensor)], "marker_kinds": ["attention" | "ffn" | "block"],
     "float_params": {state_dict-style name: array}}.  Two graph families:
      * ITAViTLSTM (models/ITA_single_layer_upsample_shuffle/export/model.py): E = 64, one layer, `x + x` markers, the
        3x3 fusion conv and a 4608 -> 512 decoder;
      * the no-tail family (models/ITA/export/ITA_ONNX.py:22-38,41-115 -- the graph ITA_spec.mlir:69-85 matches -- and its
        single-layer member): E = 128, `neg` (attention) / `abs` (feed-forward) markers, one or two layers, no 3x3 conv,
        decoder Linear(E * 128 -> 512) on the flattened tokens.
    LayerNorm parameters come back under this repo's `norms1.l` / `norms2.l` keys whatever the exporting module called
    its ModuleLists (`norm1_layers` in ITA_ONNX.py:61-62)."""
    if model.opset and model.opset < 17:
        raise OnnxShimError(f"opset {model.opset}: the reference exports opset 17 (LayerNormalization)")
    if tuple(n for n, _ in model.inputs) != IO_INPUTS:
        raise OnnxShimError(f"inputs {[n for n, _ in model.inputs]} != {list(IO_INPUTS)}")
    if tuple(n for n, _ in model.outputs) != IO_OUTPUTS:
        raise OnnxShimError(f"outputs {[n for n, _ in model.outputs]} != {list(IO_OUTPUTS)}")
    shp = dict(model.inputs)
    if shp["image"] is not None and list(shp["image"][1:]) != [1, 60, 90]:
        raise OnnxShimError(f"image shape {shp['image']} is not (B, 1, 60, 90)")
    if shp["hidden_in_h"] is not None and (shp["hidden_in_h"][0] != 3 or shp["hidden_in_h"][2] != 128):
        raise OnnxShimError(f"hidden_in_h shape {shp['hidden_in_h']} is not (3, B, 128)")
    fp, markers, kinds = {}, [], []
    convs, lns, lstms, linears = [], [], [], []
    for idx, n in enumerate(model.nodes):
        if n.op_type == "Conv":
            convs.append(n)
        elif n.op_type == "LayerNormalization":
            lns.append(n)
        elif n.op_type == "LSTM":
            lstms.append(n)
        elif n.op_type in ("Gemm", "MatMul"):
            linears.append((idx, n))
        elif n.op_type == "Add" and len(n.inputs) == 2 and n.inputs[0] == n.inputs[1]:
            markers.append((idx, n.inputs[0]))           # DummyHardwareBlock: x + x (export/model.py:29)
            kinds.append("block")                        # (attention or feed-forward by position)
        elif n.op_type in ("Neg", "Abs"):
            markers.append((idx, n.inputs[0]))           # ITA_ONNX.py:26 (attention), :38 (feed-forward)
            kinds.append("attention" if n.op_type == "Neg" else "ffn")
    # tokenizer conv 7x7 stride 2 and fusion conv 3x3
    tok = [c for c in convs if (_const_of(model, c.inputs[1]) is not None and _const_of(model, c.inputs[1]).shape[2:] == (7, 7))]
    dwn = [c for c in convs if (_const_of(model, c.inputs[1]) is not None and _const_of(model, c.inputs[1]).shape[2:] == (3, 3))]
    if len(tok) != 1 or len(dwn) > 1:
        raise OnnxShimError(f"expected one 7x7 Conv and at most one 3x3 Conv, found {len(tok)} and {len(dwn)}")
    has_tail = len(dwn) == 1
    wt = _const_of(model, tok[0].inputs[1])
    E = int(wt.shape[0])
    if wt.shape != (E, 1, 7, 7) or list(tok[0].attrs.get("strides", [])) != [2, 2] or list(tok[0].attrs.get("pads", [])) != [3, 3, 3, 3]:
        raise OnnxShimError("tokenizer Conv is not Conv2d(1, E, 7, stride 2, padding 3) (layers.py:30-37)")
    if has_tail:
        wd = _const_of(model, dwn[0].inputs[1])
        if wd.shape != (9, E // 4 + E, 3, 3) or list(dwn[0].attrs.get("pads", [])) != [1, 1, 1, 1]:
            raise OnnxShimError(f"fusion Conv weight {wd.shape} is not (9, 5E/4, 3, 3) with padding 1 (QAT/model.py:90)")
    for key, c in (("tokenizer.conv", tok[0]),) + ((("down_sample", dwn[0]),) if has_tail else ()):
        if len(c.inputs) < 3 or _const_of(model, c.inputs[2]) is None:
            raise OnnxShimError(f"{key}: Conv without a constant bias")
        fp[key + ".weight"] = np.asarray(_const_of(model, c.inputs[1]), np.float32)
        fp[key + ".bias"] = np.asarray(_const_of(model, c.inputs[2]), np.float32)
    # LayerNorms in graph order: tokenizer.norm, then norms1.l, norms2.l per layer
    if len(lns) < 3 or (len(lns) - 1) % 2:
        raise OnnxShimError(f"{len(lns)} LayerNormalization nodes: expected 1 + 2 per layer")
    L = (len(lns) - 1) // 2
    names = ["tokenizer.norm"] + [f"norms{k}.{l}" for l in range(L) for k in (1, 2)]
    for nm, n in zip(names, lns):
        s, b = _const_of(model, n.inputs[1]), _const_of(model, n.inputs[2]) if len(n.inputs) > 2 else None
        if s is None or b is None or s.shape != (E,) or b.shape != (E,):
            raise OnnxShimError(f"{nm}: LayerNormalization without constant scale / bias of length {E}")
        if abs(float(n.attrs.get("epsilon", 1e-5)) - 1e-5) > 1e-9:
            raise OnnxShimError(f"{nm}: epsilon {n.attrs.get('epsilon')} != 1e-5")
        fp[nm + ".weight"], fp[nm + ".bias"] = np.asarray(s, np.float32), np.asarray(b, np.float32)
    if len(markers) != 2 * L:
        raise OnnxShimError(f"{len(markers)} marker ops (x + x / neg / abs): expected {2 * L} (attention, feed-forward per layer)")
    for j, k in enumerate(kinds):                        # neg / abs must sit at their own block's position
        want = "attention" if j % 2 == 0 else "ffn"
        if k not in ("block", want):
            raise OnnxShimError(f"marker {j} is a {k} marker at a {want} position (ITA_ONNX.py:100-107: attention, then feed-forward)")
    # linears: decoder (4608 -> 512 behind the fusion tail, E * 128 -> 512 on the flattened tokens) and fc (128 -> 3)
    def linear(n_in, n_out, key):
        for idx, n in linears:
            w = _const_of(model, n.inputs[1])
            if w is None or w.ndim != 2:
                continue
            if n.op_type == "Gemm":
                if int(n.attrs.get("transB", 0)) == 1 and w.shape == (n_out, n_in):
                    wm = w
                elif int(n.attrs.get("transB", 0)) == 0 and w.shape == (n_in, n_out):
                    wm = w.T
                else:
                    continue
                b = _const_of(model, n.inputs[2]) if len(n.inputs) > 2 else None
            else:
                if w.shape != (n_in, n_out):
                    continue
                wm, b = w.T, None
                for m in model.nodes[idx + 1:idx + 3]:   # MatMul + Add(bias)
                    if m.op_type == "Add" and n.outputs[0] in m.inputs:
                        other = [x for x in m.inputs if x != n.outputs[0]]
                        b = _const_of(model, other[0]) if other else None
            if b is None or b.shape != (n_out,):
                raise OnnxShimError(f"{key}: Linear({n_in}, {n_out}) without a constant bias")
            fp[key + ".weight"] = np.ascontiguousarray(wm, np.float32)
            fp[key + ".bias"] = np.asarray(b, np.float32)
            return
        raise OnnxShimError(f"{key}: no Gemm / MatMul of shape {n_in} -> {n_out}")
    linear(4608 if has_tail else E * 128, 512, "decoder")
    linear(128, 3, "nn_fc2")
    if len(lstms) != 3:
        raise OnnxShimError(f"{len(lstms)} LSTM nodes: expected 3 (nn.LSTM(517, 128, num_layers=3), QAT/model.py:84)")
    for l, n in enumerate(lstms):
        W, R, B = (_const_of(model, x) if x else None for x in (n.inputs + ["", "", ""])[1:4])
        n_in = 517 if l == 0 else 128
        if W is None or R is None or B is None or W.shape != (1, 512, n_in) or R.shape != (1, 512, 128) or B.shape != (1, 1024):
            raise OnnxShimError(f"LSTM layer {l}: W / R / B are not constant (1,512,{n_in}) / (1,512,128) / (1,1024)")
        if int(n.attrs.get("hidden_size", 128)) != 128 or n.attrs.get("direction", b"forward") not in (b"forward", "forward"):
            raise OnnxShimError(f"LSTM layer {l}: hidden_size / direction")
        fp[f"lstm.weight_ih_l{l}"] = _lstm_reorder(np.asarray(W[0], np.float32))
        fp[f"lstm.weight_hh_l{l}"] = _lstm_reorder(np.asarray(R[0], np.float32))
        fp[f"lstm.bias_ih_l{l}"] = _lstm_reorder(np.asarray(B[0, :512], np.float32))
        fp[f"lstm.bias_hh_l{l}"] = _lstm_reorder(np.asarray(B[0, 512:], np.float32))
    return {"E": E, "num_layers": L, "has_tail": has_tail, "markers": markers,
            "marker_kinds": [("attention" if j % 2 == 0 else "ffn") if k == "block" else k for j, k in enumerate(kinds)],
            "float_params": fp}


# --------------------------------------------------------------------------------- the marker graph as MLIR text
# `iree-import-onnx model.onnx --opset-version 17 -o model.mlir` (tests/export_onnx_for_FPGA.py:86-89) is the second entry
# point: the same marker graph as text.  Two textual forms carry the markers, and both are read here:
#   * torch-onnx dialect (what iree-import-onnx writes):   torch.operator "onnx.Neg"(%x) : (!torch.vtensor<[1,128,128],f32>) -> ...
#   * linalg on tensors (what the transform spec matches, ITA_spec.mlir:69-85):
#         linalg.generic {...} ins(%x : tensor<1x128x128xf32>) outs(...) { ^bb0(...): %r = arith.negf %in : f32 ; linalg.yield %r }
#     with `math.absf` for the feed-forward marker (the reference ships no matcher for it) and
#     `arith.addf %in, %in` for the E = 64 export's `x + x`.
# Weights are not taken from MLIR text (they are dense resources there); this reader answers WHICH dispatches a module
# needs: marker kind, position and tensor shape, i.e. E, the number of layers and which plugin entry each marker gets.
import re as _re

_MLIR_SHAPE_VT = _re.compile(r"!torch\.vtensor<\[([0-9,\s]+)\],\s*f(?:32|16)>")
_MLIR_SHAPE_T = _re.compile(r"tensor<((?:\d+x)+)f(?:32|16)>")
_MLIR_ONNX_OP = _re.compile(r'torch\.operator\s+"onnx\.(Neg|Abs|Add)"\s*\(([^)]*)\)\s*:\s*\(([^)]*)\)')


def scan_mlir_markers(text: str) -> List[dict]:
    """-> [{"kind": "attention" | "ffn" | "block", "op": ..., "shape": (1, S, E), "form": "onnx" | "linalg", "line": n}]
    in textual order.  Raises OnnxShimError on text that holds a marker op whose tensor type cannot be read."""
    out = []
    lines = text.splitlines()
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = _MLIR_ONNX_OP.search(ln)
        if m:
            op, args, types = m.group(1), [a.strip() for a in m.group(2).split(",")], m.group(3)
            if op == "Add" and not (len(args) == 2 and args[0] == args[1]):
                i += 1
                continue                                  # an ordinary residual add
            sh = _MLIR_SHAPE_VT.search(types)
            if not sh:
                raise OnnxShimError(f"line {i + 1}: onnx.{op} marker without a static !torch.vtensor type")
            shape = tuple(int(x) for x in sh.group(1).replace(" ", "").split(",") if x)
            out.append({"kind": {"Neg": "attention", "Abs": "ffn", "Add": "block"}[op], "op": "onnx." + op, "shape": shape,
                        "form": "onnx", "line": i + 1})
            i += 1
            continue
        if "linalg.generic" in ln:
            # gather the op up to the end of its region: the closing "} -> tensor<...>" line
            j, body = i, []
            while j < len(lines):
                body.append(lines[j])
                if _re.search(r"\}\s*->\s*tensor<", lines[j]):
                    break
                j += 1
            blob = "\n".join(body)
            ops_in = _re.findall(r"=\s*((?:arith|math)\.[a-z_0-9]+)\b([^\n]*)", blob)
            payload = [(o, rest) for o, rest in ops_in if not o.startswith("arith.constant")]
            kind = None
            if len(payload) == 1:
                o, rest = payload[0]
                if o == "arith.negf":
                    kind = "attention"
                elif o == "math.absf":
                    kind = "ffn"
                elif o == "arith.addf":
                    a = [x.strip() for x in rest.split(":")[0].split(",")]
                    if len(a) == 2 and a[0] == a[1]:
                        kind = "block"
            ins_one = len(_re.findall(r"ins\(([^)]*)\)", blob)) == 1 and "," not in _re.findall(r"ins\(([^:)]*)", blob)[0]
            if kind and ins_one:
                sh = _MLIR_SHAPE_T.search(_re.findall(r"ins\(([^)]*)\)", blob)[0])
                if not sh:
                    raise OnnxShimError(f"line {i + 1}: marker linalg.generic without a static tensor type")
                shape = tuple(int(x) for x in sh.group(1).split("x") if x)
                out.append({"kind": kind, "op": payload[0][0], "shape": shape, "form": "linalg", "line": i + 1})
            i = j + 1
            continue
        i += 1
    return out


def match_mlir(text: str) -> dict:
    """Recognises a marker module: -> {"E", "S", "num_layers", "markers": scan_mlir_markers(text), "dispatch": [...]}
    where dispatch[j] names the plugin symbol marker j is routed to (include/ita_mi355x.h): `ITASelfAttention_workgroup`
    for attention markers (the symbol ITA_spec.mlir:30-33 imports), `ITAFeedForward_workgroup` for feed-forward ones.
    Checks: an even number of markers alternating attention / feed-forward, one (1, 128, E) shape throughout with
    E in {64, 128}, and -- when the text has a `func.func @main_graph` -- its five inputs (export_onnx_for_FPGA.py:78)."""
    mk = scan_mlir_markers(text)
    if not mk or len(mk) % 2:
        raise OnnxShimError(f"{len(mk)} marker ops found: expected two per encoder layer (attention, feed-forward)")
    shapes = {m["shape"] for m in mk}
    if len(shapes) != 1:
        raise OnnxShimError(f"markers of different shapes {sorted(shapes)}")
    shape = next(iter(shapes))
    if len(shape) != 3 or shape[1] != 128 or shape[2] not in (64, 128):
        raise OnnxShimError(f"marker tensor shape {shape} is not (B, 128, E) with E in (64, 128)")
    for j, m in enumerate(mk):
        want = "attention" if j % 2 == 0 else "ffn"
        if m["kind"] not in ("block", want):
            raise OnnxShimError(f"marker {j} (line {m['line']}) is a {m['kind']} marker at a {want} position")
    sig = _re.search(r"func\.func\s+@main_graph\s*\(([^{]*?)\)\s*->", text, _re.S)
    if sig:
        args = _re.findall(r"(%[\w.]+)\s*:\s*(![\w.]+<[^>]*>|tensor<[^>]*>)", sig.group(1))
        if len(args) != 5:
            raise OnnxShimError(f"@main_graph takes {len(args)} tensors: expected the five of {list(IO_INPUTS)}")
        t0 = args[0][1]
        dims = (_MLIR_SHAPE_VT.search(t0) or _MLIR_SHAPE_T.search(t0))
        if dims:
            d = [int(x) for x in _re.split(r"[x,]", dims.group(1).replace(" ", "")) if x]
            if d[1:] != [1, 60, 90]:
                raise OnnxShimError(f"@main_graph image argument {t0} is not (B, 1, 60, 90)")
    kinds = [("attention" if j % 2 == 0 else "ffn") for j in range(len(mk))]
    return {"E": shape[2], "S": shape[1], "num_layers": len(mk) // 2, "markers": mk,
            "dispatch": ["ITASelfAttention_workgroup" if k == "attention" else "ITAFeedForward_workgroup" for k in kinds]}
