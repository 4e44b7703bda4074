"""Host side of the MI355X ITA engine: ctypes binding of ``csrc/libita_mi355x.so`` (C ABI in
include/ita_mi355x.h) and a mirror of the reference's model interface.

PyTorch is used for device memory, streams and (in bench.py) torch.distributed only; all
arithmetic happens in the HIP kernels behind the C ABI.  There is deliberately NO CPU or
eager fallback: if the extension is missing or no GPU is visible, calls raise.

Reference interface mirrored here (all paths relative to the reference repository):
  ITALSTMNetVIT_QAT.forward(X)   models/ITA_single_layer_upsample_shuffle/QAT/model.py:93-132
      X = [img (B,1,60,90), desvel (B,1), quat (B,4) | None, (h, c) | None] -> (vel (B,3), (h, c))
  refine_inputs                  same file :22-31 (default quaternion, resize to 60x90)
  ITASelfAttention_QAT / ITAFeedForward_QAT   models/ITA/QAT/layers.py:47-127
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.path.join(_CSRC, "libita_mi355x.so")
_REPO = os.path.dirname(_HERE)
_LIB = None

IMAGE_F32, IMAGE_U8 = 0, 1
DISPATCH_F16, DISPATCH_F32 = 0, 1

EXPORTED_SYMBOLS = (
    "ita_abi_version", "ita_create", "ita_destroy", "ita_load_weights", "ita_validate_blob", "ita_reserve", "ita_get_dims",
    "ita_last_error", "ita_error_string", "ita_mha_int8", "ita_mha_int8_taps", "ita_mha_q8", "ita_mha_long_q8", "ita_ffn_int8", "ita_ffn_int8_taps",
    "ita_encoder_layer", "ita_tokenizer", "ita_fusion_tail", "ita_vitlstm_forward", "ita_bind_dispatch",
    "ita_profile_begin", "ita_profile_begin_sampled", "ita_profile_end", "ita_set_tail_mode", "ita_debug_encoder_stamps",
    "ita_fusion_tail_load", "ita_fusion_tail_large",
    "ita_wire_unpack_packet", "ita_wire_postprocess", "ita_vitlstm_forward_slots", "ita_vitlstm_front",
    "ita_vitlstm_back", "ita_vitlstm_pipelined", "ita_vitlstm_front_ev", "ita_vitlstm_encode", "ita_vitlstm_fold", "ita_vitlstm_tail", "ita_debug_softmax_rows",
    "ITASelfAttention_workgroup", "ITASelfAttention_workgroup_expanded", "ITAFeedForward_workgroup",
)

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function"]


class ITAError(RuntimeError):
    pass


def build_extension(force: bool = False, verbose: bool = False) -> str:
    """hipcc cross-compiles the plugin for gfx950 in-tree (works without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h"))]
    srcs += [os.path.join(_REPO, "include", f) for f in ("ita_mi355x.h", "ita_weights.h")]
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs):
        return _SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + ["-I", os.path.join(_REPO, "include"), os.path.join(_CSRC, "ita_plugin.hip"),
                                   "-o", _SO]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return _SO


def build_samples(force: bool = False, verbose: bool = False) -> str:
    """the UDP inference host (reference samples/inference_udp_FPGA_custom_dispatch/main.cpp)"""
    src = os.path.join(_HERE, "samples", "ita_udp_server.cpp")
    exe = os.path.join(_HERE, "samples", "ita_udp_server")
    deps = [src, _SO, os.path.join(_REPO, "include", "ita_wire.h"), os.path.join(_REPO, "include", "ita_mi355x.h")]
    if not force and os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps):
        return exe
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O2", "-std=c++17", "-Wall", "-I", os.path.join(_REPO, "include"), src, "-L", _CSRC,
           "-lita_mi355x", "-Wl,-rpath,$ORIGIN/../csrc", "-o", exe]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return exe


class _MhaTaps(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x_q", "Q", "K", "V", "logits", "probs", "ctx", "out_q")]


class _FfnTaps(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x_q", "h", "out_q")]


class _FwdTaps(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("tokens", "x1", "x2", "feat", "dec")]


def lib():
    """Loads the C-ABI library; fails loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise ITAError(f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
        L = C.CDLL(_SO)
        L.ita_error_string.restype = C.c_char_p
        vp, i = C.c_void_p, C.c_int
        L.ita_create.argtypes = [C.POINTER(vp), i]
        L.ita_destroy.argtypes = [vp]
        L.ita_load_weights.argtypes = [vp, vp, C.c_size_t]
        L.ita_reserve.argtypes = [vp, i]
        L.ita_get_dims.argtypes = [vp] + [C.POINTER(i)] * 6
        L.ita_mha_int8.argtypes = [vp, i, vp, vp, i, vp]
        L.ita_mha_int8_taps.argtypes = [vp, i, vp, vp, i, C.POINTER(_MhaTaps), vp]
        L.ita_ffn_int8.argtypes = [vp, i, vp, vp, i, vp]
        L.ita_ffn_int8_taps.argtypes = [vp, i, vp, vp, i, C.POINTER(_FfnTaps), vp]
        L.ita_encoder_layer.argtypes = [vp, i, vp, vp, i, vp]
        L.ita_tokenizer.argtypes = [vp, vp, i, vp, i, vp]
        L.ita_fusion_tail.argtypes = [vp, vp, vp, i, vp]
        L.ita_vitlstm_forward.argtypes = [vp, vp, i, vp, vp, vp, vp, vp, vp, vp, i, C.POINTER(_FwdTaps), vp]
        L.ita_vitlstm_forward_slots.argtypes = [vp, vp, i, vp, vp, vp, vp, vp, i, vp, i, vp]
        L.ita_vitlstm_front.argtypes = [vp, vp, i, i, i, vp]
        L.ita_vitlstm_front_ev.argtypes = [vp, vp, i, i, i, vp, vp]
        L.ita_vitlstm_encode.argtypes = [vp, vp, i, i, i, vp]
        L.ita_vitlstm_fold.argtypes = [vp, i, i, i, vp]
        L.ita_vitlstm_back.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, i, vp]
        L.ita_vitlstm_pipelined.argtypes = [vp, vp, i, vp, vp, vp, vp, vp, i, i, vp, vp]
        L.ita_mha_q8.argtypes = [vp, i, vp, vp, i, vp]
        L.ita_mha_long_q8.argtypes = [vp, i, vp, vp, i, i, vp]
        L.ita_vitlstm_tail.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i, vp]
        L.ita_debug_softmax_rows.argtypes = [vp, vp, vp, i, vp]
        L.ita_validate_blob.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.ita_bind_dispatch.argtypes = [vp, i, i]
        L.ita_wire_unpack_packet.argtypes = [vp, C.c_size_t, i, vp]
        L.ita_wire_postprocess.argtypes = [vp, C.c_float, C.c_float, vp]
        L.ita_wire_postprocess.restype = None
        L.ita_set_tail_mode.argtypes = [vp, i]
        L.ita_debug_encoder_stamps.argtypes = [vp, i, vp, vp, vp, i, vp, vp]
        L.ita_fusion_tail_load.argtypes = [vp, vp, vp, i, i]
        L.ita_fusion_tail_large.argtypes = [vp, vp, vp, i, i, i, vp]
        L.ita_profile_begin.argtypes = [vp, i]
        L.ita_profile_begin_sampled.argtypes = [vp, i, i, i]
        L.ita_profile_end.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(i)]
        L.ITASelfAttention_workgroup.argtypes = [vp, vp]
        L.ITASelfAttention_workgroup.restype = None
        L.ITAFeedForward_workgroup.argtypes = [vp, vp]
        L.ITAFeedForward_workgroup.restype = None
        L.ITASelfAttention_workgroup_expanded.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t] * 2
        L.ITASelfAttention_workgroup_expanded.restype = None
        _LIB = L
    return _LIB


def _chk(rc: int):
    if rc != 0:
        raise ITAError(f"ita status {rc}: {lib().ita_error_string().decode()}")


def _torch():
    import torch
    return torch


def np_prod(shape) -> int:
    n = 1
    for d in shape:
        n *= int(d)
    return n


def _stream_ptr(device=None):
    """handle of torch's current stream ON `device` (an Engine passes its own ordinal: the current stream of whatever
    device happens to be current would belong to another GPU)"""
    return C.c_void_p(_torch().cuda.current_stream(device).cuda_stream)


def _dev_f32(t, shape=None):
    torch = _torch()
    if not t.is_cuda:
        raise ITAError("tensors handed to the ITA engine must live on the GPU (no CPU fallback)")
    t = t.contiguous()
    if t.dtype != torch.float32:
        t = t.float()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ITAError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


class Engine:
    """One ita_context: weights resident on one GPU, stateless compute calls on caller-owned tensors."""

    def __init__(self, blob: bytes, device: Optional[int] = None, reserve: int = 0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise ITAError("no GPU visible: the ITA engine has no CPU path")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self._h = C.c_void_p()
        _chk(lib().ita_create(C.byref(self._h), self.device))
        buf = C.create_string_buffer(blob, len(blob))
        _chk(lib().ita_load_weights(self._h, buf, len(blob)))
        d = [C.c_int() for _ in range(6)]
        _chk(lib().ita_get_dims(self._h, *[C.byref(x) for x in d]))
        self.E, self.S, self.P, self.F, self.H, self.num_layers = [x.value for x in d]
        self._reserved = 0          # frames the workspace is pinned for (ita_reserve); grows only
        self._graphs = []           # weakrefs of live captured graphs: they hold raw workspace pointers
        if reserve:
            self.reserve(reserve)

    def reserve(self, batch: int):
        """Size and pin the workspace for `batch` frames (ita_reserve).  Growing it frees and reallocates every internal
        buffer, which would leave a HIP graph captured earlier on this engine replaying freed pointers: refused while
        such a graph is alive (build the largest graph first, or use one Engine per batch size)."""
        batch = int(batch)
        if batch <= self._reserved:
            return
        self._graphs = [g for g in self._graphs if g() is not None]
        if self._graphs:
            raise ITAError(f"reserve({batch}) would reallocate the workspace (pinned for {self._reserved} frames) under "
                           f"{len(self._graphs)} live captured graph(s) of this engine: delete them first, or create the "
                           "engine with reserve= the largest batch")
        _chk(lib().ita_reserve(self._h, batch))
        self._reserved = batch

    def _track_graph(self, g):
        import weakref
        self._graphs.append(weakref.ref(g))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().ita_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- int8 blocks -------------------------------------------------------------------
    def mha(self, x, layer: int = 0, taps: bool = False):
        """ITASelfAttention_QAT.forward: (B,128,E) f32 -> (B,128,E) f32 [, dict of int tensors]."""
        torch = _torch()
        x = _dev_f32(x)
        B = x.shape[0]
        y = torch.empty_like(x)
        if not taps:
            _chk(lib().ita_mha_int8(self._h, layer, x.data_ptr(), y.data_ptr(), B, _stream_ptr(self.device)))
            return y
        mk = lambda *s, dt=torch.int8: torch.empty(s, dtype=dt, device=x.device)
        t = dict(x_q=mk(B, 128, self.E), Q=mk(B, 128, self.P), K=mk(B, 128, self.P), V=mk(B, 128, self.P),
                 logits=mk(B, 128, 128), probs=mk(B, 128, 128, dt=torch.uint8), ctx=mk(B, 128, self.P),
                 out_q=mk(B, 128, self.E))
        st = _MhaTaps(**{k: v.data_ptr() for k, v in t.items()})
        _chk(lib().ita_mha_int8_taps(self._h, layer, x.data_ptr(), y.data_ptr(), B, C.byref(st), _stream_ptr(self.device)))
        return y, t

    def ffn(self, x, layer: int = 0, taps: bool = False):
        torch = _torch()
        x = _dev_f32(x)
        B = x.shape[0]
        y = torch.empty_like(x)
        if not taps:
            _chk(lib().ita_ffn_int8(self._h, layer, x.data_ptr(), y.data_ptr(), B, _stream_ptr(self.device)))
            return y
        mk = lambda *s: torch.empty(s, dtype=torch.int8, device=x.device)
        t = dict(x_q=mk(B, 128, self.E), h=mk(B, 128, self.F), out_q=mk(B, 128, self.E))
        st = _FfnTaps(**{k: v.data_ptr() for k, v in t.items()})
        _chk(lib().ita_ffn_int8_taps(self._h, layer, x.data_ptr(), y.data_ptr(), B, C.byref(st), _stream_ptr(self.device)))
        return y, t

    def encoder_layer(self, x, layer: int = 0):
        x = _dev_f32(x)
        y = _torch().empty_like(x)
        _chk(lib().ita_encoder_layer(self._h, layer, x.data_ptr(), y.data_ptr(), x.shape[0], _stream_ptr(self.device)))
        return y

    # ---- float stages ------------------------------------------------------------------
    def tokenizer(self, img):
        torch = _torch()
        img, dt = self._image(img)
        B = img.shape[0]
        tok = torch.empty((B, 128, self.E), dtype=torch.float32, device=img.device)
        _chk(lib().ita_tokenizer(self._h, img.data_ptr(), dt, tok.data_ptr(), B, _stream_ptr(self.device)))
        return tok

    def fusion_tail(self, x):
        torch = _torch()
        x = _dev_f32(x)
        feat = torch.empty((x.shape[0], 4608), dtype=torch.float32, device=x.device)
        _chk(lib().ita_fusion_tail(self._h, x.data_ptr(), feat.data_ptr(), x.shape[0], _stream_ptr(self.device)))
        return feat

    def _image(self, img):
        torch = _torch()
        if not img.is_cuda:
            raise ITAError("image must be a GPU tensor")
        if img.device.index != self.device:
            raise ITAError(f"tensor lives on cuda:{img.device.index}, the engine on cuda:{self.device}")
        if img.dtype == torch.uint8:
            if tuple(img.shape[-2:]) != (60, 90):   # wire frames are exactly 60 x 90 (ita_wire.h); no silent re-framing
                raise ITAError(f"u8 wire frames must be (..., 60, 90), got {tuple(img.shape)}")
            img = img.reshape(-1, 60, 90).contiguous()
            return img, IMAGE_U8
        img = img.float()
        if img.shape[-2:] != (60, 90):   # refine_inputs: bilinear resize, align_corners=False (QAT/model.py:29-30)
            img = torch.nn.functional.interpolate(img.reshape(-1, 1, *img.shape[-2:]), size=(60, 90),
                                                  mode="bilinear", align_corners=False)
        return img.reshape(-1, 60, 90).contiguous(), IMAGE_F32

    # ---- whole graph -------------------------------------------------------------------
    def forward(self, img, desvel, quat=None, hidden=None, taps: bool = False, out=None):
        """module.main_graph with a leading batch.  Returns (vel (B,3), (h, c) each (3,B,128))."""
        torch = _torch()
        img, dt = self._image(img)
        B, dev = img.shape[0], img.device
        desvel = _dev_f32(desvel).reshape(B)
        if quat is None:   # refine_inputs default quaternion [1,0,0,0] (QAT/model.py:23-27)
            quat = torch.zeros((B, 4), dtype=torch.float32, device=dev)
            quat[:, 0] = 1
        quat = _dev_f32(quat, (B, 4))
        if hidden is None:
            h_in = torch.zeros((3, B, 128), dtype=torch.float32, device=dev)
            c_in = torch.zeros((3, B, 128), dtype=torch.float32, device=dev)
        else:
            h_in, c_in = _dev_f32(hidden[0], (3, B, 128)), _dev_f32(hidden[1], (3, B, 128))
        if out is None:
            vel = torch.empty((B, 3), dtype=torch.float32, device=dev)
            h_out, c_out = torch.empty_like(h_in), torch.empty_like(c_in)
        else:
            vel, h_out, c_out = out
        tp, st = {}, None
        if taps:
            mk = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
            tp = dict(tokens=mk(B, 128, self.E), x1=mk(B, 128, self.E), x2=mk(B, 128, self.E))
            if getattr(self, "_tail_mode", 1) == 0:   # the folded tail (mode 1) never materialises feat / dec
                tp.update(feat=mk(B, 4608), dec=mk(B, 512))
            st = C.byref(_FwdTaps(**{k: tp[k].data_ptr() if k in tp else None for k in ("tokens", "x1", "x2", "feat", "dec")}))
        _chk(lib().ita_vitlstm_forward(self._h, img.data_ptr(), dt, desvel.data_ptr(), quat.data_ptr(),
                                       h_in.data_ptr(), c_in.data_ptr(), vel.data_ptr(), h_out.data_ptr(),
                                       c_out.data_ptr(), B, st, _stream_ptr(self.device)))
        if taps:
            return vel, (h_out, c_out), tp
        return vel, (h_out, c_out)

    def mha_q8(self, x_q, layer: int = 0):
        """attention block on int8 codes: x_q (B,128,E) int8 -> out_q (B,128,E) int8 (ita_mha_q8)"""
        torch = _torch()
        assert x_q.dtype == torch.int8 and x_q.is_cuda and tuple(x_q.shape[1:]) == (128, self.E)
        x_q = x_q.contiguous()
        out = torch.empty_like(x_q)
        _chk(lib().ita_mha_q8(self._h, layer, x_q.data_ptr(), out.data_ptr(), x_q.shape[0], _stream_ptr(self.device)))
        return out

    def mha_long_q8(self, x_q, layer: int = 0):
        """attention block on int8 codes over a long sequence: x_q (B,S,E) int8, S a multiple of 128 -> out_q (B,S,E) int8
        (ita_mha_long_q8; BASELINE config 5 as worded: S = 8192)"""
        torch = _torch()
        assert x_q.dtype == torch.int8 and x_q.is_cuda and x_q.dim() == 3 and x_q.shape[2] == self.E
        x_q = x_q.contiguous()
        out = torch.empty_like(x_q)
        _chk(lib().ita_mha_long_q8(self._h, layer, x_q.data_ptr(), out.data_ptr(), x_q.shape[0], x_q.shape[1],
                                   _stream_ptr(self.device)))
        return out

    def softmax_rows(self, logits):
        """IntegerApproximatedSoftmax through the encoder kernel's own device function: int8 (R,128) -> uint8 (R,128)"""
        torch = _torch()
        assert logits.dtype == torch.int8 and logits.is_cuda and logits.shape[-1] == 128
        lg = logits.reshape(-1, 128).contiguous()
        out = torch.empty(lg.shape, dtype=torch.uint8, device=lg.device)
        _chk(lib().ita_debug_softmax_rows(self._h, lg.data_ptr(), out.data_ptr(), lg.shape[0], _stream_ptr(self.device)))
        return out.reshape(logits.shape)

    def tail(self, x2, desvel, quat, hidden=None):
        """the graph behind the encoder: x2 (B,128,64) -> (vel, (h, c))   (fusion tail, decoder, LSTM, fc)"""
        torch = _torch()
        x2 = _dev_f32(x2)
        B, dev = x2.shape[0], x2.device
        desvel, quat = _dev_f32(desvel).reshape(B), _dev_f32(quat, (B, 4))
        if hidden is None:
            h_in = torch.zeros((3, B, 128), dtype=torch.float32, device=dev)
            c_in = torch.zeros((3, B, 128), dtype=torch.float32, device=dev)
        else:
            h_in, c_in = _dev_f32(hidden[0], (3, B, 128)), _dev_f32(hidden[1], (3, B, 128))
        vel = torch.empty((B, 3), dtype=torch.float32, device=dev)
        h_out, c_out = torch.empty_like(h_in), torch.empty_like(c_in)
        _chk(lib().ita_vitlstm_tail(self._h, x2.data_ptr(), desvel.data_ptr(), quat.data_ptr(), h_in.data_ptr(),
                                    c_in.data_ptr(), vel.data_ptr(), h_out.data_ptr(), c_out.data_ptr(), B,
                                    _stream_ptr(self.device)))
        return vel, (h_out, c_out)

    def encoder_stamps(self, x, layer: int = 0):
        """diagnostic: per-phase s_memtime stamps of the encoder stream kernel -> int64 [blocks, 8 frames, 2 waves
        (0 and 4), 16 slots].  x: (B,128,64) f32 tokens, or (B,60,90) u8 frames (tokenizer fused in front; slots 9, 10)"""
        torch = _torch()
        B = x.shape[0]
        if x.dtype == torch.uint8:
            img, xp = x.reshape(-1, 60, 90).contiguous(), None
        else:
            img, xp = None, _dev_f32(x)
        y = torch.empty((B, 128, self.E), dtype=torch.float32, device=x.device)
        nb = min(B, torch.cuda.get_device_properties(x.device).multi_processor_count)
        st = torch.zeros((nb, 8, 2, 16), dtype=torch.int64, device=x.device)
        _chk(lib().ita_debug_encoder_stamps(self._h, layer, xp.data_ptr() if xp is not None else None,
                                            img.data_ptr() if img is not None else None, y.data_ptr(), B, st.data_ptr(),
                                            _stream_ptr(self.device)))
        return st

    def set_tail_mode(self, mode: int):
        """1: folded conv+decoder, f16x3 split-precision MFMA tail (default); 0: exact f32 kernels"""
        _chk(lib().ita_set_tail_mode(self._h, mode))
        self._tail_mode = mode

    # ---- per-stage device timing ---------------------------------------------------------
    STAGES = ("tokenizer", "mha", "ffn", "tail", "decoder", "lstm_fc")

    def profile_begin(self, max_forwards: int, every_n: int = 1, only_stage: Optional[str] = None):
        """only_stage: one of STAGES -> record only that stage's two events, on every every_n-th forward"""
        st = -1 if only_stage is None else self.STAGES.index(only_stage)
        _chk(lib().ita_profile_begin_sampled(self._h, max_forwards, every_n, st))

    def profile_end(self):
        """-> ({stage: summed ms}, forwards covered)"""
        ms = (C.c_double * 6)()
        n = C.c_int()
        _chk(lib().ita_profile_end(self._h, ms, C.byref(n)))
        return dict(zip(self.STAGES, list(ms))), n.value

    def front(self, img, buf: int, stream=None, encoder_done=None):
        """image-only half of a time step (tokenizer, encoder, folded GEMM) into internal buffer `buf`;
        encoder_done: optional torch.cuda.Event (already recorded once, so that its handle exists) recorded between
        the encoder and the GEMM"""
        img, dt = self._image(img)
        sp = _stream_ptr(self.device) if stream is None else C.c_void_p(stream.cuda_stream)
        ev = None if encoder_done is None else C.c_void_p(encoder_done.cuda_event)
        _chk(lib().ita_vitlstm_front_ev(self._h, img.data_ptr(), dt, img.shape[0], buf, sp, ev))
        return img.shape[0]

    def encode(self, img, plane_set: int, stream=None):
        """tokenizer + encoder of a time step into plane set 0 / 1 (ita_vitlstm_encode)"""
        img, dt = self._image(img)
        sp = _stream_ptr(self.device) if stream is None else C.c_void_p(stream.cuda_stream)
        _chk(lib().ita_vitlstm_encode(self._h, img.data_ptr(), dt, img.shape[0], plane_set, sp))
        return img.shape[0]

    def fold(self, batch: int, plane_set: int, buf: int, stream=None):
        """folded GEMM of a time step: plane set -> partial buffer `buf` (ita_vitlstm_fold)"""
        sp = _stream_ptr(self.device) if stream is None else C.c_void_p(stream.cuda_stream)
        _chk(lib().ita_vitlstm_fold(self._h, batch, plane_set, buf, sp))

    def pipelined(self, imgs, desvels, quats, state, vels, stream_front, stream_back):
        """n time steps pipelined on two torch streams by the library (ita_vitlstm_pipelined): imgs / desvels / quats /
        vels are sequences of per-step tensors (or a tensor with a leading step axis), state = (h, c) updated in place"""
        torch = _torch()
        n = len(imgs)
        if n < 1 or not (len(desvels) == len(quats) == len(vels) == n):
            raise ITAError("pipelined: imgs / desvels / quats / vels must hold one tensor per step")
        checked = [self._image(im) for im in imgs]       # device ordinal, (60, 90), dtype, contiguity -- as forward does
        if len({dt for _, dt in checked}) != 1 or len({im.shape[0] for im, _ in checked}) != 1:
            raise ITAError("pipelined: every step must have the same batch and image dtype")
        imgs, dt = [im for im, _ in checked], checked[0][1]
        B = imgs[0].shape[0]

        def same(ts, shape, what):
            out = []
            for t in ts:
                if not t.is_cuda or t.device.index != self.device or t.dtype != torch.float32 or not t.is_contiguous() \
                        or t.numel() != int(np_prod(shape)):
                    raise ITAError(f"pipelined: every {what} must be a contiguous f32 tensor of {shape} on cuda:{self.device}")
                out.append(t)
            return out
        desvels, quats, vels = same(desvels, (B,), "desvel"), same(quats, (B, 4), "quat"), same(vels, (B, 3), "vel")
        same(state, (3, B, 128), "state tensor")
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        _chk(lib().ita_vitlstm_pipelined(self._h, arr(imgs), dt, arr(desvels), arr(quats), state[0].data_ptr(), state[1].data_ptr(),
                                         arr(vels), B, n, C.c_void_p(stream_front.cuda_stream), C.c_void_p(stream_back.cuda_stream)))

    def back(self, desvel, quat, hidden, out, buf: int, stream=None):
        """state half of a time step (LSTM + fc) from buffer `buf`; out = (vel, h_out, c_out) tensors"""
        B = out[0].shape[0]
        sp = _stream_ptr(self.device) if stream is None else C.c_void_p(stream.cuda_stream)
        _chk(lib().ita_vitlstm_back(self._h, desvel.data_ptr(), quat.data_ptr(), hidden[0].data_ptr(),
                                    hidden[1].data_ptr(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), B,
                                    buf, sp))

    def forward_slots(self, img, desvel, quat, state_h, state_c, slot_idx):
        """Serving form: state_h/state_c (3, num_slots, 128) persistent, updated in place for the
        streams named by slot_idx (int32 device tensor, distinct).  Returns vel (B,3)."""
        torch = _torch()
        img, dt = self._image(img)
        B = img.shape[0]
        desvel, quat = _dev_f32(desvel).reshape(B), _dev_f32(quat, (B, 4))
        assert state_h.is_contiguous() and state_c.is_contiguous() and state_h.dtype == torch.float32
        slot_idx = slot_idx.to(torch.int32).contiguous()
        vel = torch.empty((B, 3), dtype=torch.float32, device=img.device)
        _chk(lib().ita_vitlstm_forward_slots(self._h, img.data_ptr(), dt, desvel.data_ptr(), quat.data_ptr(),
                                             state_h.data_ptr(), state_c.data_ptr(), slot_idx.data_ptr(),
                                             state_h.shape[1], vel.data_ptr(), B, _stream_ptr(self.device)))
        return vel

    def graphed_step(self, batch: int) -> "GraphedStep":
        """a HIP-graph replay of one time step with static buffers (see GraphedStep)"""
        self.reserve(batch)
        g = GraphedStep(self, batch)
        self._track_graph(g)
        return g

    def pipelined_steps(self, batch: int, n_steps: int = 8, stages: int = 3) -> "PipelinedSteps":
        """n_steps time steps per HIP-graph replay: encoder(t+2) | folded GEMM(t+1) | LSTM + fc(t) on three streams
        (stages = 2: front(t+1) | back(t) on two); see PipelinedSteps"""
        if n_steps < 2 or stages not in (2, 3):
            raise ITAError("pipelined_steps: n_steps >= 2 and stages in (2, 3)")
        self.reserve(batch)
        g = PipelinedSteps(self, batch, n_steps, stages)
        self._track_graph(g)
        return g

    # ---- drop-in symbols (host buffers) --------------------------------------------------
    def bind_dispatch(self, layer: int = 0, dtype: int = DISPATCH_F16):
        _chk(lib().ita_bind_dispatch(self._h, layer, dtype))


class GraphedStep:
    """One time step of ita_vitlstm_forward captured in a HIP graph: at small batch the six kernel launches
    are launch-bound, a graph replays them with one host call.  Static buffers: write `img` (B,60,90) u8,
    `desvel` (B,), `quat` (B,4), then call the object; `vel` (B,3) holds the result, the LSTM state `h`, `c`
    (3,B,128) is updated in place from step to step (zero it to start a new stream)."""

    def __init__(self, engine: "Engine", batch: int):
        torch = _torch()
        dev = torch.device("cuda", engine.device)
        self.engine, self.B = engine, batch
        self.img = torch.zeros((batch, 60, 90), dtype=torch.uint8, device=dev)
        self.desvel = torch.zeros((batch,), dtype=torch.float32, device=dev)
        self.quat = torch.zeros((batch, 4), dtype=torch.float32, device=dev)
        self.quat[:, 0] = 1
        self.h = torch.zeros((3, batch, 128), dtype=torch.float32, device=dev)
        self.c = torch.zeros((3, batch, 128), dtype=torch.float32, device=dev)
        self.vel = torch.zeros((batch, 3), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):          # warm-up outside the capture: workspace, kernel attributes
            for _ in range(2):
                self._step()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.h.zero_()
        self.c.zero_()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._step()
        self.h.zero_()
        self.c.zero_()

    def _step(self):
        self.engine.forward(self.img, self.desvel, self.quat, (self.h, self.c), out=(self.vel, self.h, self.c))

    def __call__(self):
        self.graph.replay()
        return self.vel


PART_BUFFERS = 8    # ITA_PART_BUFFERS of include/ita_mi355x.h: partial-sum buffers of the front/back form


class PipelinedSteps:
    """`n_steps` consecutive time steps captured in ONE HIP graph.  stages = 3 (default), three streams: the encoder of step
    t+2 (ita_vitlstm_encode), the folded GEMM of step t+1 (ita_vitlstm_fold) and the LSTM layers + fc of step t
    (ita_vitlstm_back) run side by side, with two sets of encoder-output planes and three partial-sum buffers; measured per
    step (tools/pipeline3_probe.py): 1 frame 31.7 us, 64 frames 36.1, 128 frames 38.2, 256 frames 47.7 -- against 34.0 / 39.7 /
    43.0 / 47.6 for stages = 2, which is the rest of this description.
    stages = 2, two streams: the image-only front of step t+1
    (tokenizer, encoder, folded GEMM: ita_vitlstm_front) runs while the recurrent back of step t (LSTM layers, fc:
    ita_vitlstm_back) is still in flight.  The recurrence is respected -- back(t) follows back(t-1) on its stream and
    front(t) -- and so is the reuse of the partial-sum buffers (front(t) waits for back(t - PART_BUFFERS); with n_steps <=
    PART_BUFFERS the front stream never waits for the back stream inside a replay).  This pays when a GPU
    holds few streams (up to ~256: the strong-scaling regime of BASELINE config 4 on 8 GPUs): the encoder then leaves CUs
    free for the small LSTM kernels, and one graph launch replaces 6 * n_steps kernel launches.  At 1024 streams per GPU
    the encoder owns every CU and nothing overlaps (measured), so bench.py uses this below 256 streams only.
    Static buffers: `img` (n,B,60,90) u8, `desvel` (n,B), `quat` (n,B,4) in; `vel` (n,B,3) out; `h`, `c` (3,B,128) carried
    in place from step to step and from replay to replay (zero them to start new streams)."""

    def __init__(self, engine: "Engine", batch: int, n_steps: int = 8, stages: int = 3):
        torch = _torch()
        dev = torch.device("cuda", engine.device)
        self.engine, self.B, self.n, self.stages = engine, batch, n_steps, stages
        if n_steps < 2 or stages not in (2, 3):
            raise ITAError("PipelinedSteps: n_steps >= 2 and stages in (2, 3)")
        engine.reserve(batch)
        self.img = torch.zeros((n_steps, batch, 60, 90), dtype=torch.uint8, device=dev)
        self.desvel = torch.zeros((n_steps, batch), dtype=torch.float32, device=dev)
        self.quat = torch.zeros((n_steps, batch, 4), dtype=torch.float32, device=dev)
        self.quat[..., 0] = 1
        self.h = torch.zeros((3, batch, 128), dtype=torch.float32, device=dev)
        self.c = torch.zeros((3, batch, 128), dtype=torch.float32, device=dev)
        self.vel = torch.zeros((n_steps, batch, 3), dtype=torch.float32, device=dev)
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(warm):              # outside the capture: kernel attributes, lazy module loads
            for i in range(2):
                engine.front(self.img[i], i & 1, stream=warm)
                engine.back(self.desvel[i], self.quat[i], (self.h, self.c), (self.vel[i], self.h, self.c), i & 1, stream=warm)
        torch.cuda.current_stream(dev).wait_stream(warm)
        torch.cuda.synchronize(dev)
        self.h.zero_()
        self.c.zero_()
        self.graph = torch.cuda.CUDAGraph()
        s_front, s_back = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        ev_front = [torch.cuda.Event() for _ in range(n_steps)]
        ev_back = [torch.cuda.Event() for _ in range(n_steps)]
        # Two partial buffers, although the library has PART_BUFFERS: with front(i) waiting for back(i-2) the HIP runtime
        # overlaps the two branches of the replayed graph (41.9 us per 128-frame step for 32.6 us of front and 17.3 us of
        # back); without that edge (8 buffers) it runs them one after the other (52.5 us; tools/pipeline_probe.py).
        nb = 2
        if stages == 3:
            # encoder(i+2) | folded GEMM(i+1) | LSTM + fc(i): three branches, two plane sets, three partial buffers
            s_fold = torch.cuda.Stream(device=dev)
            ev_enc = [torch.cuda.Event() for _ in range(n_steps)]
            ev_planes = [torch.cuda.Event() for _ in range(n_steps)]   # (one event per waiter)
            ev_fork = torch.cuda.Event()
            with torch.cuda.graph(self.graph, stream=s_front):
                # both side streams fork from the ORIGIN stream (a stream that joins the capture through an event of
                # another forked stream crashes hipStreamEndCapture on ROCm 7.0)
                ev_fork.record(s_front)
                s_fold.wait_event(ev_fork)
                s_back.wait_event(ev_fork)
                def cap_encode(i):
                    if i >= 2:
                        s_front.wait_event(ev_planes[i - 2])       # encode(i) overwrites the planes fold(i-2) read
                    if i >= 3:
                        # fold(i) overwrites the partials back(i-3) read (three partial buffers).  The edge goes through
                        # the origin stream -- fold(i) follows encode(i) anyway, and back(i-3) is two steps behind the
                        # encoder in steady state -- because a forked stream that waits for an event downstream of its
                        # own earlier work crashes hipStreamEndCapture on ROCm 7.0
                        s_front.wait_event(ev_back[i - 3])
                    engine.encode(self.img[i], i & 1, stream=s_front)
                    ev_enc[i].record(s_front)

                def cap_fold_back(i):
                    s_fold.wait_event(ev_enc[i])
                    engine.fold(batch, i & 1, i % 3, stream=s_fold)
                    ev_front[i].record(s_fold)
                    ev_planes[i].record(s_fold)
                    s_back.wait_event(ev_front[i])
                    engine.back(self.desvel[i], self.quat[i], (self.h, self.c), (self.vel[i], self.h, self.c), i % 3, stream=s_back)
                    ev_back[i].record(s_back)

                # Node order of the capture (the dependencies are the same; ROCm 7.0's graph executor is sensitive to it -- tools/
                # bench_small.py with ITA_GRAPH_ORDER).  Default: step by step, encode(i) fold(i) back(i).  "1": encode(i + 1) before
                # fold(i) back(i).  A permutation of "EFB": level by level -- level s holds encode(s), fold(s - 1), back(s - 2), which
                # do not depend on each other -- in that order inside a level.
                order = os.environ.get("ITA_GRAPH_ORDER", "0")
                if order == "1":
                    cap_encode(0)
                    for i in range(n_steps):
                        if i + 1 < n_steps:
                            cap_encode(i + 1)                       # (its waits -- fold(i - 1), back(i - 2) -- are recorded already)
                        cap_fold_back(i)
                elif sorted(order) == ["B", "E", "F"]:
                    def cap_fold(i):
                        s_fold.wait_event(ev_enc[i])
                        engine.fold(batch, i & 1, i % 3, stream=s_fold)
                        ev_front[i].record(s_fold)
                        ev_planes[i].record(s_fold)

                    def cap_back(i):
                        s_back.wait_event(ev_front[i])
                        engine.back(self.desvel[i], self.quat[i], (self.h, self.c), (self.vel[i], self.h, self.c), i % 3, stream=s_back)
                        ev_back[i].record(s_back)

                    for lvl in range(n_steps + 2):
                        for ch in order:
                            i = lvl - "EFB".index(ch)
                            if 0 <= i < n_steps:
                                {"E": cap_encode, "F": cap_fold, "B": cap_back}[ch](i)
                else:
                    for i in range(n_steps):
                        cap_encode(i)
                        cap_fold_back(i)
                s_front.wait_stream(s_fold)                         # join
                s_front.wait_stream(s_back)
            self.h.zero_()
            self.c.zero_()
            return
        with torch.cuda.graph(self.graph, stream=s_front):
            for i in range(n_steps):
                if i >= nb:
                    s_front.wait_event(ev_back[i - nb])         # front(i) overwrites the partial buffer back(i-nb) read
                engine.front(self.img[i], i % nb, stream=s_front)
                ev_front[i].record(s_front)
                s_back.wait_event(ev_front[i])                  # (also forks s_back into the capture at i = 0)
                engine.back(self.desvel[i], self.quat[i], (self.h, self.c), (self.vel[i], self.h, self.c), i % nb, stream=s_back)
                ev_back[i].record(s_back)
            s_front.wait_stream(s_back)                          # join
        self.h.zero_()
        self.c.zero_()

    def __call__(self):
        self.graph.replay()
        return self.vel


class FusionTailLarge:
    """The fusion tail (PixelShuffle(2) || Upsample x2 align_corners -> cat -> Conv2d 3x3) on an arbitrary
    token grid -- BASELINE config 5 (reference models/ITA_single_layer_upsample_shuffle/QAT/model.py:116-121,
    layer sizes of models/ITA_upsample_shuffle/model.py:70-79).  conv_w (CO, 5E/4, 3, 3), conv_b (CO,): numpy."""

    def __init__(self, conv_w, conv_b, device: Optional[int] = None):
        import numpy as np
        torch = _torch()
        if not torch.cuda.is_available():
            raise ITAError("no GPU visible: the ITA engine has no CPU path")
        self.device = torch.cuda.current_device() if device is None else int(device)
        w = np.ascontiguousarray(conv_w, dtype=np.float32)
        b = np.ascontiguousarray(conv_b, dtype=np.float32)
        self.CO, cin = w.shape[0], w.shape[1]
        self.E = cin * 4 // 5
        if w.shape[2:] != (3, 3) or self.E // 4 + self.E != cin or b.shape != (self.CO,):
            raise ITAError("conv_w must be (CO, 5E/4, 3, 3) and conv_b (CO,)")
        self._h = C.c_void_p()
        _chk(lib().ita_create(C.byref(self._h), self.device))
        _chk(lib().ita_fusion_tail_load(self._h, w.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), self.E, self.CO))

    def __call__(self, x, tok_h: int, tok_w: int, out=None):
        """x (B, tok_h*tok_w, E) f32 on the GPU -> (B, CO, 2 tok_h, 2 tok_w) f32"""
        torch = _torch()
        x = _dev_f32(x)
        B = x.shape[0]
        if x.shape[1] != tok_h * tok_w or x.shape[2] != self.E:
            raise ITAError(f"x must be (B, {tok_h * tok_w}, {self.E})")
        if out is None:
            out = torch.empty((B, self.CO, 2 * tok_h, 2 * tok_w), dtype=torch.float32, device=x.device)
        _chk(lib().ita_fusion_tail_large(self._h, x.data_ptr(), out.data_ptr(), B, tok_h, tok_w, _stream_ptr(self.device)))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().ita_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ITAViTLSTM:
    """Mirror of the reference's ``ITALSTMNetVIT_QAT`` call convention (QAT/model.py:93-132)."""

    def __init__(self, blob: bytes, device: Optional[int] = None, reserve: int = 0):
        self.engine = Engine(blob, device, reserve)

    def __call__(self, X: Sequence):
        return self.forward(X)

    def forward(self, X: Sequence):
        img, desvel = X[0], X[1]
        quat = X[2] if len(X) > 2 else None
        hidden = X[3] if len(X) > 3 else None
        return self.engine.forward(img, desvel, quat, hidden)
