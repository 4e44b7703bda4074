"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI,
against the CPU oracle on the same inputs, and against the committed golden fixtures captured
from the reference.

Bars: integer / byte tensors bit-exact.  Float stages share the oracle's operation order and are
compared for equality too, except where the hardware's summation order is not architecturally
documented (f32 MFMA inside the decoder / LSTM GEMMs): tolerance 1e-5 absolute there, stated
at the assertion.
"""
import ctypes
import glob
import os

import numpy as np
import pytest

from conftest import golden_files
from drone_oa_iree_vit_accelerator_amd import host, params, synth

pytestmark = pytest.mark.gpu

FIX_ALL = golden_files("*_seed*.npz")
FIX_VIT = golden_files("vitlstm_*.npz")


def _ids(paths):
    return [p.split("/")[-1][:-4] for p in paths]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _engine(d, E, with_float=True, seed=None):
    fp = synth.float_params(int(d["meta.seed"]) if seed is None else seed, E=E) if with_float else None
    blob = params.blob_from_record(d, fp, E=E)
    return host.Engine(blob, device=0), blob, fp


def _block_tensors(d):
    t = params.attention_tensors(d, "attn0.", 0)
    t.update(params.ffn_tensors(d, "ffn0.", 0))
    return t


@pytest.mark.parametrize("path", FIX_ALL, ids=_ids(FIX_ALL))
def test_mha_block_bit_exact(torch_cuda, oracle, path):
    torch = torch_cuda
    d = params.load_fixture(path)
    E = int(d["meta.E"])
    eng, _, _ = _engine(d, E, with_float=(E == 64 and "in0.img_u8" in d))
    x = d["s0.attn0.x_q.in"]
    y, taps = eng.mha(torch.from_numpy(x).cuda(), taps=True)
    torch.cuda.synchronize()
    oy, otaps = oracle.mha(x, _block_tensors(d), taps=True)
    for k in ("x_q", "Q", "K", "V", "logits", "probs", "ctx", "out_q"):
        np.testing.assert_array_equal(taps[k].cpu().numpy(), otaps[k], err_msg=k)
    np.testing.assert_array_equal(y.cpu().numpy(), oy)
    # and against the reference's own tensors (fixture): everything but near-tie logit rows
    np.testing.assert_array_equal(taps["Q"].cpu().numpy(), d["s0.attn0.Q"])
    np.testing.assert_array_equal(taps["V"].cpu().numpy(), d["s0.attn0.V"])
    ref_l = d["s0.attn0.probs.in"][:, 0]
    gl = taps["logits"].cpu().numpy()
    assert (gl != ref_l).sum() <= 3
    rows_ok = (gl == ref_l).all(axis=-1)
    np.testing.assert_array_equal(taps["probs"].cpu().numpy()[rows_ok], d["s0.attn0.probs"][:, 0][rows_ok])
    np.testing.assert_array_equal(taps["out_q"].cpu().numpy()[rows_ok], d["s0.attn0.out_q"][rows_ok])
    np.testing.assert_array_equal(y.cpu().numpy()[rows_ok], d["s0.attn0.out_f"][rows_ok])
    # the tap-less entry runs a different kernel (ita_stream_kernel: weights resident in LDS, activations chained
    # through registers): same bits; and the int8-in / int8-out form of it (ita_mha_q8) maps x_q to out_q
    np.testing.assert_array_equal(eng.mha(torch.from_numpy(x).cuda()).cpu().numpy(), oy)
    np.testing.assert_array_equal(eng.mha_q8(taps["x_q"]).cpu().numpy(), otaps["out_q"])
    rs = np.random.RandomState(3)
    xq = rs.randint(-128, 128, size=(259, 128, E)).astype(np.int8)                 # beyond one frame per workgroup
    got = eng.mha_q8(torch.from_numpy(xq).cuda()).cpu().numpy()
    sel = [0, 1, 255, 256, 258]
    np.testing.assert_array_equal(got[sel], oracle.mha_q8(xq[sel], _block_tensors(d)))   # the oracle's int8-in entry: unconditional
    eng.close()


@pytest.mark.parametrize("path", FIX_ALL, ids=_ids(FIX_ALL))
def test_ffn_block_bit_exact(torch_cuda, oracle, path):
    torch = torch_cuda
    d = params.load_fixture(path)
    E = int(d["meta.E"])
    eng, _, _ = _engine(d, E, with_float=False)
    x = d["s0.ffn0.x_q.in"]
    y, taps = eng.ffn(torch.from_numpy(x).cuda(), taps=True)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(taps["x_q"].cpu().numpy(), d["s0.ffn0.x_q"])
    np.testing.assert_array_equal(taps["h"].cpu().numpy(), d["s0.ffn0.h1_relu"])
    np.testing.assert_array_equal(taps["out_q"].cpu().numpy(), d["s0.ffn0.out_q"])
    np.testing.assert_array_equal(y.cpu().numpy(), d["s0.ffn0.out_f"])
    eng.close()


@pytest.mark.parametrize("E", [64, 128])
def test_blocks_random_batch_and_saturation(torch_cuda, oracle, E):
    """ragged batch (not a multiple of the grid), saturating / zero / constant rows"""
    torch = torch_cuda
    d = params.load_fixture(golden_files(f"blocks_E{E}_seed*.npz")[0])
    eng, _, _ = _engine(d, E, with_float=False)
    rs = np.random.RandomState(5)
    B = 37
    x = rs.standard_normal((B, 128, E)).astype(np.float32)
    x[0] = 0.0
    x[1] = 50.0                      # every code saturates at +127
    x[2] = -50.0
    x[3, :, ::2] *= 30.0
    x[4, 5] = 1e30
    x[5] = np.float32(0.5) * np.float32(float(d["attn0.quant.scale"]))   # exact rounding ties (x/s = 0.5)
    x[6] = np.float32(1.5) * np.float32(float(d["attn0.quant.scale"]))
    t = _block_tensors(d)
    y, taps = eng.mha(torch.from_numpy(x).cuda(), taps=True)
    oy, otaps = oracle.mha(x, t, taps=True)
    for k in otaps:
        np.testing.assert_array_equal(taps[k].cpu().numpy(), otaps[k], err_msg=k)
    np.testing.assert_array_equal(y.cpu().numpy(), oy)
    y2, taps2 = eng.ffn(torch.from_numpy(x).cuda(), taps=True)
    oy2, otaps2 = oracle.ffn(x, t, taps=True)
    for k in otaps2:
        np.testing.assert_array_equal(taps2[k].cpu().numpy(), otaps2[k], err_msg=k)
    np.testing.assert_array_equal(y2.cpu().numpy(), oy2)
    eng.close()


def test_mha_more_frames_than_workgroups(torch_cuda, oracle):
    """grid-stride path: B larger than the number of CUs"""
    torch = torch_cuda
    d = params.load_fixture(golden_files("blocks_E64_seed2_B1.npz")[0])
    eng, _, _ = _engine(d, 64, with_float=False)
    rs = np.random.RandomState(9)
    B = 2 * 256 + 19
    x = (1.5 * rs.standard_normal((B, 128, 64))).astype(np.float32)
    y = eng.mha(torch.from_numpy(x).cuda()).cpu().numpy()
    oy = oracle.mha(x, _block_tensors(d))
    np.testing.assert_array_equal(y, oy)
    eng.close()


@pytest.mark.parametrize("path", FIX_VIT, ids=_ids(FIX_VIT))
def test_float_stages_equal_oracle(torch_cuda, oracle, path):
    torch = torch_cuda
    d = params.load_fixture(path)
    eng, blob, fp = _engine(d, 64)
    for key, dtype in (("u8", None), ("f32", np.float32)):
        img = d["in0.img_u8"] if dtype is None else d["in0.img_u8"].astype(np.float32) / np.float32(255.0)
        tok = eng.tokenizer(torch.from_numpy(img).cuda()).cpu().numpy()
        # u8 wire frames: exact integer blend (ita_oracle_tokenizer_u8); f32 frames: the float blend
        otok = oracle.tokenizer(img, fp["tokenizer.conv.weight"].reshape(64, 49), fp["tokenizer.conv.bias"],
                                fp["tokenizer.norm.weight"], fp["tokenizer.norm.bias"])
        np.testing.assert_array_equal(tok, otok, err_msg=key)
        np.testing.assert_allclose(tok, d["s0.tok.out"], atol=2e-5, rtol=0)
    x2 = d["s0.x2"]
    feat = eng.fusion_tail(torch.from_numpy(x2).cuda()).cpu().numpy()
    ofeat = oracle.tail(x2, fp["down_sample.weight"], fp["down_sample.bias"])
    np.testing.assert_array_equal(feat, ofeat)
    np.testing.assert_allclose(feat.reshape(-1, 9, 16, 32), d["s0.tail.conv"], atol=2e-5, rtol=0)
    # one encoder layer with the fused residual + LayerNorm epilogues
    x = d["s0.tok.out"]
    y = eng.encoder_layer(torch.from_numpy(x).cuda()).cpu().numpy()
    t = _block_tensors(d)
    x1 = oracle.add_ln(x, oracle.mha(x, t), fp["norms1.0.weight"], fp["norms1.0.bias"])
    ox2 = oracle.add_ln(x1, oracle.ffn(x1, t), fp["norms2.0.weight"], fp["norms2.0.bias"])
    np.testing.assert_array_equal(y, ox2)
    eng.close()


@pytest.mark.parametrize("mode", [1, 0], ids=["tail_f16x3", "tail_exact_f32"])
@pytest.mark.parametrize("path", FIX_VIT, ids=_ids(FIX_VIT))
def test_full_forward_two_steps(torch_cuda, oracle, path, mode):
    """mode 1 (default): conv+decoder folded, tail GEMMs on split-precision f16 MFMA; tolerance
    2e-5 absolute against the oracle (task tolerance for the tail: 1e-4).  mode 0: exact f32
    kernels, equality.  Everything up to and including x2 is equal to the oracle in both modes."""
    torch = torch_cuda
    d = params.load_fixture(path)
    eng, blob, fp = _engine(d, 64)
    eng.set_tail_mode(mode)
    tol = 2e-5 if mode == 1 else 0.0
    cu = lambda a: torch.from_numpy(a).cuda()
    vel0, (h0, c0), tp = eng.forward(cu(d["in0.img_u8"]), cu(d["in0.desvel"]), cu(d["in0.quat"]), taps=True)
    ovel0, oh0, oc0, otp = oracle.forward(blob, d["in0.img_u8"], d["in0.desvel"], d["in0.quat"], taps=True)
    for k in ("tokens", "x1", "x2") + (("feat",) if mode == 0 else ()):
        np.testing.assert_array_equal(tp[k].cpu().numpy(), otp[k], err_msg=k)
    if mode == 0:     # in mode 1 the decoder is folded into LSTM layer 0: `dec` is never materialised
        np.testing.assert_array_equal(tp["dec"].cpu().numpy(), otp["dec"])
    np.testing.assert_allclose(vel0.cpu().numpy(), ovel0, atol=tol, rtol=0)
    np.testing.assert_allclose(h0.cpu().numpy(), oh0, atol=tol, rtol=0)
    np.testing.assert_allclose(c0.cpu().numpy(), oc0, atol=tol, rtol=0)
    print(f"\n[tail mode {mode}] max|vel - oracle| = {np.abs(vel0.cpu().numpy() - ovel0).max():.3e}, "
          f"max|h - oracle| = {np.abs(h0.cpu().numpy() - oh0).max():.3e}, "
          f"max|c - oracle| = {np.abs(c0.cpu().numpy() - oc0).max():.3e}")
    # second time step, state carried on the device like the reference host carries it
    vel1, (h1, c1) = eng.forward(cu(d["in1.img_u8"]), cu(d["in1.desvel"]), cu(d["in1.quat"]), (h0, c0))
    ovel1, oh1, oc1 = oracle.forward(blob, d["in1.img_u8"], d["in1.desvel"], d["in1.quat"], oh0, oc0)
    np.testing.assert_allclose(vel1.cpu().numpy(), ovel1, atol=2 * tol, rtol=0)
    np.testing.assert_allclose(h1.cpu().numpy(), oh1, atol=2 * tol, rtol=0)
    # against the reference's own outputs (fixture): int8 flips behind float LayerNorms allowed
    np.testing.assert_allclose(vel0.cpu().numpy(), d["s0.vel"], atol=5e-4, rtol=0)
    np.testing.assert_allclose(vel1.cpu().numpy(), d["s1.vel"], atol=5e-4, rtol=0)
    np.testing.assert_allclose(c1.cpu().numpy(), d["s1.c"], atol=1e-3, rtol=0)   # (cell state: tests/test_oracle_golden.py, same bound)
    # reference call convention: default quaternion, no hidden state
    model = host.ITAViTLSTM(blob, device=0)
    v, (hh, cc) = model([cu(d["in0.img_u8"]), cu(d["in0.desvel"])])
    q = np.zeros((2, 4), np.float32); q[:, 0] = 1
    ov, _, _ = oracle.forward(blob, d["in0.img_u8"], d["in0.desvel"], q)
    np.testing.assert_allclose(v.cpu().numpy(), ov, atol=2e-5, rtol=0)
    eng.close()


def test_decoder_lstm_exactness_report(torch_cuda, oracle):
    """Not a gate: records whether the f32-MFMA GEMMs reproduce the oracle's fmaf chain bit for bit."""
    torch = torch_cuda
    d = params.load_fixture(FIX_VIT[0])
    eng, blob, fp = _engine(d, 64)
    eng.set_tail_mode(0)
    cu = lambda a: torch.from_numpy(a).cuda()
    vel, (h, c), tp = eng.forward(cu(d["in0.img_u8"]), cu(d["in0.desvel"]), cu(d["in0.quat"]), taps=True)
    ovel, oh, oc, otp = oracle.forward(blob, d["in0.img_u8"], d["in0.desvel"], d["in0.quat"], taps=True)
    nd = int((tp["dec"].cpu().numpy() != otp["dec"]).sum())
    nh = int((h.cpu().numpy() != oh).sum())
    nv = int((vel.cpu().numpy() != ovel).sum())
    print(f"\n[exactness] decoder mismatches {nd}/{otp['dec'].size}, h {nh}/{oh.size}, vel {nv}/{ovel.size}; "
          f"max |dec diff| {np.abs(tp['dec'].cpu().numpy() - otp['dec']).max():.3e}")
    eng.close()


def test_full_size_properties(torch_cuda, oracle):
    """BASELINE config 4 size (B = 1024): determinism, batch-composition independence, and a
    sampled oracle check (the oracle needs ~25 ms per frame)."""
    torch = torch_cuda
    d = params.load_fixture(FIX_VIT[0])
    eng, blob, fp = _engine(d, 64)
    B = 1024
    fr = synth.frames(77, B)
    cu = lambda a: torch.from_numpy(a).cuda()
    img, dv, qt = cu(fr["img_u8"]), cu(fr["desvel"]), cu(fr["quat"])
    h = torch.from_numpy(np.random.RandomState(3).standard_normal((3, B, 128)).astype(np.float32) * 0.1).cuda()
    c = torch.from_numpy(np.random.RandomState(4).standard_normal((3, B, 128)).astype(np.float32) * 0.1).cuda()
    v1, (h1, c1) = eng.forward(img, dv, qt, (h, c))
    v2, (h2, c2) = eng.forward(img, dv, qt, (h, c))
    assert torch.equal(v1, v2) and torch.equal(h1, h2) and torch.equal(c1, c2)
    # a frame's result does not depend on what else is in the batch or where it sits
    idx = torch.tensor([1023, 0, 511, 300, 7], device="cuda")
    vs, (hs, cs) = eng.forward(img[idx], dv[idx], qt[idx], (h[:, idx].contiguous(), c[:, idx].contiguous()))
    assert torch.equal(vs, v1[idx]) and torch.equal(hs, h1[:, idx]) and torch.equal(cs, c1[:, idx])
    # ... nor on which of the three folded-GEMM kernels serves the batch: one M tile (<= 32 frames, one wave per tile and
    # K slice), two to four M tiles per workgroup (33..128), several such workgroups with a ragged last one (129..256),
    # the 128 x 128-tile kernel (> 256); LSTM frame tiles of 32 with a ragged tail everywhere
    for n in (32, 33, 65, 100, 128, 129, 200, 256, 257, 300):
        sub = torch.arange(1023, 1023 - n, -1, device="cuda")
        vn, (hn, cn) = eng.forward(img[sub], dv[sub], qt[sub], (h[:, sub].contiguous(), c[:, sub].contiguous()))
        assert torch.equal(vn, v1[sub]) and torch.equal(hn, h1[:, sub]) and torch.equal(cn, c1[:, sub]), n
    sel = [0, 255, 256, 777, 1023]
    ov, oh, oc = oracle.forward(blob, fr["img_u8"][sel], fr["desvel"][sel], fr["quat"][sel],
                                h.cpu().numpy()[:, sel], c.cpu().numpy()[:, sel])
    np.testing.assert_allclose(v1.cpu().numpy()[sel], ov, atol=2e-5, rtol=0)
    np.testing.assert_allclose(c1.cpu().numpy()[:, sel], oc, atol=2e-5, rtol=0)
    eng.close()


@pytest.mark.parametrize("B", [1, 3, 257, 300])
def test_fused_tokenizer_path_equals_split_path(torch_cuda, oracle, B):
    """u8 wire frames take the kernel with the tokenizer fused in front of the encoder layer; the stand-alone tokenizer
    launch followed by the encoder-layer entry point is the split form of the same computation.  Both must give the same
    bits, for batch sizes below, at and above the number of workgroups (256 CUs: frames 256.. are a workgroup's second
    frame).  f32 frames (float(pixel) / 255.0f done by the caller, main.cpp:168-169) go through the float blend: equal
    to the oracle's float entry point, and within 4e-6 of the u8 tokens (the two blends round differently)."""
    torch = torch_cuda
    d = params.load_fixture(FIX_VIT[0])
    eng, blob, fp = _engine(d, 64)
    fr = synth.frames(900 + B, B)
    cu = lambda a: torch.from_numpy(a).cuda()
    img8 = cu(fr["img_u8"])
    img_f = fr["img_u8"].astype(np.float32) / np.float32(255.0)   # IEEE division, done in numpy
    dv, qt = cu(fr["desvel"]), cu(fr["quat"])
    va, (ha, ca), ta = eng.forward(img8, dv, qt, taps=True)
    tok_split = eng.tokenizer(img8)
    assert torch.equal(ta["tokens"], tok_split)
    assert torch.equal(ta["x2"], eng.encoder_layer(tok_split))
    sel = sorted({0, B - 1, B // 2})
    otok = oracle.tokenizer(fr["img_u8"][sel], fp["tokenizer.conv.weight"].reshape(64, 49), fp["tokenizer.conv.bias"],
                            fp["tokenizer.norm.weight"], fp["tokenizer.norm.bias"])
    np.testing.assert_array_equal(ta["tokens"].cpu().numpy()[sel], otok)
    vb, (hb, cb), tb = eng.forward(cu(img_f), dv, qt, taps=True)
    ovel, oh, oc, otp = oracle.forward(blob, img_f[sel], fr["desvel"][sel], fr["quat"][sel], taps=True)
    for k in ("tokens", "x1", "x2"):
        np.testing.assert_array_equal(tb[k].cpu().numpy()[sel], otp[k], err_msg=k)
    np.testing.assert_allclose(tb["tokens"].cpu().numpy(), ta["tokens"].cpu().numpy(), atol=4e-6, rtol=0)
    # f32 frames need not be k/255: arbitrary floats (a caller that normalises depth differently, negative values,
    # values far above 1) take the same blend, conv and LayerNorm expressions as the oracle, bit for bit
    rs = np.random.RandomState(4000 + B)
    n = min(B, 5)
    wild = (rs.standard_normal((n, 60, 90)) * np.exp(rs.uniform(-3, 3, (n, 1, 1)))).astype(np.float32)
    otw = oracle.tokenizer(wild, fp["tokenizer.conv.weight"].reshape(64, 49), fp["tokenizer.conv.bias"],
                           fp["tokenizer.norm.weight"], fp["tokenizer.norm.bias"])
    np.testing.assert_array_equal(eng.tokenizer(cu(wild)).cpu().numpy(), otw)
    eng.close()


def test_refine_inputs_resize_and_default_quaternion(torch_cuda, oracle):
    """refine_inputs (QAT/model.py:22-31): frames that are not 60 x 90 are resized bilinearly (align_corners=False) and a
    missing quaternion is [1,0,0,0].  The resize runs through torch on the GPU; against the same resize done by torch
    on the CPU (what the reference executes) the tokens agree to 1e-5 and the velocities to the tolerance the int8
    path allows behind a float difference of that size (5e-4, as against the reference fixtures)."""
    torch = torch_cuda
    d = params.load_fixture(FIX_VIT[0])
    eng, blob, fp = _engine(d, 64)
    rs = np.random.RandomState(21)
    big = rs.uniform(0, 1, size=(3, 1, 120, 180)).astype(np.float32)
    dv = rs.uniform(2, 8, size=(3, 1)).astype(np.float32)
    small = torch.nn.functional.interpolate(torch.from_numpy(big), size=(60, 90), mode="bilinear", align_corners=False)
    model = host.ITAViTLSTM(blob, device=0)
    v_big, _ = model([torch.from_numpy(big).cuda(), torch.from_numpy(dv).cuda()])          # no quaternion, no state
    q = np.zeros((3, 4), np.float32); q[:, 0] = 1
    ov, _, _, otp = oracle.forward(blob, small.numpy().reshape(3, 60, 90), dv, q, taps=True)
    _, _, tpb = eng.forward(torch.from_numpy(big).cuda(), torch.from_numpy(dv).cuda(), taps=True)
    np.testing.assert_allclose(tpb["tokens"].cpu().numpy(), otp["tokens"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(v_big.cpu().numpy(), ov, atol=5e-4, rtol=0)
    eng.close()


@pytest.mark.parametrize("dtype", ["f16", "f32"])
def test_dropin_symbol_host_buffers(torch_cuda, oracle, dtype):
    """ITASelfAttention_workgroup(in, out) with the reference's prototype: host buffers of
    1 x 128 x 128 elements (ITA_spec.mlir:30-33), full attention block instead of a copy."""
    d = params.load_fixture(golden_files("blocks_E128_seed0_B1.npz")[0])
    eng, _, _ = _engine(d, 128, with_float=False)
    x = d["s0.attn0.x_q.in"][0]
    lib = host.lib()
    if dtype == "f16":
        eng.bind_dispatch(0, host.DISPATCH_F16)
        xin = x.astype(np.float16)
        out = np.empty_like(xin)
        want = oracle.mha(xin.astype(np.float32)[None], _block_tensors(d))[0].astype(np.float16)
    else:
        eng.bind_dispatch(0, host.DISPATCH_F32)
        xin = x.copy()
        out = np.empty_like(xin)
        want = oracle.mha(xin[None], _block_tensors(d))[0]
    lib.ITASelfAttention_workgroup(xin.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    assert lib.ita_last_error() == 0, lib.ita_error_string()
    np.testing.assert_array_equal(out, want)
    out2 = np.empty_like(xin)
    lib.ITASelfAttention_workgroup_expanded(None, xin.ctypes.data_as(ctypes.c_void_p), 0, xin.size, 1, None,
                                            out2.ctypes.data_as(ctypes.c_void_p), 0, out2.size, 1)
    np.testing.assert_array_equal(out2, want)
    fo = np.empty_like(xin)
    lib.ITAFeedForward_workgroup(xin.ctypes.data_as(ctypes.c_void_p), fo.ctypes.data_as(ctypes.c_void_p))
    wf = oracle.ffn(xin.astype(np.float32)[None], _block_tensors(d))[0].astype(xin.dtype)
    np.testing.assert_array_equal(fo, wf)
    eng.close()


def test_error_codes_on_gpu(torch_cuda):
    lib = host.lib()
    h = ctypes.c_void_p()
    assert lib.ita_create(ctypes.byref(h), 0) == 0
    x = torch_cuda.zeros((1, 128, 64), device="cuda")
    assert lib.ita_mha_int8(h, 0, x.data_ptr(), x.data_ptr(), 1, None) == -3          # no weights
    bad = ctypes.create_string_buffer(b"NOTABLOB" + b"\0" * 128)
    assert lib.ita_load_weights(h, bad, 136) == -2
    d = params.load_fixture(golden_files("blocks_E64_seed2_B1.npz")[0])
    blob = params.blob_from_record(d, None, E=64)
    buf = ctypes.create_string_buffer(blob, len(blob))
    assert lib.ita_load_weights(h, buf, len(blob)) == 0
    assert lib.ita_mha_int8(h, 3, x.data_ptr(), x.data_ptr(), 1, None) == -1          # layer out of range
    assert lib.ita_mha_int8(h, 0, x.data_ptr(), x.data_ptr(), 0, None) == -1          # empty batch
    # a block-only blob cannot run the whole graph
    assert lib.ita_vitlstm_forward(h, x.data_ptr(), 0, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(),
                                   x.data_ptr(), x.data_ptr(), x.data_ptr(), 1, None, None) == -2
    assert lib.ita_destroy(h) == 0


# ------------------------------------------------------------------------------------------------------------
# round-2 additions: holes named by the round-1 review

def test_softmax_adversarial_rows_on_gpu(torch_cuda):
    """The GPU softmax is a different formulation from the oracle's ((inv >> 8) >> min(d, 15) on packed u16 lanes), and
    inside the kernels it only ever sees what QK^T produces.  Here the adversarial rows of the fixture -- all equal,
    one-hot maxima, the shift 8 / 9 boundary, +-127 / -128 extremes, captured from the reference's
    IntegerApproximatedSoftmax -- and every (max, x) pair go through the same device function in the same register
    layout.  Bit-exact."""
    torch = torch_cuda
    d = params.load_fixture(golden_files("softmax_rows.npz")[0])
    fx = params.load_fixture(FIX_VIT[0])
    eng, _, _ = _engine(fx, 64)
    y = eng.softmax_rows(torch.from_numpy(d["x"]).cuda()).cpu().numpy()
    np.testing.assert_array_equal(y, d["y"])
    # exhaustive over the difference to the row maximum: rows [m, v, -128 ...] for every m >= v, plus ragged row counts
    rows = []
    for m in range(-128, 128, 5):
        for v in range(-128, m + 1, 3):
            r = np.full(128, -128, np.int8)
            r[(m * 7) % 128] = m
            r[(v * 13 + 5) % 128 if (v * 13 + 5) % 128 != (m * 7) % 128 else 0] = v
            rows.append(r)
    rs = np.random.RandomState(5)
    rows += list(rs.randint(-128, 128, size=(37, 128)).astype(np.int8))          # 37: not a multiple of 16 rows
    rows += list(np.clip(rs.normal(100, 6, size=(16, 128)), -128, 127).astype(np.int8))   # many near-maximum entries
    x = np.stack(rows)
    from oracle import oracle as orc
    np.testing.assert_array_equal(eng.softmax_rows(torch.from_numpy(x).cuda()).cpu().numpy(), orc.softmax(x))
    eng.close()


def _two_layer_blob():
    """a 2-layer E = 64 model from two fixtures: layer 0 = seed 0's int8 block, layer 1 = seed 1's"""
    d0, d1 = params.load_fixture(FIX_VIT[0]), params.load_fixture(FIX_VIT[1])
    rec = dict(d0)
    for k, v in d1.items():
        if k.startswith(("attn0.", "ffn0.")):
            rec[k.replace("attn0.", "attn1.").replace("ffn0.", "ffn1.")] = v
    fp = dict(synth.float_params(0, E=64))
    fp1 = synth.float_params(1, E=64)
    for k in ("norms1.0.weight", "norms1.0.bias", "norms2.0.weight", "norms2.0.bias"):
        fp[k.replace(".0.", ".1.")] = fp1[k]
    return params.blob_from_record(rec, fp, E=64, num_layers=2), d0


@pytest.mark.parametrize("mode", [1, 0], ids=["tail_f16x3", "tail_exact_f32"])
def test_two_layer_model_equals_oracle(torch_cuda, oracle, mode):
    """num_layers = 2: the first layer runs with the tokenizer fused in front and writes f32 tokens, the second reads
    them and writes the GEMM planes -- the multi-layer branch of ita_vitlstm_forward had no coverage."""
    torch = torch_cuda
    blob, d = _two_layer_blob()
    eng = host.Engine(blob, device=0)
    eng.set_tail_mode(mode)
    cu = lambda a: torch.from_numpy(a).cuda()
    for img in (d["in0.img_u8"], d["in0.img_u8"].astype(np.float32) / np.float32(255.0)):
        vel, (h, c), tp = eng.forward(cu(img), cu(d["in0.desvel"]), cu(d["in0.quat"]), taps=True)
        ovel, oh, oc, otp = oracle.forward(blob, img, d["in0.desvel"], d["in0.quat"], taps=True)
        for k in ("tokens", "x1", "x2"):
            np.testing.assert_array_equal(tp[k].cpu().numpy(), otp[k], err_msg=k)
        if mode == 0:
            np.testing.assert_array_equal(vel.cpu().numpy(), ovel)
            np.testing.assert_array_equal(h.cpu().numpy(), oh)
        else:
            np.testing.assert_allclose(vel.cpu().numpy(), ovel, atol=2e-5, rtol=0)
            np.testing.assert_allclose(h.cpu().numpy(), oh, atol=2e-5, rtol=0)
            np.testing.assert_allclose(c.cpu().numpy(), oc, atol=2e-5, rtol=0)
    # B beyond one frame per workgroup, second step with carried state
    fr = synth.frames(77, 300)
    v0, st = eng.forward(cu(fr["img_u8"]), cu(fr["desvel"]), cu(fr["quat"]))
    v1, _ = eng.forward(cu(fr["img_u8"][::-1].copy()), cu(fr["desvel"]), cu(fr["quat"]), st)
    sel = [0, 255, 256, 299]
    o0 = oracle.forward(blob, fr["img_u8"][sel], fr["desvel"][sel], fr["quat"][sel])
    o1 = oracle.forward(blob, fr["img_u8"][::-1][sel], fr["desvel"][sel], fr["quat"][sel], o0[1], o0[2])
    tol = dict(atol=2e-5, rtol=0) if mode == 1 else dict(atol=0, rtol=0)
    np.testing.assert_allclose(v0.cpu().numpy()[sel], o0[0], **tol)
    np.testing.assert_allclose(v1.cpu().numpy()[sel], o1[0], **tol)
    eng.close()


@pytest.mark.parametrize("path", FIX_VIT, ids=_ids(FIX_VIT))
def test_tail_from_reference_x2_within_1e4(torch_cuda, oracle, path):
    """north_star's bound for the float tail, taken literally: start BEHIND the int8 blocks, from the reference's own
    LayerNorm2 output, and compare velocity and LSTM state with the reference's at 1e-4 (the end-to-end test has to
    allow 5e-4 because an upstream int8 code can flip behind a float LayerNorm)."""
    torch = torch_cuda
    d = params.load_fixture(path)
    eng, blob, fp = _engine(d, 64)
    cu = lambda a: torch.from_numpy(a).cuda()
    for mode in (1, 0):
        eng.set_tail_mode(mode)
        vel, (h, c) = eng.tail(cu(d["s0.x2"]), cu(d["in0.desvel"]), cu(d["in0.quat"]))
        for got, key in ((vel, "s0.vel"), (h, "s0.h"), (c, "s0.c")):
            np.testing.assert_allclose(got.cpu().numpy(), d[key], atol=1e-4, rtol=0, err_msg=f"{key} mode {mode}")
        # and against the oracle's tail from the same x2: equality in mode 0, 2e-5 in mode 1
        feat = oracle.tail(d["s0.x2"], fp["down_sample.weight"], fp["down_sample.bias"])
        dec = oracle.linear_f32(feat, fp["decoder.weight"], fp["decoder.bias"])
        ovel, oh, oc = oracle.head_from_dec(dec, d["in0.desvel"], d["in0.quat"], fp)
        if mode == 0:
            np.testing.assert_array_equal(vel.cpu().numpy(), ovel)
            np.testing.assert_array_equal(h.cpu().numpy(), oh)
            np.testing.assert_array_equal(c.cpu().numpy(), oc)
        else:
            np.testing.assert_allclose(vel.cpu().numpy(), ovel, atol=2e-5, rtol=0)
            np.testing.assert_allclose(h.cpu().numpy(), oh, atol=2e-5, rtol=0)
    eng.close()


FIX_2L = golden_files("vit2l_*.npz") + golden_files("vit1l_*.npz")


@pytest.mark.parametrize("mode", [1, 0], ids=["tail_f16x3", "tail_exact_f32"])
@pytest.mark.parametrize("path", FIX_2L, ids=_ids(FIX_2L))
def test_two_layer_e128_no_tail_graph(torch_cuda, oracle, path, mode):
    """The second graph family end to end (models/ITA/QAT/model.py:22-87; the only graph whose attention marker
    ITA_spec.mlir:69-85 matches): E = 128, two encoder layers, decoder on the flattened tokens.  Tokenizer launch,
    per layer the stream attention kernel (+ LayerNorm1) and the FFN block kernel (+ LayerNorm2), folded GEMM with
    K = 16384, LSTM head.  Everything up to x2 equals the oracle; the head within 2e-5 (mode 1) / equal (mode 0); and
    the reference's own fixture within the end-to-end int8 bound."""
    torch = torch_cuda
    d = params.load_fixture(path)
    nl = int(d["meta.num_layers"])      # 2: models/ITA/QAT/model.py; 1: models/ITA_single_layer/QAT/model.py
    fp = synth.float_params(int(d["meta.seed"]), E=128, num_layers=nl, tail=False)
    blob = params.blob_from_record(d, fp, E=128, num_layers=nl)
    eng = host.Engine(blob, device=0)
    eng.set_tail_mode(mode)
    cu = lambda a: torch.from_numpy(a).cuda()
    vel, (h, c), tp = eng.forward(cu(d["in0.img_u8"]), cu(d["in0.desvel"]), cu(d["in0.quat"]), taps=True)
    ovel, oh, oc, otp = oracle.forward(blob, d["in0.img_u8"], d["in0.desvel"], d["in0.quat"], taps=True)
    for k in ("tokens", "x1", "x2"):
        np.testing.assert_array_equal(tp[k].cpu().numpy(), otp[k], err_msg=k)
    if mode == 0:
        np.testing.assert_array_equal(tp["dec"].cpu().numpy(), otp["dec"])
        np.testing.assert_array_equal(vel.cpu().numpy(), ovel)
        np.testing.assert_array_equal(h.cpu().numpy(), oh)
        np.testing.assert_array_equal(c.cpu().numpy(), oc)
    else:
        for got, want in ((vel, ovel), (h, oh), (c, oc)):
            np.testing.assert_allclose(got.cpu().numpy(), want, atol=2e-5, rtol=0)
    # against the reference's fixture, from the image: velocity 5e-4; the state only loosely (int8 codes on rounding ties can
    # flip behind the tokenizer's 1e-6 float noise: tests/test_oracle_golden.py states and checks the cause, and
    # test_no_tail_head_from_reference_x2_within_1e4 / test_forward_from_reference_tokens hold the tight bounds)
    for got, key in ((vel, "s0.vel"), (h, "s0.h"), (c, "s0.c")):
        np.testing.assert_allclose(got.cpu().numpy(), d[key], atol=5e-4 if key.endswith("vel") else 1e-2, rtol=0, err_msg=key)
    # second time step with carried state, and a batch beyond one frame per workgroup
    vel1, _ = eng.forward(cu(d["in1.img_u8"]), cu(d["in1.desvel"]), cu(d["in1.quat"]), (cu(d["s0.h"]), cu(d["s0.c"])))
    np.testing.assert_allclose(vel1.cpu().numpy(), d["s1.vel"], atol=5e-4, rtol=0)
    fr = synth.frames(88, 260)
    v, _ = eng.forward(cu(fr["img_u8"]), cu(fr["desvel"]), cu(fr["quat"]))
    sel = [0, 255, 256, 259]
    ov, _, _ = oracle.forward(blob, fr["img_u8"][sel], fr["desvel"][sel], fr["quat"][sel])
    np.testing.assert_allclose(v.cpu().numpy()[sel], ov, atol=2e-5 if mode == 1 else 0, rtol=0)
    eng.close()


# ------------------------------------------------------------------------------------------------------------
# round-3 additions

@pytest.mark.parametrize("path", FIX_2L, ids=_ids(FIX_2L))
def test_no_tail_head_from_reference_x2_within_1e4(torch_cuda, oracle, path):
    """north_star's 1e-4 bound for the float tail on the SECOND graph family (decoder on the flattened tokens, K = 16384):
    start behind the int8 blocks, from the reference's own last LayerNorm2 output, and compare velocity and LSTM state
    with the reference's -- the end-to-end test of this family has to allow 5e-4 / 1e-3 for upstream int8 flips."""
    torch = torch_cuda
    d = params.load_fixture(path)
    nl = int(d["meta.num_layers"])
    fp = synth.float_params(int(d["meta.seed"]), E=128, num_layers=nl, tail=False)
    blob = params.blob_from_record(d, fp, E=128, num_layers=nl)
    eng = host.Engine(blob, device=0)
    cu = lambda a: torch.from_numpy(a).cuda()
    x2 = d[f"s0.x2_{nl - 1}"]
    for mode in (1, 0):
        eng.set_tail_mode(mode)
        vel, (h, c) = eng.tail(cu(x2), cu(d["in0.desvel"]), cu(d["in0.quat"]))
        for got, key in ((vel, "s0.vel"), (h, "s0.h"), (c, "s0.c")):
            np.testing.assert_allclose(got.cpu().numpy(), d[key], atol=1e-4, rtol=0, err_msg=f"{key} mode {mode}")
        dec = oracle.linear_f32(x2.reshape(x2.shape[0], -1), fp["decoder.weight"], fp["decoder.bias"])
        np.testing.assert_allclose(dec, d["s0.dec"], atol=1e-4, rtol=0)          # the oracle's decoder vs the reference's
        ovel, oh, oc = oracle.head_from_dec(dec, d["in0.desvel"], d["in0.quat"], fp)
        tol = dict(atol=2e-5, rtol=0) if mode == 1 else dict(atol=0, rtol=0)
        np.testing.assert_allclose(vel.cpu().numpy(), ovel, **tol)
        np.testing.assert_allclose(h.cpu().numpy(), oh, **tol)
        np.testing.assert_allclose(c.cpu().numpy(), oc, **tol)
    eng.close()


def _blob_with_wide_bias(d, fp, bump):
    """the seed-0 ITAViTLSTM blob with ONE q_proj bias pushed to `bump` accumulator units: sum|w| * 128 + |bias| then
    exceeds 2^22 for that row, the stream kernels' biased-float accumulators no longer cover it (stream_range_ok), and
    ita_load_weights builds no LDS image for the layer -- the forward falls back to the tile-phased block kernels"""
    rec = dict(d, **{"attn0.q_proj.bias": _wide_bias_vector(d, bump)})     # bias_q = rne(b / (s_w s_x)) ~ bump
    return params.blob_from_record(rec, fp, E=64)


def test_block_kernel_fallback_for_wide_accumulators(torch_cuda, oracle):
    """A blob whose int32 bias pushes a row past the 2^22 accumulator range of the stream kernels runs on the block
    kernels (launch_encoder's fallback): same bits as the oracle up to x2, the head within 2e-5, and the serving entry
    that needs the stream kernel (slot-indexed state) refuses BEFORE launching anything."""
    torch = torch_cuda
    d = params.load_fixture(FIX_VIT[0])
    fp = synth.float_params(0, E=64)
    blob = _blob_with_wide_bias(d, fp, 5.0e6)
    t = params.attention_tensors(dict(d, **{"attn0.q_proj.bias": _wide_bias_vector(d, 5.0e6)}), "attn0.", 0)
    assert abs(int(t["attn0.bq"][5])) >= (1 << 22)
    eng = host.Engine(blob, device=0)
    cu = lambda a: torch.from_numpy(a).cuda()
    for img in (d["in0.img_u8"], d["in0.img_u8"].astype(np.float32) / np.float32(255.0)):
        vel, (h, c), tp = eng.forward(cu(img), cu(d["in0.desvel"]), cu(d["in0.quat"]), taps=True)
        ovel, oh, oc, otp = oracle.forward(blob, img, d["in0.desvel"], d["in0.quat"], taps=True)
        for k in ("tokens", "x1", "x2"):
            np.testing.assert_array_equal(tp[k].cpu().numpy(), otp[k], err_msg=k)
        np.testing.assert_allclose(vel.cpu().numpy(), ovel, atol=2e-5, rtol=0)
        np.testing.assert_allclose(h.cpu().numpy(), oh, atol=2e-5, rtol=0)
    # in-place state (hidden_out aliasing hidden_in) stages layer 0's h by a plain copy on this path
    st = (torch.zeros((3, 2, 128), device="cuda"), torch.zeros((3, 2, 128), device="cuda"))
    v_inplace, _ = eng.forward(cu(d["in0.img_u8"]), cu(d["in0.desvel"]), cu(d["in0.quat"]), st, out=(torch.empty((2, 3), device="cuda"), *st))
    np.testing.assert_allclose(v_inplace.cpu().numpy(), ovel, atol=2e-5, rtol=0)
    sh, sc = torch.zeros((3, 4, 128), device="cuda"), torch.zeros((3, 4, 128), device="cuda")
    with pytest.raises(host.ITAError, match="stream kernel"):
        eng.forward_slots(cu(d["in0.img_u8"]), cu(d["in0.desvel"]), cu(d["in0.quat"]), sh, sc, torch.tensor([2, 0], device="cuda"))
    torch.cuda.synchronize()
    assert float(sh.abs().max()) == 0.0 and float(sc.abs().max()) == 0.0        # nothing ran
    eng.close()


def _wide_bias_vector(d, bump):
    s_w, s_x = float(np.float32(d["attn0.q_proj.w_scale"])), float(d["attn0.quant.scale"])
    b = np.array(d["attn0.q_proj.bias"], np.float32).copy()
    b[5] = np.float32(bump * (s_w * s_x))
    return b


def test_full_size_properties_f32_frames(torch_cuda, oracle):
    """BASELINE config 4 size through the graph's OWN input type (f32 frames: stand-alone tokenizer launch + the encoder
    kernel without the fused tokenizer): determinism, batch-composition independence, sampled oracle check."""
    torch = torch_cuda
    d = params.load_fixture(FIX_VIT[0])
    eng, blob, fp = _engine(d, 64)
    B = 1024
    fr = synth.frames(78, B)
    img_np = fr["img_u8"].astype(np.float32) / np.float32(255.0)
    cu = lambda a: torch.from_numpy(a).cuda()
    img, dv, qt = cu(img_np), cu(fr["desvel"]), cu(fr["quat"])
    h = torch.from_numpy(np.random.RandomState(5).standard_normal((3, B, 128)).astype(np.float32) * 0.1).cuda()
    c = torch.from_numpy(np.random.RandomState(6).standard_normal((3, B, 128)).astype(np.float32) * 0.1).cuda()
    v1, (h1, c1) = eng.forward(img, dv, qt, (h, c))
    v2, (h2, c2) = eng.forward(img, dv, qt, (h, c))
    assert torch.equal(v1, v2) and torch.equal(h1, h2) and torch.equal(c1, c2)
    for n in (1, 33, 129, 257):
        sub = torch.arange(1023, 1023 - n, -1, device="cuda")
        vn, (hn, cn) = eng.forward(img[sub], dv[sub], qt[sub], (h[:, sub].contiguous(), c[:, sub].contiguous()))
        assert torch.equal(vn, v1[sub]) and torch.equal(hn, h1[:, sub]) and torch.equal(cn, c1[:, sub]), n
    sel = [0, 256, 511, 1023]
    ov, oh, oc = oracle.forward(blob, img_np[sel], fr["desvel"][sel], fr["quat"][sel], h.cpu().numpy()[:, sel], c.cpu().numpy()[:, sel])
    np.testing.assert_allclose(v1.cpu().numpy()[sel], ov, atol=2e-5, rtol=0)
    np.testing.assert_allclose(c1.cpu().numpy()[:, sel], oc, atol=2e-5, rtol=0)
    eng.close()


def _long_inputs(seed, B, S, E=128):
    rs = np.random.RandomState(seed)
    base = rs.standard_normal((B, S // 32, E)).repeat(32, axis=1) * 18.0
    x = base + rs.standard_normal((B, S, E)) * 9.0
    return np.clip(np.rint(x), -128, 127).astype(np.int8)


@pytest.mark.parametrize("S,B", [(128, 3), (256, 2), (384, 2), (1024, 2), (2048, 1)])
def test_long_sequence_attention_equals_oracle(torch_cuda, oracle, S, B):
    """ita_mha_long_q8 (three sweeps over key tiles, logits never materialised) against the oracle's S x S form, bit for bit;
    at S = 128 also against ita_mha_q8, the single-tile stream kernel."""
    torch = torch_cuda
    for fixture in ("blocks_E128_seed0_B1.npz", "blocks_E128_seed1_B1.npz"):   # seed 0: single-rounding form, seed 1: exact form
        d = params.load_fixture(golden_files(fixture)[0])
        eng, _, _ = _engine(d, 128, with_float=False)
        t = _block_tensors(d)
        xq = _long_inputs(S + B, B, S)
        got = eng.mha_long_q8(torch.from_numpy(xq).cuda()).cpu().numpy()
        want = oracle.mha_q8(xq, t)
        np.testing.assert_array_equal(got, want)
        assert len({r.tobytes() for r in got[0]}) > S // 4
        if S == 128:
            np.testing.assert_array_equal(eng.mha_q8(torch.from_numpy(xq).cuda()).cpu().numpy(), want)
        eng.close()


def test_long_sequence_attention_config5_size(torch_cuda, oracle):
    """BASELINE config 5 as worded: S = 8192 tokens.  Sampled oracle rows (ita_oracle_mha_q8_rows: K, V for all tokens,
    softmax and A.V for the sampled queries), run-to-run determinism, independence of a frame from its batch position; and
    rows whose logits are all equal (probabilities underflow to 0: the context is 0 and out_proj gives its bias codes)."""
    torch = torch_cuda
    d = params.load_fixture(golden_files("blocks_E128_seed0_B1.npz")[0])
    eng, _, _ = _engine(d, 128, with_float=False)
    t = _block_tensors(d)
    S, B = 8192, 2
    xq = _long_inputs(99, B, S)
    xq[1, 4096:] = 0                                   # half of frame 1: identical tokens
    x = torch.from_numpy(xq).cuda()
    a1 = eng.mha_long_q8(x)
    a2 = eng.mha_long_q8(x)
    sw = eng.mha_long_q8(x.flip(0).contiguous())
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and torch.equal(a1, sw.flip(0))
    got = a1.cpu().numpy()
    rows = [0, 1, 127, 128, 4095, 4096, 5000, 8191]
    for b in range(B):
        np.testing.assert_array_equal(got[b][rows], oracle.mha_q8_rows(xq[b], t, rows), err_msg=f"frame {b}")
    assert len({r.tobytes() for r in got[0][::64]}) > 32
    eng.close()
