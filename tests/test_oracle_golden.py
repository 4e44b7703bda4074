"""Pins the CPU oracle (oracle/ita_oracle.c) against golden vectors captured from the
reference's own PyTorch int8 modules (tools/gen_golden.py, run in the build container).

Integer stages: bit-exact, each stage fed with the fixture's own input for that stage.
matmul1 (QK^T): the reference goes through a float32 sgemm (torch.ops.quantized.matmul fallback:
dequantize -> matmul -> quantize), so elements whose exact product is within 2e-4 of a rounding
tie may differ by one LSB; the test proves every mismatch is such a near-tie and that there
are at most 3 per fixture.
Float stages: <= 2e-5 absolute (PyTorch's conv / LayerNorm / LSTM summation order is not ours).
"""
import numpy as np
import pytest

from conftest import golden_files
from drone_oa_iree_vit_accelerator_amd import params, synth

FIX_ALL = golden_files("*_seed*.npz")
FIX_VIT = golden_files("vitlstm_*.npz")


def _ids(paths):
    return [p.split("/")[-1][:-4] for p in paths]


def test_fixtures_present():
    assert len(FIX_ALL) >= 6 and len(FIX_VIT) >= 3


def test_softmax_rows(oracle):
    d = params.load_fixture(golden_files("softmax_rows.npz")[0])
    assert float(d["out_scale"]) == 1.0 / 255.0
    y = oracle.softmax(d["x"])
    np.testing.assert_array_equal(y, d["y"])


def test_softmax_inverse_exhaustive(oracle):
    """floor(fp32(255*2^16 / sum)) for every reachable denominator equals the reference's tensor
    expression AND exact integer floor division (so a kernel may use either)."""
    inv_ref = params.load_fixture(golden_files("softmax_inverse.npz")[0])["inv"]
    sums = np.arange(1, 32769, dtype=np.int64)
    ours = np.array([oracle.lib().ita_oracle_softmax_inverse(int(s)) for s in sums])
    np.testing.assert_array_equal(ours, inv_ref)
    np.testing.assert_array_equal(ours[255:], (16711680 // sums)[255:])   # reachable sums are >= 256


def test_softmax_all_diffs(oracle):
    """every diff 0..255 against the closed form y = ((256 >> d) * inv) >> 16"""
    rows = []
    for d in range(256):
        r = np.full(128, -128 + 0, np.int16)
        r[:] = 127 - d
        r[0] = 127
        rows.append(np.clip(r, -128, 127).astype(np.int8))
    x = np.stack(rows)
    y = oracle.softmax(x)
    for d in range(256):
        xi = x[d].astype(np.int64)
        n = np.where(xi.max() - xi > 31, 0, 256 >> np.minimum(xi.max() - xi, 31))
        inv = 16711680 // max(int(n.sum()), 1)
        np.testing.assert_array_equal(y[d], (n * inv) >> 16)


@pytest.mark.parametrize("path", FIX_ALL, ids=_ids(FIX_ALL))
def test_attention_stages(oracle, path):
    d = params.load_fixture(path)
    t = params.attention_tensors(d, "attn0.", 0)
    sc = t["attn0.scal"]
    xq = oracle.quantize(d["s0.attn0.x_q.in"], sc[0])
    np.testing.assert_array_equal(xq, d["s0.attn0.x_q"])
    for nm, key, m in (("Q", "q", 1), ("K", "k", 2), ("V", "v", 3)):
        y = oracle.linear_q(d["s0.attn0.x_q"], t[f"attn0.w{key}"], t[f"attn0.b{key}"], sc[m])
        np.testing.assert_array_equal(y, d[f"s0.attn0.{nm}"])
    # matmul1 with near-tie analysis
    Q, K = d["s0.attn0.Q"].astype(np.int64), d["s0.attn0.K"].astype(np.int64)
    acc = Q @ K.transpose(0, 2, 1)
    ours = np.clip(np.rint(acc.astype(np.float32) * sc[4]), -128, 127).astype(np.int8)
    ref = d["s0.attn0.probs.in"][:, 0]
    bad = np.argwhere(ours != ref)
    assert len(bad) <= 3
    for b, i, j in bad:
        exact = float(acc[b, i, j]) * float(sc[4])
        assert abs(abs(exact - np.floor(exact)) - 0.5) < 2e-4 and abs(int(ours[b, i, j]) - int(ref[b, i, j])) == 1
    probs = oracle.softmax(ref)
    np.testing.assert_array_equal(probs, d["s0.attn0.probs"][:, 0])
    # matmul2 + out_proj + dequant from the fixture's probs / V
    A, V = d["s0.attn0.probs"][:, 0].astype(np.int64), d["s0.attn0.V"].astype(np.int64)
    ctx = np.clip(np.rint((A @ V).astype(np.float32) * sc[5]), -128, 127).astype(np.int8)
    np.testing.assert_array_equal(ctx, d["s0.attn0.out_q.in"])
    o = oracle.linear_q(d["s0.attn0.out_q.in"], t["attn0.wo"], t["attn0.bo"], sc[6])
    np.testing.assert_array_equal(o, d["s0.attn0.out_q"])
    np.testing.assert_array_equal(o.astype(np.float32) * sc[7], d["s0.attn0.out_f"])


@pytest.mark.parametrize("path", FIX_ALL, ids=_ids(FIX_ALL))
def test_attention_block_end_to_end(oracle, path):
    """whole ita_oracle_mha from the float block input; only rows touched by a near-tie logit
    may differ from the reference."""
    d = params.load_fixture(path)
    t = params.attention_tensors(d, "attn0.", 0)
    out, tp = oracle.mha(d["s0.attn0.x_q.in"], t, taps=True)
    np.testing.assert_array_equal(tp["x_q"], d["s0.attn0.x_q"])
    np.testing.assert_array_equal(tp["Q"], d["s0.attn0.Q"])
    np.testing.assert_array_equal(tp["K"], d["s0.attn0.K"])
    np.testing.assert_array_equal(tp["V"], d["s0.attn0.V"])
    ref_l = d["s0.attn0.probs.in"][:, 0]
    bad_rows = {(b, i) for b, i, _ in np.argwhere(tp["logits"] != ref_l)}
    assert len(bad_rows) <= 3
    mask = np.ones(ref_l.shape[:2], bool)
    for b, i in bad_rows:
        mask[b, i] = False
    np.testing.assert_array_equal(tp["probs"][mask], d["s0.attn0.probs"][:, 0][mask])
    np.testing.assert_array_equal(tp["ctx"][mask], d["s0.attn0.out_q.in"][mask])
    np.testing.assert_array_equal(tp["out_q"][mask], d["s0.attn0.out_q"][mask])
    np.testing.assert_array_equal(out[mask], d["s0.attn0.out_f"][mask])
    assert np.abs(tp["out_q"].astype(int) - d["s0.attn0.out_q"].astype(int)).max() <= 2


@pytest.mark.parametrize("path", FIX_ALL, ids=_ids(FIX_ALL))
def test_attention_block_int8_entry(oracle, path):
    """ita_oracle_mha_q8 (int8 codes in, out_proj codes out: the checker of the GPU's ita_mha_q8) from the REFERENCE's
    own quantised block input: equals the taps of the float-in entry, and the reference's out_q except in rows a
    near-tie logit touches."""
    d = params.load_fixture(path)
    t = params.attention_tensors(d, "attn0.", 0)
    got = oracle.mha_q8(d["s0.attn0.x_q"], t)
    _, tp = oracle.mha(d["s0.attn0.x_q.in"], t, taps=True)
    np.testing.assert_array_equal(got, tp["out_q"])
    rows_ok = (tp["logits"] == d["s0.attn0.probs.in"][:, 0]).all(axis=-1)
    np.testing.assert_array_equal(got[rows_ok], d["s0.attn0.out_q"][rows_ok])


def _long_inputs(seed, B, S, E=128):
    """int8 block inputs for long-sequence attention tests: smooth low-frequency token content + noise, moderate range, so
    that rows have a few dominant keys (non-trivial probabilities) at every S"""
    rs = np.random.RandomState(seed)
    base = rs.standard_normal((B, S // 32, E)).repeat(32, axis=1) * 18.0
    x = base + rs.standard_normal((B, S, E)) * 9.0
    return np.clip(np.rint(x), -128, 127).astype(np.int8)


def test_long_attention_rows_entry(oracle):
    """ita_oracle_mha_q8_rows (the checker of ita_mha_long_q8 at S = 8192: selected query rows only) against the S x S
    form ita_oracle_mha_q8 on S = 384 -- identical rows -- and the softmax over a 384-long row against its definition."""
    d = params.load_fixture(golden_files("blocks_E128_seed0_B1.npz")[0])
    t = params.attention_tensors(d, "attn0.", 0)
    xq = _long_inputs(3, 1, 384)
    full = oracle.mha_q8(xq, t)
    rows = [0, 1, 127, 128, 200, 383]
    np.testing.assert_array_equal(oracle.mha_q8_rows(xq[0], t, rows), full[0][rows])
    assert len({r.tobytes() for r in full[0]}) > 300                      # the rows really differ: attention is not degenerate here
    lg = np.random.RandomState(1).randint(-128, 128, size=(3, 384)).astype(np.int8)
    lg[1] = 100                                                           # 384 equal logits: sum = 384 * 256, inv = 170, p = 0
    pr = oracle.softmax(lg)
    for r in range(3):
        sh = (lg[r].max().astype(int) - lg[r].astype(int))
        num = np.where(sh > 31, 0, 256 >> np.minimum(sh, 31))
        inv = int(np.floor(np.float32(1.0) / np.float32(max(num.sum(), 1)) * np.float32(16711680.0)))
        np.testing.assert_array_equal(pr[r], (num * inv) >> 16)
    assert pr[1].max() == 0


@pytest.mark.parametrize("path", FIX_ALL, ids=_ids(FIX_ALL))
def test_ffn_block(oracle, path):
    d = params.load_fixture(path)
    t = params.ffn_tensors(d, "ffn0.", 0)
    out, tp = oracle.ffn(d["s0.ffn0.x_q.in"], t, taps=True)
    np.testing.assert_array_equal(tp["x_q"], d["s0.ffn0.x_q"])
    np.testing.assert_array_equal(tp["h"], d["s0.ffn0.h1_relu"])
    np.testing.assert_array_equal(np.maximum(d["s0.ffn0.h1"], 0), d["s0.ffn0.h1_relu"])
    np.testing.assert_array_equal(tp["out_q"], d["s0.ffn0.out_q"])
    np.testing.assert_array_equal(out, d["s0.ffn0.out_f"])


@pytest.mark.parametrize("path", FIX_VIT, ids=_ids(FIX_VIT))
def test_float_stages(oracle, path):
    d = params.load_fixture(path)
    seed = int(d["meta.seed"])
    fp = synth.float_params(seed, E=64)
    assert synth.digest(fp) == str(d["meta.params_sha256"]), "synthetic parameter generator drifted"
    img = d["in0.img_u8"].astype(np.float32) / np.float32(255.0)
    tok = oracle.tokenizer(img, fp["tokenizer.conv.weight"].reshape(64, 49), fp["tokenizer.conv.bias"],
                           fp["tokenizer.norm.weight"], fp["tokenizer.norm.bias"])
    np.testing.assert_allclose(tok, d["s0.tok.out"], atol=2e-5, rtol=0)
    x1 = oracle.add_ln(d["s0.attn0.x_q.in"], d["s0.attn0.out_f"], fp["norms1.0.weight"], fp["norms1.0.bias"])
    np.testing.assert_allclose(x1, d["s0.x1"], atol=2e-5, rtol=0)
    x2 = oracle.add_ln(d["s0.ffn0.x_q.in"], d["s0.ffn0.out_f"], fp["norms2.0.weight"], fp["norms2.0.bias"])
    np.testing.assert_allclose(x2, d["s0.x2"], atol=2e-5, rtol=0)
    feat, fused = oracle.tail(d["s0.x2"], fp["down_sample.weight"], fp["down_sample.bias"], want_fused=True)
    np.testing.assert_array_equal(fused[:, :16], d["s0.tail.shuffled"])
    np.testing.assert_allclose(fused[:, 16:], d["s0.tail.upsampled"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(feat.reshape(-1, 9, 16, 32), d["s0.tail.conv"], atol=2e-5, rtol=0)
    dec = oracle.linear_f32(d["s0.tail.conv"].reshape(-1, 4608), fp["decoder.weight"], fp["decoder.bias"])
    np.testing.assert_allclose(dec, d["s0.dec"], atol=2e-5, rtol=0)


@pytest.mark.parametrize("path", FIX_VIT, ids=_ids(FIX_VIT))
def test_lstm_head_from_fixture_decoder(oracle, path):
    """The recurrent head on its own: the reference's decoder output in, its (h, c) and velocity out, both time steps.
    No int8 stage sits between input and output here, so the bound is the float one (observed 3e-7): this is what pins
    the oracle's LSTM / fc restatement, which the end-to-end test only sees through a 5e-4 tolerance."""
    d = params.load_fixture(path)
    fp = synth.float_params(int(d["meta.seed"]), E=64)
    vel, h, c = oracle.head_from_dec(d["s0.dec"], d["in0.desvel"], d["in0.quat"], fp)
    for got, key in ((vel, "s0.vel"), (h, "s0.h"), (c, "s0.c")):
        np.testing.assert_allclose(got, d[key], atol=1e-6, rtol=0, err_msg=key)
    if "s1.dec" in d:
        vel1, h1, c1 = oracle.head_from_dec(d["s1.dec"], d["in1.desvel"], d["in1.quat"], fp, d["s0.h"], d["s0.c"])
        for got, key in ((vel1, "s1.vel"), (h1, "s1.h"), (c1, "s1.c")):
            np.testing.assert_allclose(got, d[key], atol=1e-6, rtol=0, err_msg=key)


def _all_block_tensors(d, nl):
    t = {}
    for l in range(nl):
        t.update(params.attention_tensors(d, f"attn{l}.", l))
        t.update(params.ffn_tensors(d, f"ffn{l}.", l))
    return t


FIX_GRAPHS = FIX_VIT + golden_files("vit2l_*.npz") + golden_files("vit1l_*.npz")


@pytest.mark.parametrize("path", FIX_GRAPHS, ids=_ids(FIX_GRAPHS))
def test_forward_from_reference_tokens(oracle, path):
    """Everything BEHIND the tokenizer, started from the reference's own token tensor: encoder layer(s) with their int8
    blocks, fusion tail / flattened tokens, decoder, LSTM head -- against the reference's outputs of the same forward.
    No float stage of ours sits in front of the first quantiser here, so no input code can flip: what is left is the
    documented near-tie exception of matmul1 (<= 3 logits per fixture).  Bounds: velocity 1e-5, h and c 5e-4, x2 2e-6
    except in rows a near-tie logit touches.  The from-the-image tests below add the tokenizer (<= 2e-6 on the tokens)
    and have to allow for input codes that sit on a rounding tie; this test is the tight one."""
    d = params.load_fixture(path)
    E = int(d["meta.E"]) if "meta.E" in d else 64
    nl = int(d["meta.num_layers"]) if "meta.num_layers" in d else 1
    fp = synth.float_params(int(d["meta.seed"]), E=E, num_layers=nl, tail=(E == 64))
    t = _all_block_tensors(d, nl)
    vel, h, c, o = oracle.forward_from_tokens(d["s0.tok.out"], t, fp, nl, d["in0.desvel"], d["in0.quat"])
    np.testing.assert_array_equal(o["x_q0"], d["s0.attn0.x_q"])                   # the first quantiser: bit-exact
    x2_ref = d[f"s0.x2_{nl - 1}"] if f"s0.x2_{nl - 1}" in d else d["s0.x2"]
    off = np.abs(o["x2"] - x2_ref) > 2e-6
    assert off.any(axis=-1).sum() <= (3 if nl == 1 else 2 * 128)   # one layer: the touched rows; two: layer 1 mixes them into every row of a frame
    np.testing.assert_allclose(o["x2"], x2_ref, atol=1e-2, rtol=0)
    np.testing.assert_allclose(vel, d["s0.vel"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(h, d["s0.h"], atol=5e-4, rtol=0)
    np.testing.assert_allclose(c, d["s0.c"], atol=5e-4, rtol=0)
    # against oracle.forward's own composition from the same tokens: identical
    blob = params.blob_from_record(d, fp, E=E, num_layers=nl)
    v2, h2, c2, tp = oracle.forward(blob, d["in0.img_u8"], d["in0.desvel"], d["in0.quat"], taps=True)
    v3, h3, c3, o3 = oracle.forward_from_tokens(tp["tokens"], t, fp, nl, d["in0.desvel"], d["in0.quat"])
    np.testing.assert_array_equal(v3, v2); np.testing.assert_array_equal(h3, h2); np.testing.assert_array_equal(o3["x2"], tp["x2"])


def _input_code_flips(oracle, d, tokens, t):
    """layer-0 input codes that differ from the reference's, each checked to be a rounding tie within float noise"""
    xq = oracle.quantize(tokens, t["attn0.scal"][0])
    ref = d["s0.attn0.x_q"]
    idx = np.argwhere(xq != ref)
    for i in idx:
        i = tuple(i)
        assert abs(int(xq[i]) - int(ref[i])) == 1
        scaled = float(d["s0.tok.out"][i]) * float(t["attn0.scal"][0])
        assert abs(abs(scaled - np.floor(scaled)) - 0.5) < 1e-3, scaled          # the reference's own value sits on the tie
    return len(idx)


@pytest.mark.parametrize("path", FIX_VIT, ids=_ids(FIX_VIT))
def test_full_forward_two_steps(oracle, path):
    """module.main_graph twice, carrying (h, c) like the reference host.  The int8 blocks sit
    behind float LayerNorms, so a 1e-7 float difference can flip an int8 code; the end-to-end
    bound is therefore looser than the per-stage ones (observed: < 2e-4 on velocity)."""
    d = params.load_fixture(path)
    fp = synth.float_params(int(d["meta.seed"]), E=64)
    blob = params.blob_from_record(d, fp, E=64)
    vel0, h0, c0, tp = oracle.forward(blob, d["in0.img_u8"], d["in0.desvel"], d["in0.quat"], taps=True)
    np.testing.assert_allclose(tp["tokens"], d["s0.tok.out"], atol=4e-6, rtol=0)
    np.testing.assert_allclose(tp["x1"], d["s0.x1"], atol=5e-2, rtol=0)
    # isolated int8 flips only: an input code that sits on a rounding tie may flip behind the 1e-6 float difference of the
    # tokenizer (each such code is checked to BE a tie in the reference's own tokens), and then its whole token row of x1
    # moves -- at most 2 of the 256 token rows.  test_forward_from_reference_tokens is the same forward without that noise.
    assert _input_code_flips(oracle, d, tp["tokens"], params.attention_tensors(d, "attn0.", 0)) <= 2
    assert int((np.abs(tp["x1"] - d["s0.x1"]) > 1e-4).any(axis=-1).sum()) <= 2
    np.testing.assert_allclose(tp["dec"], d["s0.dec"], atol=2e-3, rtol=0)
    np.testing.assert_allclose(vel0, d["s0.vel"], atol=5e-4, rtol=0)
    np.testing.assert_allclose(h0, d["s0.h"], atol=5e-4, rtol=0)
    # (c is the unbounded cell state: one flipped token row moves it by up to 8.5e-4 on the seed-2 fixture -- same 1e-3 bound
    #  as the E = 128 family below; h = o * tanh(c) and the velocity stay inside 5e-4)
    np.testing.assert_allclose(c0, d["s0.c"], atol=1e-3, rtol=0)
    vel1, h1, c1 = oracle.forward(blob, d["in1.img_u8"], d["in1.desvel"], d["in1.quat"], d["s0.h"], d["s0.c"])
    np.testing.assert_allclose(vel1, d["s1.vel"], atol=5e-4, rtol=0)
    np.testing.assert_allclose(h1, d["s1.h"], atol=5e-4, rtol=0)
    np.testing.assert_allclose(c1, d["s1.c"], atol=1e-3, rtol=0)
    # float-image entry point (float(pixel) / 255.0f and the float blend) against the u8 wire entry point (integer
    # blend of the pixel codes, 1/65280 in the conv weights): the same linear map, rounded differently -- the tokens
    # agree to 4e-6, the velocity to what the int8 blocks allow behind such a difference
    img_f = d["in0.img_u8"].astype(np.float32) / np.float32(255.0)
    vel0f, _, _, tpf = oracle.forward(blob, img_f, d["in0.desvel"], d["in0.quat"], taps=True)
    np.testing.assert_allclose(tpf["tokens"], tp["tokens"], atol=4e-6, rtol=0)
    np.testing.assert_allclose(tpf["tokens"], d["s0.tok.out"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(vel0f, d["s0.vel"], atol=5e-4, rtol=0)


FIX_2L = golden_files("vit2l_*.npz") + golden_files("vit1l_*.npz")


@pytest.mark.parametrize("path", FIX_2L, ids=_ids(FIX_2L))
def test_two_layer_no_tail_graph(oracle, path):
    """The second graph family: E = 128, no fusion tail -- the decoder reads the flattened tokens -- with two encoder layers
    (models/ITA/QAT/model.py:22-87) and with one (models/ITA_single_layer/QAT/model.py:30-101).  Fixtures = the reference's own
    modules run through its QAT flow (tools/gen_golden.py: gen_vit2l, gen_vit1l).  Same bounds as the ITAViTLSTM fixtures: float stages 2e-5, end to end 5e-4
    (an int8 code can flip behind a float LayerNorm), isolated flips only."""
    d = params.load_fixture(path)
    nl = int(d["meta.num_layers"])
    fp = synth.float_params(int(d["meta.seed"]), E=128, num_layers=nl, tail=False)
    assert synth.digest(fp) == str(d["meta.params_sha256"])
    blob = params.blob_from_record(d, fp, E=128, num_layers=nl)
    assert np.frombuffer(blob[8 + 4 * 7:8 + 4 * 8], np.int32)[0] == 0           # has_tail = 0
    # layer 0 block by block from the reference's own block inputs: bit-exact int8 codes
    t0 = params.attention_tensors(d, "attn0.", 0)
    t0.update(params.ffn_tensors(d, "ffn0.", 0))
    _, tp = oracle.mha(d["s0.tok.out"], t0, taps=True)
    np.testing.assert_array_equal(tp["x_q"], d["s0.attn0.x_q"])
    assert (tp["out_q"] != d["s0.attn0.out_q"]).mean() < 2e-3                    # near-tie logits only (see test_mha)
    vel0, h0, c0, tpf = oracle.forward(blob, d["in0.img_u8"], d["in0.desvel"], d["in0.quat"], taps=True)
    np.testing.assert_allclose(tpf["tokens"], d["s0.tok.out"], atol=4e-6, rtol=0)
    assert _input_code_flips(oracle, d, tpf["tokens"], t0) <= 2
    # From the image the 1e-6 float difference of the tokens can flip a code at ANY of the graph's quantisers (four per layer),
    # and in the two-layer graph layer 1's attention then mixes the moved row into every row of its frame; the K = 16384
    # decoder sum carries that into the unsquashed cell state.  Observed over the fixtures and both image entry points: up to
    # 3.5e-3 on h, 5.5e-3 on c, 3e-5 on the velocity.  The same forward from the reference's own tokens
    # (test_forward_from_reference_tokens) stays inside 5e-4 / 1e-5: the difference is flips, not arithmetic.
    np.testing.assert_allclose(tpf["dec"], d["s0.dec"], atol=2e-2, rtol=0)
    for got, key in ((vel0, "s0.vel"), (h0, "s0.h"), (c0, "s0.c")):
        np.testing.assert_allclose(got, d[key], atol=5e-4 if key.endswith("vel") else 1e-2, rtol=0, err_msg=key)
    vel1, h1, c1 = oracle.forward(blob, d["in1.img_u8"], d["in1.desvel"], d["in1.quat"], d["s0.h"], d["s0.c"])
    for got, key in ((vel1, "s1.vel"), (h1, "s1.h"), (c1, "s1.c")):
        np.testing.assert_allclose(got, d[key], atol=5e-4 if key.endswith("vel") else 1e-2, rtol=0, err_msg=key)
    # the float head from the reference's decoder output: 1e-6
    fpr = dict(fp)
    vel, h, c = oracle.head_from_dec(d["s0.dec"], d["in0.desvel"], d["in0.quat"], fpr)
    np.testing.assert_allclose(vel, d["s0.vel"], atol=1e-6, rtol=0)
