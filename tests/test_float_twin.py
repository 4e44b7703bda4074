"""Row a11: the float twin (true-softmax f32 graph = what the reference's CPU .vmfb holds) restated as this repo's own torch
module, pinned against a fixture produced by the reference's ITALSTMNetVIT (tools/gen_golden.py: gen_float_twin).
It is the CPU baseline of record in bench.py (kind "torch-f32-eager"), never part of the product path."""
import numpy as np
import pytest

from conftest import golden_files
from drone_oa_iree_vit_accelerator_amd import params, synth
from drone_oa_iree_vit_accelerator_amd.float_twin import FloatTwin


@pytest.mark.parametrize("path", golden_files("floattwin_*.npz"))
def test_float_twin_matches_reference_fixture(path):
    d = params.load_fixture(path)
    fp = synth.float_params(int(d["meta.seed"]), E=64)
    assert synth.digest(fp) == str(d["meta.params_sha256"])
    tw = FloatTwin(fp)
    img = d["in0.img_u8"].astype(np.float32) / np.float32(255.0)
    vel, (h, c), tp = tw.forward(img, d["in0.desvel"], d["in0.quat"], taps=True)
    # same libraries, same graph: the only freedom is op fusion inside torch -> 1e-5 (observed ~1e-6)
    np.testing.assert_allclose(tp["tokens"].numpy(), d["s0.tok.out"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(tp["x1"].numpy(), d["s0.x1"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(tp["x2"].numpy(), d["s0.x2"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(tp["dec"].numpy(), d["s0.dec"], atol=1e-5, rtol=0)
    for got, key in ((vel, "s0.vel"), (h, "s0.h"), (c, "s0.c")):
        np.testing.assert_allclose(got.numpy(), d[key], atol=1e-5, rtol=0, err_msg=key)
    vel1, (h1, c1) = tw.forward(d["in1.img_u8"], d["in1.desvel"], d["in1.quat"], (d["s0.h"], d["s0.c"]))   # u8 entry
    for got, key in ((vel1, "s1.vel"), (h1, "s1.h"), (c1, "s1.c")):
        np.testing.assert_allclose(got.numpy(), d[key], atol=1e-5, rtol=0, err_msg=key)


def test_float_twin_is_close_to_the_int8_graph():
    """the int8 graph approximates this one: same weights, velocities within the QAT quantisation error (loose bound --
    this documents the relation of the two graphs, it is not a parity claim)"""
    d = params.load_fixture(golden_files("vitlstm_E64_seed0_B2.npz")[0])
    tw = FloatTwin(synth.float_params(0, E=64))
    vel, _ = tw.forward(d["in0.img_u8"], d["in0.desvel"], d["in0.quat"])
    assert np.abs(vel.numpy() - d["s0.vel"]).max() < 0.1
