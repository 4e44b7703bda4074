"""CPU-side checks of the drop-in boundary: the library builds for gfx950, loads, and exports
every symbol include/ita_mi355x.h declares; the weight blob round-trips through the header's
parser.  No compute call is made here (no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO, golden_files
from drone_oa_iree_vit_accelerator_amd import host, params, synth


@pytest.fixture(scope="module")
def so():
    return host.build_extension()


def test_header_symbols_exported(so):
    hdr = open(os.path.join(REPO, "include", "ita_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ita_[a-z0-9_]+|ITA[A-Za-z]+_workgroup(?:_expanded)?)\s*\(", hdr))
    assert declared == set(host.EXPORTED_SYMBOLS), declared ^ set(host.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(so)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.ita_abi_version() == 1


def test_error_paths_without_gpu(so):
    lib = host.lib()
    h = ctypes.c_void_p()
    rc = lib.ita_create(ctypes.byref(h), 0)
    if rc == 0:           # a GPU is present (GPU box): nothing to check here
        lib.ita_destroy(h)
        return
    assert rc == -6 and lib.ita_last_error() == -6 and b"no HIP device" in lib.ita_error_string()
    # the void drop-in symbol reports through ita_last_error
    buf = (ctypes.c_uint16 * (128 * 128))()
    lib.ITASelfAttention_workgroup(buf, buf)
    assert lib.ita_last_error() == -7


def test_code_object_targets_gfx950(so):
    import tempfile
    with tempfile.TemporaryDirectory() as td:      # --offloading drops the extracted code objects in cwd
        import shutil
        cp = shutil.copy(so, td)
        out = os.popen(f"cd {td} && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading {cp} 2>/dev/null").read()
    if not out:
        pytest.skip("llvm-objdump unavailable")
    assert "gfx950" in out


def test_blob_layout_matches_header():
    d = params.load_fixture(golden_files("vitlstm_E64_seed0_B2.npz")[0])
    blob = params.blob_from_record(d, synth.float_params(0), E=64)
    assert blob[:8] == b"ITAW0001"
    n, E, S, P, F, H, L, tail = np.frombuffer(blob[8:40], np.int32)
    assert (E, S, P, F, H, L, tail) == (64, 128, 192, 256, 1, 1, 1)
    ent = np.dtype([("name", "S32"), ("dtype", "<i4"), ("ndim", "<i4"), ("shape", "<i4", 4), ("off", "<i8"),
                    ("nbytes", "<i8")])
    assert ent.itemsize == 72
    tab = np.frombuffer(blob[64:64 + n * 72], ent)
    names = [t["name"].decode() for t in tab]
    for need in ("attn0.wq", "attn0.scal", "ffn0.w2", "tok.conv_w", "tail.conv_w", "dec.w", "lstm.w_hh2", "fc.b"):
        assert need in names
    assert all(int(t["off"]) % 64 == 0 and int(t["off"]) + int(t["nbytes"]) <= len(blob) for t in tab)
    wq = tab[names.index("attn0.wq")]
    got = np.frombuffer(blob[int(wq["off"]):int(wq["off"]) + int(wq["nbytes"])], np.int8).reshape(192, 64)
    np.testing.assert_array_equal(got, d["attn0.q_proj.w_q"])


def test_host_refuses_cpu_tensors(so):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(host.ITAError):
        host.Engine(b"ITAW0001" + b"\0" * 100)


def _retable(blob: bytes, edit):
    """the blob with its tensor table passed through edit(name, entry-dict) -> entry-dict or None (drop)"""
    n = int(np.frombuffer(blob[8:12], np.int32)[0])
    ent = np.dtype([("name", "S32"), ("dtype", "<i4"), ("ndim", "<i4"), ("shape", "<i4", 4), ("off", "<i8"), ("nbytes", "<i8")])
    tab = np.frombuffer(blob[64:64 + n * 72], ent).copy()
    for t in tab:
        edit(t["name"].decode(), t)
    return blob[:64] + tab.tobytes() + blob[64 + n * 72:]


def test_blob_validator_rejects_missized_tensors(so):
    """every tensor the loader knows is checked against (dtype, nbytes) derived from E/P/F/num_layers BEFORE
    anything dereferences it -- a truncated or wrong-E tensor used to be read past its end on host and device."""
    lib = host.lib()
    lib.ita_validate_blob.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    d = params.load_fixture(golden_files("vitlstm_E64_seed0_B2.npz")[0])
    blob = params.blob_from_record(d, synth.float_params(0), E=64)
    bad = ctypes.create_string_buffer(32)
    assert lib.ita_validate_blob(blob, len(blob), bad) == 0
    # each of these tensors was accepted at any size by the round-1 loader
    for name in ("attn0.wk", "attn0.bv", "ffn0.b1", "tok.conv_w", "tok.ln_b", "norm1_0.w", "norm2_0.b", "tail.conv_w",
                 "dec.w", "lstm.w_ih0", "lstm.w_hh2", "lstm.b_ih1", "fc.w", "fc.b", "attn0.scal"):
        def shrink(nm, t, name=name):
            if nm == name:
                t["nbytes"] -= 16
        b2 = _retable(blob, shrink)
        assert lib.ita_validate_blob(b2, len(b2), bad) == -2 and bad.value.decode() == name, name
        assert lib.ita_last_error() == -2
    def retype(nm, t):
        if nm == "attn0.wq":
            t["dtype"] = 0
    b3 = _retable(blob, retype)
    assert lib.ita_validate_blob(b3, len(b3), bad) == -2 and bad.value == b"attn0.wq"
    def rename(nm, t):
        if nm == "ffn0.w2":
            t["name"] = b"ffn0.zz"
    b4 = _retable(blob, rename)
    assert lib.ita_validate_blob(b4, len(b4), bad) == -2 and bad.value == b"ffn0.w2"      # required tensor missing
    def oob(nm, t):
        if nm == "fc.w":
            t["off"] = len(blob) - 64
    b5 = _retable(blob, oob)
    assert lib.ita_validate_blob(b5, len(b5), bad) == -2
    # a blob of another embedding width with this header's E: every E-dependent size is off
    d128 = params.load_fixture(golden_files("blocks_E128_seed0_B1.npz")[0])
    blob128 = params.blob_from_record(d128, None, E=128)
    assert lib.ita_validate_blob(blob128, len(blob128), bad) == 0
    wrong = blob128[:12] + np.int32(64).tobytes() + blob128[16:]
    assert lib.ita_validate_blob(wrong, len(wrong), bad) == -2
    assert lib.ita_validate_blob(blob[:100], 100, bad) == -2 and lib.ita_validate_blob(b"", 0, bad) == -2
