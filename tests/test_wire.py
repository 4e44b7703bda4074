"""Row n1 (UDP host): the product's wire-format code (include/ita_wire.h, exported from
libita_mi355x.so) against the oracle's restatement of the reference's unpack_frame /
calculate_final_velocity (samples/inference_udp_FPGA_custom_dispatch/main.cpp:320-354,381-417).
Pure host code: runs without a GPU.  Floats compared for equality (same operations)."""
import ctypes
import struct

import numpy as np
import pytest

from drone_oa_iree_vit_accelerator_amd import host


def make_packet(rs, desvel=None, posx=None, quat=None):
    img = rs.randint(0, 256, size=5400).astype(np.uint8).tobytes()
    desvel = float(rs.uniform(2, 8)) if desvel is None else desvel
    posx = float(rs.uniform(0, 60)) if posx is None else posx
    quat = rs.standard_normal(4).astype(np.float32) if quat is None else quat
    return img + struct.pack(">ff", desvel, posx) + struct.pack(">4f", *[float(q) for q in quat])


def product_unpack(pkt, bug):
    out = np.zeros(6, np.float32)
    buf = ctypes.create_string_buffer(pkt, len(pkt))
    rc = host.lib().ita_wire_unpack_packet(buf, len(pkt), int(bug), out.ctypes.data_as(ctypes.c_void_p))
    return rc, out


def product_post(raw, dv, px):
    raw = np.ascontiguousarray(raw, np.float32)
    out = np.zeros(3, np.float32)
    host.lib().ita_wire_postprocess(raw.ctypes.data_as(ctypes.c_void_p), dv, px, out.ctypes.data_as(ctypes.c_void_p))
    return out


@pytest.mark.parametrize("bug", [False, True])
def test_unpack_matches_reference_restatement(oracle, bug):
    host.build_extension()
    rs = np.random.RandomState(3)
    for _ in range(50):
        pkt = make_packet(rs)
        assert len(pkt) == 5424
        rc, got = product_unpack(pkt, bug)
        want = oracle.unpack_packet(pkt, ref_bug=bug)
        assert rc == 0
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    # the sender's values come back when the stride is right; the reference's stride drops 3 of 4 quaternion elements
    q = np.array([0.5, -0.25, 0.125, 2.0], np.float32)
    pkt = make_packet(rs, 4.0, 1.5, q)
    _, good = product_unpack(pkt, False)
    np.testing.assert_array_equal(good, np.array([4.0, 1.5, 0.5, -0.25, 0.125, 2.0], np.float32))
    _, bad = product_unpack(pkt, True)
    np.testing.assert_array_equal(bad, np.array([4.0, 1.5, 0.5, 0.125, 0.0, 0.0], np.float32))
    # short packets are rejected
    rc, _ = product_unpack(pkt[:5000], False)
    assert rc != 0 and oracle.unpack_packet(pkt[:5000]) is None


def test_postprocess_matches_reference_restatement(oracle):
    host.build_extension()
    rs = np.random.RandomState(4)
    cases = [((0.0, 0.0, 0.0), 5.0, 10.0),            # zero norm: no division
             ((3.0, 0.1, -0.2), 5.0, 10.0),           # x clipped to 1
             ((-3.0, 0.1, -0.2), 5.0, 10.0),
             ((0.2, 0.3, 0.1), 6.0, 1.0),             # near-start override, (pos_x/2)*v > 1
             ((0.2, 0.3, 0.1), 1.0, 0.5),             # near-start override floors at 1.0
             ((0.2, 0.3, 0.1), 6.0, 2.0)]             # threshold is strict
    for _ in range(200):
        cases.append((tuple(rs.standard_normal(3) * 0.7), float(rs.uniform(2, 8)), float(rs.uniform(0, 6))))
    for raw, dv, px in cases:
        got = product_post(raw, dv, px)
        want = oracle.final_velocity(raw, dv, px)
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    np.testing.assert_allclose(product_post((3.0, 0.0, 0.0), 5.0, 10.0), [5.0, 0.0, 0.0])
    assert product_post((0.2, 0.3, 0.1), 6.0, 1.0)[0] == 3.0
