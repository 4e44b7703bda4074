"""Row n1 on the GPU: the slot-indexed (persistent per-stream state) forward and the UDP server
binary end to end over the loopback interface, checked against the oracle's pipeline
(unpack -> u8/255 -> forward -> calculate_final_velocity)."""
import os
import socket
import struct
import subprocess
import tempfile
import time

import numpy as np
import pytest

from conftest import golden_files
from drone_oa_iree_vit_accelerator_amd import host, params, synth

pytestmark = pytest.mark.gpu


def _blob():
    d = params.load_fixture(golden_files("vitlstm_E64_seed0_B2.npz")[0])
    return params.blob_from_record(d, synth.float_params(0), E=64)


def test_forward_slots_equals_plain_forward():
    import torch
    blob = _blob()
    eng = host.Engine(blob, device=0)
    B, NS = 5, 16
    fr = synth.frames(21, B)
    cu = lambda a: torch.from_numpy(a).cuda()
    img, dv, qt = cu(fr["img_u8"]), cu(fr["desvel"]), cu(fr["quat"])
    rs = np.random.RandomState(0)
    h0 = (0.1 * rs.standard_normal((3, NS, 128))).astype(np.float32)
    c0 = (0.1 * rs.standard_normal((3, NS, 128))).astype(np.float32)
    slots = np.array([7, 0, 15, 3, 9], np.int32)
    sh, sc = cu(h0.copy()), cu(c0.copy())
    vel = eng.forward_slots(img, dv, qt, sh, sc, cu(slots))
    v2, (h2, c2) = eng.forward(img, dv, qt, (cu(h0[:, slots].copy()), cu(c0[:, slots].copy())))
    assert torch.equal(vel, v2)
    assert torch.equal(sh[:, slots.tolist()], h2) and torch.equal(sc[:, slots.tolist()], c2)
    untouched = [i for i in range(NS) if i not in slots.tolist()]
    np.testing.assert_array_equal(sh.cpu().numpy()[:, untouched], h0[:, untouched])
    # in-place plain forward (hidden_out aliases hidden_in) gives the same result
    hh, cc = cu(h0[:, slots].copy()), cu(c0[:, slots].copy())
    v3, _ = eng.forward(img, dv, qt, (hh, cc), out=(torch.empty_like(v2), hh, cc))
    assert torch.equal(v3, v2) and torch.equal(hh, h2) and torch.equal(cc, c2)
    eng.close()


def _packet(img_u8, desvel, posx, quat):
    return img_u8.tobytes() + struct.pack(">ff", desvel, posx) + struct.pack(">4f", *[float(q) for q in quat])


def test_udp_server_end_to_end(oracle):
    exe = host.build_samples()
    blob = _blob()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    nclients, steps = 3, 4
    with tempfile.NamedTemporaryFile(suffix=".itaw") as f:
        f.write(blob); f.flush()
        srv = subprocess.Popen([exe, "--blob", f.name, "--port", str(port), "--max-packets", str(nclients * steps),
                                "--max-streams", "8", "--max-batch", "8"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                               text=True)
        try:
            line = srv.stdout.readline()
            assert "listening" in line, line
            socks = [socket.socket(socket.AF_INET, socket.SOCK_DGRAM) for _ in range(nclients)]
            for sk in socks:
                sk.settimeout(30)
            h = [np.zeros((3, 1, 128), np.float32) for _ in range(nclients)]
            c = [np.zeros((3, 1, 128), np.float32) for _ in range(nclients)]
            for t in range(steps):
                fr = synth.frames(300 + t, nclients)
                posx = [0.5, 1.9, 30.0]
                for i, sk in enumerate(socks):      # all three in flight -> the server batches them
                    sk.sendto(_packet(fr["img_u8"][i], float(fr["desvel"][i, 0]), posx[i], fr["quat"][i]),
                              ("127.0.0.1", port))
                for i, sk in enumerate(socks):
                    reply, _ = sk.recvfrom(64)
                    assert len(reply) == 12
                    got = np.frombuffer(reply, "<f4")
                    dv = np.float32(fr["desvel"][i, 0])
                    vel, h[i], c[i] = oracle.forward(blob, fr["img_u8"][i:i + 1], np.array([[dv / np.float32(10.0)]], np.float32),
                                                     fr["quat"][i:i + 1], h[i], c[i])
                    want = oracle.final_velocity(vel[0], float(dv), posx[i])
                    # the model output is within 4e-6 of the oracle; normalising and scaling by up to 8 m/s amplifies it
                    np.testing.assert_allclose(got, want, atol=3e-4, rtol=0)
            out, _ = srv.communicate(timeout=60)
            assert srv.returncode == 0, out
            assert f"served {nclients * steps} packets" in out
        finally:
            if srv.poll() is None:
                srv.kill()


def test_front_back_pipeline_equals_forward():
    """five time steps through ita_vitlstm_front / _back on two streams (front(t+1) overlapping back(t),
    ordered by events) give exactly what five plain ita_vitlstm_forward calls give."""
    import torch
    eng = host.Engine(_blob(), device=0)
    B, T = 64, 5
    cu = lambda a: torch.from_numpy(a).cuda()
    frames = [synth.frames(400 + t, B) for t in range(T)]
    imgs = [cu(f["img_u8"]) for f in frames]
    dvs, qts = [cu(f["desvel"]) for f in frames], [cu(f["quat"]) for f in frames]
    # reference: single-stream forwards
    hid = (torch.zeros((3, B, 128), device="cuda"), torch.zeros((3, B, 128), device="cuda"))
    ref = []
    for t in range(T):
        v, hid = eng.forward(imgs[t], dvs[t], qts[t], hid)
        ref.append((v.clone(), hid[0].clone(), hid[1].clone()))
    torch.cuda.synchronize()
    sf, sb = torch.cuda.Stream(), torch.cuda.Stream()
    evf, evb = [torch.cuda.Event() for _ in range(2)], [torch.cuda.Event() for _ in range(2)]
    state = [(torch.zeros((3, B, 128), device="cuda"), torch.zeros((3, B, 128), device="cuda")) for _ in range(2)]
    outs = [torch.empty((B, 3), device="cuda") for _ in range(T)]
    hs = []
    torch.cuda.synchronize()
    for t in range(T):
        buf = t & 1
        if t >= 2:
            sf.wait_event(evb[buf])
        eng.front(imgs[t], buf, stream=sf)
        evf[buf].record(sf)
        sb.wait_event(evf[buf])
        src, dst = state[t & 1], state[(t + 1) & 1]
        eng.back(dvs[t].reshape(B), qts[t], src, (outs[t], dst[0], dst[1]), buf, stream=sb)
        evb[buf].record(sb)
        with torch.cuda.stream(sb):
            hs.append((dst[0].clone(), dst[1].clone()))
    torch.cuda.synchronize()
    for t in range(T):
        assert torch.equal(outs[t], ref[t][0]), t
        assert torch.equal(hs[t][0], ref[t][1]) and torch.equal(hs[t][1], ref[t][2]), t
    eng.close()


def test_graphed_step_equals_eager_forward():
    """the HIP-graph replay of a time step (in-place state, static buffers) gives the bits of the eager calls"""
    import torch
    eng = host.Engine(_blob(), device=0)
    for B in (1, 5):
        fr = [synth.frames(40 + t, B) for t in range(3)]
        g = eng.graphed_step(B)
        st = (torch.zeros((3, B, 128), device="cuda"), torch.zeros((3, B, 128), device="cuda"))
        for t in range(3):
            img, dv, qt = (torch.from_numpy(fr[t][k]).cuda() for k in ("img_u8", "desvel", "quat"))
            g.img.copy_(img); g.desvel.copy_(dv.reshape(B)); g.quat.copy_(qt)
            vg = g().clone()
            ve, st = eng.forward(img, dv, qt, st)
            assert torch.equal(vg, ve), (B, t)
            assert torch.equal(g.h, st[0]) and torch.equal(g.c, st[1])
        if B == 1:
            # a larger graph would reallocate the workspace this one's captured kernels point into: refused while it lives
            with pytest.raises(host.ITAError, match="live captured graph"):
                eng.graphed_step(5)
        del g
    with pytest.raises(host.ITAError):
        eng.pipelined_steps(4, n_steps=1)
    with pytest.raises(host.ITAError):
        eng.pipelined_steps(4, n_steps=8, stages=4)
    eng.close()


@pytest.mark.parametrize("stages", [2, 3])
@pytest.mark.parametrize("B", [3, 130])
def test_pipelined_graph_steps_equal_sequential_forward(B, stages):
    """host.PipelinedSteps: 8 time steps per HIP-graph replay with front(t+1) overlapping back(t) on a second stream -- or,
    stages = 3, encoder(t+2) | folded GEMM(t+1) | LSTM(t) on three (ita_vitlstm_encode / _fold, two plane sets) -- must
    give, step by step and across two replays (state carried), exactly the velocities and the state of 16 sequential
    ita_vitlstm_forward calls."""
    import torch
    fx = params.load_fixture(os.path.join(os.path.dirname(__file__), "golden", "vitlstm_E64_seed0_B2.npz"))
    eng = host.Engine(params.blob_from_record(fx, synth.float_params(0, E=64), E=64), device=0)
    n = 8
    frs = [synth.frames(500 + t, B) for t in range(2 * n)]
    ps = eng.pipelined_steps(B, n, stages)
    got = []
    for rep in range(2):
        for t in range(n):
            f = frs[rep * n + t]
            ps.img[t].copy_(torch.from_numpy(f["img_u8"]))
            ps.desvel[t].copy_(torch.from_numpy(f["desvel"]).reshape(B))
            ps.quat[t].copy_(torch.from_numpy(f["quat"]))
        got.append(ps().clone())
    torch.cuda.synchronize()
    h_pipe, c_pipe = ps.h.clone(), ps.c.clone()
    eng2 = host.Engine(params.blob_from_record(fx, synth.float_params(0, E=64), E=64), device=0)
    st = None
    for t in range(2 * n):
        f = frs[t]
        v, st = eng2.forward(torch.from_numpy(f["img_u8"]).cuda(), torch.from_numpy(f["desvel"]).cuda(),
                             torch.from_numpy(f["quat"]).cuda(), st)
        assert torch.equal(got[t // n][t % n], v), f"step {t}"
    assert torch.equal(h_pipe, st[0]) and torch.equal(c_pipe, st[1])
    del ps
    eng.close(); eng2.close()


def test_three_stage_graph_on_the_two_layer_e128_graph():
    """The three-stage pipeline on the second graph family (E = 128, two encoder layers, no fusion tail): there the encode
    stage runs a tokenizer launch and two layer launches through the shared token buffer while the fold stage of the previous
    step reads the other plane set.  Must equal sequential ita_vitlstm_forward calls bit for bit."""
    import glob
    import torch
    path = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "vit2l_*.npz")))[0]
    d = params.load_fixture(path)
    nl = int(d["meta.num_layers"])
    fp = synth.float_params(int(d["meta.seed"]), E=128, num_layers=nl, tail=False)
    blob = params.blob_from_record(d, fp, E=128, num_layers=nl)
    B, n = 37, 6
    eng, eng2 = host.Engine(blob, device=0), host.Engine(blob, device=0)
    ps = eng.pipelined_steps(B, n, 3)
    frs = [synth.frames(700 + t, B) for t in range(n)]
    for t, f in enumerate(frs):
        ps.img[t].copy_(torch.from_numpy(f["img_u8"]))
        ps.desvel[t].copy_(torch.from_numpy(f["desvel"]).reshape(B))
        ps.quat[t].copy_(torch.from_numpy(f["quat"]))
    got = ps().clone()
    torch.cuda.synchronize()
    st = None
    for t, f in enumerate(frs):
        v, st = eng2.forward(torch.from_numpy(f["img_u8"]).cuda(), torch.from_numpy(f["desvel"]).cuda(),
                             torch.from_numpy(f["quat"]).cuda(), st)
        assert torch.equal(got[t], v), f"step {t}"
    assert torch.equal(ps.h, st[0]) and torch.equal(ps.c, st[1])
    del ps
    eng.close(); eng2.close()


@pytest.mark.parametrize("B,n", [(3, 5), (130, 12)])
def test_library_pipelined_steps_equal_sequential_forward(B, n):
    """ita_vitlstm_pipelined (the library's own two-stream loop: front(t+1) next to back(t), ping-pong state, event
    ordering, an odd step count leaving the state in the internal copy, more steps than partial buffers) must give exactly
    the velocities and the final state of n sequential ita_vitlstm_forward calls."""
    import torch
    fx = params.load_fixture(os.path.join(os.path.dirname(__file__), "golden", "vitlstm_E64_seed0_B2.npz"))
    eng = host.Engine(params.blob_from_record(fx, synth.float_params(0, E=64), E=64), device=0)
    frs = [synth.frames(700 + t, B) for t in range(n)]
    imgs = [torch.from_numpy(f["img_u8"]).cuda() for f in frs]
    dvs = [torch.from_numpy(f["desvel"]).reshape(B).cuda() for f in frs]
    qts = [torch.from_numpy(f["quat"]).cuda() for f in frs]
    rs = np.random.RandomState(9)
    h0 = torch.from_numpy(rs.standard_normal((3, B, 128)).astype(np.float32) * 0.1).cuda()
    c0 = torch.from_numpy(rs.standard_normal((3, B, 128)).astype(np.float32) * 0.1).cuda()
    h, c = h0.clone(), c0.clone()
    vels = [torch.empty((B, 3), dtype=torch.float32, device="cuda") for _ in range(n)]
    sf, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    eng.pipelined(imgs, dvs, qts, (h, c), vels, sf, sb)
    sf.synchronize()                      # the call joins on the front stream
    st = (h0, c0)
    for t in range(n):
        v, st = eng.forward(imgs[t], dvs[t], qts[t], st)
        assert torch.equal(vels[t], v), f"step {t}"
    assert torch.equal(h, st[0]) and torch.equal(c, st[1])
    eng.close()


def test_udp_server_evicts_least_recent_stream(oracle):
    """--max-streams 2 with three senders: the third sender takes the slot of the sender that has been silent longest and
    starts from ZERO state (round 1 dropped every sender beyond the table size forever); a returning evicted sender
    also restarts from zero."""
    exe = host.build_samples()
    blob = _blob()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    order = [0, 1, 0, 2, 2, 1]          # sender per packet: 2 evicts 1 (0 spoke more recently); later 1 evicts 0
    with tempfile.NamedTemporaryFile(suffix=".itaw") as f:
        f.write(blob); f.flush()
        srv = subprocess.Popen([exe, "--blob", f.name, "--port", str(port), "--max-packets", str(len(order)),
                                "--max-streams", "2", "--max-batch", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        try:
            assert "listening" in srv.stdout.readline()
            socks = [socket.socket(socket.AF_INET, socket.SOCK_DGRAM) for _ in range(3)]
            for sk in socks:
                sk.settimeout(30)
            zero = lambda: (np.zeros((3, 1, 128), np.float32), np.zeros((3, 1, 128), np.float32))
            st = {0: zero(), 1: zero(), 2: zero()}
            table = []                      # senders holding a slot, least recently seen first
            for t, i in enumerate(order):
                if i in table:
                    table.remove(i)
                elif len(table) == 2:
                    st[table.pop(0)] = zero()      # evicted: its state is gone
                    st[i] = zero()
                table.append(i)
                fr = synth.frames(700 + t, 1)
                dv = np.float32(fr["desvel"][0, 0])
                socks[i].sendto(_packet(fr["img_u8"][0], float(dv), 10.0, fr["quat"][0]), ("127.0.0.1", port))
                reply, _ = socks[i].recvfrom(64)
                vel, h, c = oracle.forward(blob, fr["img_u8"], np.array([[dv / np.float32(10.0)]], np.float32), fr["quat"], *st[i])
                st[i] = (h, c)
                np.testing.assert_allclose(np.frombuffer(reply, "<f4"), oracle.final_velocity(vel[0], float(dv), 10.0),
                                           atol=3e-4, rtol=0, err_msg=f"packet {t} from sender {i}")
            out, _ = srv.communicate(timeout=60)
            assert srv.returncode == 0 and "2 evictions" in out, out
        finally:
            if srv.poll() is None:
                srv.kill()
