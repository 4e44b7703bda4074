"""Trajectory replay harness against the conventions of the reference's
samples/inference_trainingset_custom_dispatch/main.cpp (cited per assertion).  CPU part: directory scan, CSV
matching, PNG reading.  GPU part (-m gpu): the batched replay equals the reference's schedule -- one trajectory
after the other, one frame per call, state carried -- bit for bit."""
import os

import numpy as np
import pytest

from drone_oa_iree_vit_accelerator_amd import host, params, replay, synth

HEADER = "idx,timestamp,desired_vel,quat_1,quat_2,quat_3,quat_4,pos_x,pos_y,pos_z,vel_x,vel_y,vel_z,extra\n"


def _write_png(path, arr):
    from PIL import Image
    Image.fromarray(arr).save(path)


def _make_root(tmp_path, n_traj=3, lens=(4, 2, 5), seed=0):
    rs = np.random.RandomState(seed)
    root = tmp_path / "data"
    root.mkdir()
    meta = {}
    for t in range(n_traj):
        d = root / f"traj_{t:02d}"
        d.mkdir()
        rows, frames = [HEADER], []
        for k in range(lens[t]):
            ts = 100.0 + t + 0.1 * k
            img = rs.randint(0, 256, size=(60, 90)).astype(np.uint8)
            name = f"{ts:.3f}.png"
            _write_png(str(d / name), img)
            dv = float(rs.uniform(2, 8))
            q = rs.standard_normal(4)
            q /= np.linalg.norm(q)
            gt = rs.standard_normal(3)
            rows.append(f"{k},{ts + 0.0004:.4f},{dv:.6f},{q[0]:.6f},{q[1]:.6f},{q[2]:.6f},{q[3]:.6f},0,0,0,"
                        f"{gt[0]:.6f},{gt[1]:.6f},{gt[2]:.6f},x\n")
            frames.append((name, img, dv, q, gt))
        (d / "data.csv").write_text("".join(rows))
        meta[d.name] = frames
    return root, meta


def test_scan_and_telemetry_conventions(tmp_path):
    root, meta = _make_root(tmp_path)
    (root / "no_csv").mkdir()                                    # main.cpp:109: skipped
    _write_png(str(root / "no_csv" / "1.000.png"), np.zeros((60, 90), np.uint8))
    (root / "stray.txt").write_text("not a directory")
    d0 = root / "traj_00"
    (d0 / "notes.txt").write_text("ignored: only .png is collected (main.cpp:110)")
    _write_png(str(d0 / "999.000.png"), np.full((60, 90), 7, np.uint8))   # no telemetry row
    with open(d0 / "data.csv", "a") as f:
        f.write("9,999.0,1,2,3\n")                              # <= 12 columns: ignored (main.cpp:222)
        f.write("9,abc,1,1,0,0,0,0,0,0,1,1,1,x\n")              # invalid number: swallowed (main.cpp:242)
    trajs = replay.scan_root(str(root))
    assert [t.name for t in trajs] == ["traj_00", "traj_01", "traj_02"]
    t0 = trajs[0]
    assert [os.path.basename(p) for p in t0.frames] == sorted(os.path.basename(p) for p in t0.frames)
    assert len(t0.frames) == 5 and os.path.basename(t0.frames[-1]) == "999.000.png"
    for (name, _, dv, q, gt), tel in zip(meta["traj_00"], t0.telemetry):
        assert tel.found                                          # |csv_ts - stamp| = 0.0004 < 0.001 (main.cpp:216,226)
        assert abs(tel.desired_velocity - dv) < 1e-5
        assert np.allclose(tel.quaternion, q, atol=1e-5)          # columns 3..6 = w,x,y,z (main.cpp:229-232)
        assert np.allclose(tel.ground_truth_velocity, gt, atol=1e-5)   # columns 10..12 (main.cpp:235-237)
    miss = t0.telemetry[-1]                                       # defaults (main.cpp:143-149)
    assert not miss.found and miss.desired_velocity == 0.0 and miss.quaternion == (1.0, 0.0, 0.0, 0.0)
    assert miss.ground_truth_velocity == (0.0, 0.0, 0.0)


def test_timestamp_epsilon_and_first_match():
    rows = [r.split(",") for r in ["0,10.0000,1,1,0,0,0,0,0,0,1,2,3,x", "1,10.0020,2,1,0,0,0,0,0,0,4,5,6,x",
                                   "2,10.0005,3,1,0,0,0,0,0,0,7,8,9,x"]]
    assert replay.load_telemetry(rows, "10.0009").desired_velocity == 1.0      # 0.0009 < 0.001: first match wins
    assert replay.load_telemetry(rows, "10.0012").desired_velocity == 2.0      # row 0 is 0.0012 away, row 1 0.0008
    assert not replay.load_telemetry(rows, "10.0035").found
    assert not replay.load_telemetry(rows, "not_a_number").found


def test_read_frame(tmp_path):
    img = np.random.RandomState(1).randint(0, 256, size=(60, 90)).astype(np.uint8)
    _write_png(str(tmp_path / "a.png"), img)
    assert np.array_equal(replay.read_frame(str(tmp_path / "a.png")), img)     # 90 x 60 frames pass through untouched
    big = np.random.RandomState(2).randint(0, 256, size=(120, 180)).astype(np.uint8)
    _write_png(str(tmp_path / "b.png"), big)
    r = replay.read_frame(str(tmp_path / "b.png"))
    assert r.shape == (60, 90) and r.dtype == np.uint8                          # resized (filter parity unpinned)
    (tmp_path / "c.png").write_bytes(b"not a png")
    assert replay.read_frame(str(tmp_path / "c.png")) is None                   # skipped like stbi_load failure (main.cpp:121)


@pytest.mark.gpu
def test_replay_equals_sequential_walk(tmp_path):
    import torch
    root, meta = _make_root(tmp_path, n_traj=3, lens=(4, 2, 5), seed=3)
    (root / "traj_01" / "100.950.png").write_bytes(b"corrupt")   # unreadable frame in the middle: skipped, state untouched
    fx = params.load_fixture(os.path.join(os.path.dirname(__file__), "golden", "vitlstm_E64_seed0_B2.npz"))
    eng = host.Engine(params.blob_from_record(fx, synth.float_params(0, E=64), E=64), device=0)
    res = replay.replay(eng, str(root))
    assert [r.trajectory for r in res] == ["traj_00"] * 4 + ["traj_01"] * 2 + ["traj_02"] * 5
    # the reference's schedule: trajectory by trajectory, frame by frame, B = 1, zero state at the start of each
    i = 0
    for name in sorted(meta):
        st = (torch.zeros((3, 1, 128), device="cuda"), torch.zeros((3, 1, 128), device="cuda"))
        for (fname, img, dv, q, gt) in meta[name]:
            tel = replay.load_telemetry(replay.load_rows(str(root / name / "data.csv")), fname[:-4])
            vel, st = eng.forward(torch.from_numpy(img[None]).cuda(),
                                  torch.tensor([tel.desired_velocity / 10.0], device="cuda"),     # main.cpp:155
                                  torch.tensor([tel.quaternion], device="cuda"), st)
            v = vel.cpu().numpy()[0]
            r = res[i]
            assert r.frame == fname
            assert np.array_equal(r.output, v)
            d = v - np.asarray(tel.ground_truth_velocity, np.float32)
            assert abs(r.error - float(np.sqrt((d.astype(np.float64) ** 2).sum()))) < 1e-5       # main.cpp:282-287
            i += 1
    assert i == len(res)
    s = replay.summarize(res)
    assert s["_all"]["frames"] == 11 and set(s) == {"traj_00", "traj_01", "traj_02", "_all"}
    eng.close()


@pytest.mark.gpu
def test_replay_trajectory_equals_oracle(tmp_path, oracle):
    """row n2 against the ORACLE (round 1 only compared HIP with HIP): every frame of a replayed trajectory, with the
    state the CPU oracle carries through the same schedule (zero state at the trajectory start, desvel / 10 on the host
    as main.cpp:155 does), within the float-tail tolerance; the int8 path behind it is bit-exact, so the tokens are equal."""
    import torch
    root, meta = _make_root(tmp_path, n_traj=2, lens=(4, 3), seed=11)
    fx = params.load_fixture(os.path.join(os.path.dirname(__file__), "golden", "vitlstm_E64_seed0_B2.npz"))
    blob = params.blob_from_record(fx, synth.float_params(0, E=64), E=64)
    eng = host.Engine(blob, device=0)
    res = replay.replay(eng, str(root))
    i = 0
    for name in sorted(meta):
        h = c = None
        rows = replay.load_rows(str(root / name / "data.csv"))
        for (fname, img, dv, q, gt) in meta[name]:
            tel = replay.load_telemetry(rows, fname[:-4])
            ov, h, c = oracle.forward(blob, img[None], np.float32([tel.desired_velocity / 10.0]),
                                      np.float32([tel.quaternion]), h, c)
            np.testing.assert_allclose(res[i].output, ov[0], atol=2e-5, rtol=0, err_msg=f"{name}/{fname}")
            i += 1
    assert i == len(res)
    eng.close()
