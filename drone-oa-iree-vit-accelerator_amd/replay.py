"""Trajectory replay harness: PNG frames + data.csv -> velocities + L2 error.

MI355X counterpart of the reference's training-set replay host
(samples/inference_trainingset_custom_dispatch/main.cpp:90-193 main loop, :213-246 load_telemetry_for_image,
:260-292 print_output_tensor).  Same data conventions:

  <root>/<trajectory>/data.csv       header line, then rows with more than 12 comma-separated columns:
                                     [1] timestamp, [2] desired velocity, [3..6] quaternion w,x,y,z,
                                     [10..12] ground-truth velocity x,y,z               (main.cpp:223-240)
  <root>/<trajectory>/<stamp>.png    depth frame; the file stem is its timestamp; matched to the CSV row with
                                     |csv_ts - stamp| < 0.001, first match wins            (main.cpp:216,224-226)
  no matching row                    desired velocity 0, quaternion [1,0,0,0], ground truth 0  (main.cpp:143-149)
  trajectories without data.csv      skipped                                                (main.cpp:109)
  LSTM state                         zero at the start of every trajectory, carried frame to frame (main.cpp:100-106)
  graph inputs                       image / 255, desired velocity / 10 (the graph divides by 10 again), quaternion
                                                                                            (main.cpp:128-133,155)
  error                              Euclidean distance between the raw model output and the ground truth (:282-287)

What differs is the schedule: the reference walks one trajectory after the other, one frame per graph call.
Trajectories are independent streams, so here step t runs frame t of EVERY trajectory that still has frames in
one batched call with slot-indexed state (ita_vitlstm_forward_slots): results per trajectory are the same as
the sequential walk (a frame's result does not depend on its batch -- tests/test_gpu_parity.py), the GPU sees
batches instead of single frames.

Frames that are not 90 x 60 are resized with PIL's bilinear filter.  The reference uses
stbir_resize_uint8_linear (stb is vendored there, not part of this path): the two filters are not bit-identical,
so for resized frames parity with the reference is UNPINNED; 90 x 60 frames go through untouched.

The parsing half of this module (scan_root, load_telemetry, read_frame) needs no GPU; replay() does.
"""
import math
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

FRAME_W, FRAME_H = 90, 60          # main.cpp:123-124
TS_EPSILON = 0.001                 # main.cpp:216


@dataclass
class Telemetry:
    desired_velocity: float = 0.0
    quaternion: tuple = (1.0, 0.0, 0.0, 0.0)
    ground_truth_velocity: tuple = (0.0, 0.0, 0.0)
    found: bool = False


@dataclass
class Trajectory:
    name: str
    frames: List[str] = field(default_factory=list)        # sorted PNG paths
    telemetry: List[Telemetry] = field(default_factory=list)


def _stof(s: str) -> float:
    """std::stof / std::stod accept leading blanks and trailing junk; raise ValueError like invalid_argument"""
    s = s.strip()
    try:
        return float(s)
    except ValueError:
        # longest numeric prefix, as strtod would take it
        for n in range(len(s) - 1, 0, -1):
            try:
                return float(s[:n])
            except ValueError:
                continue
        raise


def load_rows(csv_path: str):
    """rows of data.csv after the header, split on ',' exactly like the reference (no quoting rules)"""
    with open(csv_path, "r", newline="") as f:
        lines = f.read().splitlines()
    return [ln.split(",") for ln in lines[1:]]


def load_telemetry(rows, stamp: str) -> Telemetry:
    """main.cpp:213-246 for one image: first row with > 12 columns whose column 1 is within 1 ms of the stamp."""
    try:
        ts = _stof(stamp)
    except ValueError:
        return Telemetry()
    for row in rows:
        if len(row) <= 12:
            continue
        try:
            if abs(_stof(row[1]) - ts) < TS_EPSILON:
                return Telemetry(_stof(row[2]), tuple(_stof(row[i]) for i in (3, 4, 5, 6)),
                                 tuple(_stof(row[i]) for i in (10, 11, 12)), True)
        except ValueError:
            continue                    # the reference swallows invalid_argument and keeps scanning
    return Telemetry()


def scan_root(root: str) -> List[Trajectory]:
    """sorted trajectory directories, each with its sorted *.png and the matched telemetry (main.cpp:90-114)"""
    out = []
    for name in sorted(os.listdir(root)):
        tdir = os.path.join(root, name)
        if not os.path.isdir(tdir):
            continue
        csv_path = os.path.join(tdir, "data.csv")
        if not os.path.exists(csv_path):
            continue
        rows = load_rows(csv_path)
        frames = sorted(os.path.join(tdir, f) for f in os.listdir(tdir) if os.path.splitext(f)[1] == ".png")
        tel = [load_telemetry(rows, os.path.splitext(os.path.basename(p))[0]) for p in frames]
        out.append(Trajectory(name, frames, tel))
    return out


def read_frame(path: str) -> Optional[np.ndarray]:
    """u8 (60, 90) depth frame: first channel semantics of stbi_load(..., 1) = luminance; resized if needed"""
    from PIL import Image
    try:
        with Image.open(path) as im:
            im = im.convert("L")
            if im.size != (FRAME_W, FRAME_H):
                im = im.resize((FRAME_W, FRAME_H), Image.BILINEAR)
            return np.asarray(im, dtype=np.uint8).copy()
    except Exception:
        return None                      # the reference warns and skips the image (main.cpp:121)


@dataclass
class FrameResult:
    trajectory: str
    frame: str
    output: np.ndarray                   # raw model output (3,)
    ground_truth: np.ndarray
    error: float
    telemetry_found: bool


def replay(engine, root: str, max_batch: int = 1024) -> List[FrameResult]:
    """Runs every trajectory under root through `engine` (host.Engine with a full ITAViTLSTM blob) and returns
    one FrameResult per readable frame, ordered by (trajectory, frame).  Needs a GPU."""
    import torch
    trajs = scan_root(root)
    n = len(trajs)
    if n == 0:
        return []
    dev = torch.device("cuda", engine.device)
    state_h = torch.zeros((3, n, 128), device=dev)      # zero = fresh trajectory (main.cpp:100-106)
    state_c = torch.zeros((3, n, 128), device=dev)
    cursor = [0] * n
    results = {i: [] for i in range(n)}
    while True:
        batch = []                                       # (trajectory index, frame index, u8 frame)
        for i, t in enumerate(trajs):
            while cursor[i] < len(t.frames) and len(batch) < max_batch:
                k = cursor[i]
                cursor[i] += 1
                img = read_frame(t.frames[k])
                if img is not None:                      # unreadable frames are skipped, state untouched
                    batch.append((i, k, img))
                    break
        if not batch:
            break
        imgs = torch.from_numpy(np.stack([b[2] for b in batch])).to(dev)
        tel = [trajs[i].telemetry[k] for i, k, _ in batch]
        dv = torch.tensor([t.desired_velocity / 10.0 for t in tel], dtype=torch.float32, device=dev)
        qt = torch.tensor([t.quaternion for t in tel], dtype=torch.float32, device=dev)
        slots = torch.tensor([b[0] for b in batch], dtype=torch.int32, device=dev)
        vel = engine.forward_slots(imgs, dv, qt, state_h, state_c, slots).cpu().numpy()
        for j, (i, k, _) in enumerate(batch):
            gt = np.asarray(tel[j].ground_truth_velocity, dtype=np.float32)
            d = vel[j] - gt
            err = float(math.sqrt(float(d[0]) * float(d[0]) + float(d[1]) * float(d[1]) + float(d[2]) * float(d[2])))
            results[i].append(FrameResult(trajs[i].name, os.path.basename(trajs[i].frames[k]), vel[j].copy(), gt, err,
                                          tel[j].found))
    return [r for i in range(n) for r in results[i]]


def summarize(results: List[FrameResult]) -> dict:
    """per trajectory and overall: frame count, mean and max L2 error"""
    out, by = {}, {}
    for r in results:
        by.setdefault(r.trajectory, []).append(r.error)
    for k, v in by.items():
        out[k] = {"frames": len(v), "mean_error": float(np.mean(v)), "max_error": float(np.max(v))}
    allv = [r.error for r in results]
    out["_all"] = {"frames": len(allv), "mean_error": float(np.mean(allv)) if allv else 0.0,
                   "max_error": float(np.max(allv)) if allv else 0.0}
    return out
