#!/usr/bin/env python3
"""bench.py -- frames/s of the ITAViTLSTM int8 forward on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one pass of module.main_graph (tokenizer -> int8 MHA -> int8 FFN -> fusion tail ->
decoder -> 3-layer LSTM -> fc) over one batch of synthetic 60x90 depth frames per GPU
(BASELINE config 4: 1024 frames), inputs resident in HBM in the reference's wire format (u8 pixels,
--image-dtype f32 for the graph's own f32 input), LSTM state carried from step to step,
plus -- for N > 1 -- the RCCL all-gather of the (frames, 3) velocities.  Weak scaling: every GPU
always processes `--frames-per-gpu` independent streams.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel: algorithmic ops per launch / its HIP-event time measured in the
                timed region, against the MI355X peak for its arithmetic (guide: int8 MFMA
                5.0 POP/s dense, f32 157.3 TFLOP/s)
  cpu_baseline  the CPU oracle (oracle/ita_oracle.c, a scalar port) timed on the box's host cores (one process per
                core of a one-GPU share, and one core alone) on a bounded sample of the same workload (N = 1 only)
"""
import argparse
import glob
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

# algorithmic work per frame (BASELINE.md section 2 / SURVEY.md section 8(d)); E=64, S=128, P=192, F=256
STAGE_WORK = {   # stage: (ops per frame, bound, peak in Tera-op/s, arithmetic)
    "tokenizer": (2 * 4.23e6, "mfma", 157.3, "f32"),
    # int8 MHA + int8 FFN (+ both LayerNorms) are ONE launch: ita_stream_kernel
    "encoder": (2 * (12.58e6 + 4.19e6), "mfma", 5000.0, "int8"),
    # fusion conv + decoder run as ONE folded GEMM on split-precision f16 MFMA (priced against the
    # dense f16 peak with the reference graph's algorithmic flops, not the 3x split products)
    "tail_decoder": (2 * (3.32e6 + 2.36e6), "mfma", 2500.0, "f16x3"),
    "lstm_fc": (2 * 0.59e6, "mfma", 2500.0, "f16x3"),
}


# dominant-stage -> kernel whose PMC traffic (profiles/kernel_traffic.json, collected with
# tools/profile_gpu.sh on this same command) is reported as roofline.traffic
STAGE_KERNEL = {"encoder": "ita_stream_kernel<64, true, 1, false>", "tokenizer": "ita_tokenizer_kernel<64, false>",
                "tail_decoder": "ita_gemm_f16x3_kernel<128, 128, 2, 4>", "lstm_fc": "ita_lstm_layer_kernel<4>"}
# algorithmic HBM bytes per frame of each stage as it is cut here (inputs + outputs that cross a launch)
STAGE_BYTES = {"tokenizer": 21600 + 128 * 64 * 4, "encoder": 128 * 64 * 4 + 2 * 128 * 64 * 2,
               "tail_decoder": 2 * 128 * 64 * 2 + 512 * 4, "lstm_fc": 2 * 3 * 128 * 4 * 2 + 12}


def pmc_traffic(kernel, frames):
    try:
        with open(os.path.join(REPO, "profiles", "kernel_traffic.json")) as f:
            t = json.load(f)
        k = t["kernels"][kernel]
        return int(k["hbm_bytes_per_launch"] * frames / 1024), t["source"]
    except Exception:
        return None, None


def _cpu_worker(job):
    blob, n, reps, seed = job
    from drone_oa_iree_vit_accelerator_amd import synth
    from oracle import oracle
    fr = synth.frames(seed, n)
    for _ in range(reps):
        oracle.forward(blob, fr["img_u8"], fr["desvel"], fr["quat"])
    return n * reps


def cpu_baseline(blob, B_target_s=10.0):
    """The CPU oracle on the host cores of this box: one core, then one independent process per core of the GPU's
    CPU share (frames are independent streams, so the scalar port scales by processes).  Must run BEFORE the GPU is
    initialised in this process: the workers are forked."""
    import concurrent.futures as cf
    import multiprocessing as mp
    from drone_oa_iree_vit_accelerator_amd import synth
    from oracle import oracle
    fr = synth.frames(1234, 64)
    t0 = time.perf_counter()
    oracle.forward(blob, fr["img_u8"][:4], fr["desvel"][:4], fr["quat"][:4])
    per = (time.perf_counter() - t0) / 4
    n = int(max(8, min(64, B_target_s / max(per, 1e-6))))
    reps = max(1, int(round(B_target_s / (per * n))))
    t0 = time.perf_counter()
    for _ in range(reps):
        oracle.forward(blob, fr["img_u8"][:n], fr["desvel"][:n], fr["quat"][:n])
    dt1 = time.perf_counter() - t0
    single = n * reps / dt1
    workers = max(1, min(16, os.cpu_count() or 1))      # a one-GPU box's CPU share is 16 cores
    t0 = time.perf_counter()
    with cf.ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("fork")) as ex:
        done = sum(ex.map(_cpu_worker, [(blob, n, reps, 1234 + w) for w in range(workers)]))
    dtm = time.perf_counter() - t0
    return {"value": round(done / dtm, 2), "unit": "frames/s", "cores": workers, "kind": "port",
            "sample": f"{workers} processes x {reps} x {n} frames of the same synthetic workload through "
                      f"oracle/ita_oracle.c (scalar C, one thread per process, {dtm:.1f} s wall)",
            "single_core_value": round(single, 2), "single_core_sample_s": round(dt1, 1),
            "host_cores_available": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--frames-per-gpu", type=int, default=1024)
    ap.add_argument("--image-dtype", choices=["u8", "f32"], default="u8",
                    help="u8: the wire format of the reference host (ita_wire.h; float(pixel)/255.0f of main.cpp:168-169 "
                         "is done on the device, bit-identically, inside the fused tokenizer+encoder kernel); "
                         "f32: the graph's own input type (same fused kernel, 4x the frame bytes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--pipeline", action="store_true", help="two streams: step t's LSTM back (ita_vitlstm_back) runs "
                    "next to the folded GEMM of step t+1's image-only front (ita_vitlstm_front_ev).  Measured: +3 %% at 2048 "
                    "and 4096 frames per GPU; at 1024 the extra Python stream / event calls make the host the bottleneck "
                    "(slower than the default, which is one stream with ita_vitlstm_forward per step)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); 'gloo' only to "
                    "rehearse the N>1 code path with several ranks on one GPU (set ITA_FORCE_DEVICE=0)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from drone_oa_iree_vit_accelerator_amd import dist as itadist
    from drone_oa_iree_vit_accelerator_amd import host, params, synth

    rank, local_rank, world = itadist.env_world()
    cpu = None
    if world == 1 and not a.no_cpu_baseline:      # before anything touches the GPU: the CPU workers are forked
        fx0 = params.load_fixture(os.path.join(REPO, "tests", "golden", "vitlstm_E64_seed0_B2.npz"))
        cpu = cpu_baseline(params.blob_from_record(fx0, synth.float_params(0, E=64), E=64))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if "ITA_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["ITA_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    itadist.init(a.backend, local_rank)

    B, K, W = a.frames_per_gpu, a.steps, a.warmup
    fx = params.load_fixture(os.path.join(REPO, "tests", "golden", "vitlstm_E64_seed0_B2.npz"))
    blob = params.blob_from_record(fx, synth.float_params(0, E=64), E=64)     # weights: seed 0 synthetic-QAT
    eng = host.Engine(blob, device=local_rank, reserve=B)
    fr = synth.frames(1234 + rank, B)
    if a.image_dtype == "u8":
        img = torch.from_numpy(fr["img_u8"]).to(dev)
    else:
        # the reference host's float(pixel) / 255.0f (main.cpp:168-169), IEEE division on the CPU
        img = torch.from_numpy(fr["img_u8"].astype(np.float32) / np.float32(255.0)).to(dev)
    dv, qt = torch.from_numpy(fr["desvel"]).to(dev), torch.from_numpy(fr["quat"]).to(dev)
    state = [(torch.zeros((3, B, 128), device=dev), torch.zeros((3, B, 128), device=dev)) for _ in range(2)]
    vels = [torch.empty((B, 3), device=dev) for _ in range(2)]
    gather = itadist.VelocityGather(B, world, dev)
    pipelined = a.pipeline
    sf, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    ev_front = [torch.cuda.Event() for _ in range(2)]
    ev_back = [torch.cuda.Event() for _ in range(2)]
    ev_enc = [torch.cuda.Event() for _ in range(2)]
    for e in ev_enc:
        e.record()          # creates the HIP event behind it (its handle is passed through the C ABI)
    started = [False, False]
    pending = [None]        # pipelined schedule: the step whose back half has not been enqueued yet
    dvf = dv.reshape(B).contiguous()
    torch.cuda.synchronize()

    def step(i):
        """one time step over this GPU's B streams.  Pipelined form: the image-only front of step i runs on
        stream sf while the LSTM back of step i-1 is still running on stream sb; the recurrence (back(i) after
        back(i-1), same stream) and the buffer reuse (front(i) after back(i-2)) are ordered by events."""
        src, dst = state[i & 1], state[(i + 1) & 1]
        vel = vels[i & 1]
        if not pipelined:
            if world > 1:
                gather.ready()      # the all-gather of step i-2 read this velocity buffer
            eng.forward(img, dv, qt, src, out=(vel, dst[0], dst[1]))
            if world > 1:
                gather.start(vel)
            return
        buf = i & 1
        if started[buf]:
            sf.wait_event(ev_back[buf])
        eng.front(img, buf, stream=sf, encoder_done=ev_enc[buf])
        ev_front[buf].record(sf)
        # the back of the PREVIOUS step goes next to this step's GEMM: next to the encoder (one persistent workgroup
        # per CU, all LDS and VGPRs) it would only delay some of the encoder's workgroups
        if pending[0] is not None:
            run_back(pending[0], ev_enc[buf])
        pending[0] = (i, buf, src, dst, vel)

    def run_back(job, after):
        j, buf, src, dst, vel = job
        sb.wait_event(ev_front[buf])
        if after is not None:
            sb.wait_event(after)
        with torch.cuda.stream(sb):
            if world > 1:
                gather.ready()      # the all-gather of step j-2 read this velocity buffer
            eng.back(dvf, qt, src, (vel, dst[0], dst[1]), buf, stream=sb)
            ev_back[buf].record(sb)
            started[buf] = True
            if world > 1:
                gather.start(vel)

    def flush():
        if pending[0] is not None:
            run_back(pending[0], None)
            pending[0] = None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(W):
        step(i)
    flush()
    gather.finish()
    fence()
    # Untimed pass with HIP events around EVERY stage: the per-stage table and the dominant stage.
    # (Each event costs a ~5 us bubble in the stream, so this pass is ~20 % slower than the timed one.)
    NP = 20
    eng.profile_begin(NP)
    for i in range(NP):
        src, dst = state[(W + i) & 1], state[(W + i + 1) & 1]
        eng.forward(img, dv, qt, src, out=(vels[0], dst[0], dst[1]))
    gather.finish()
    fence()
    stage_all, n_all = eng.profile_end()
    per = {k: v / max(n_all, 1) for k, v in stage_all.items()}
    per["tail_decoder"] = per.pop("tail") + per.pop("decoder")
    per["encoder"] = per.pop("mha") + per.pop("ffn")
    # u8 wire frames: the tokenizer runs inside the encoder kernel (ita_stream_kernel<64, true, 1>), one stage;
    # f32 frames go through the stand-alone tokenizer launch
    fused_tok = not os.environ.get("ITA_SPLIT_TOKENIZER") and a.image_dtype == "u8"
    if fused_tok:
        per["encoder"] += per.pop("tokenizer")
    dom = max(per, key=per.get)
    dom_plugin_stage = {"encoder": "mha", "tail_decoder": "tail", "lstm_fc": "lstm_fc", "tokenizer": "tokenizer"}[dom]
    # Timed region: exactly K steps; HIP events only around the dominant kernel, on every 8th step
    # (its average launch duration for the roofline is measured here, live, on the compute stream).
    if not os.environ.get("ITA_BENCH_NOPROF"):
        eng.profile_begin(min(K, 512), every_n=8, only_stage=dom_plugin_stage)
    t0 = time.perf_counter()
    for i in range(K):
        step(W + NP + i)
    flush()
    gather.finish()
    fence()
    elapsed = time.perf_counter() - t0
    stage_ms, nprof = eng.profile_end()
    dom_ms = stage_ms[dom_plugin_stage] / max(nprof, 1)
    if world > 1:
        t = torch.tensor([elapsed], device=dev if a.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        ops, bound, peak, arith = STAGE_WORK[dom]
        if nprof == 0:
            dom_ms = per[dom]
        achieved = ops * B / (max(dom_ms, 1e-9) * 1e-3) / 1e12
        kname = STAGE_KERNEL.get(dom, dom)
        if dom == "encoder":
            kname = "ita_stream_kernel<64, true, 1, false>" if fused_tok else "ita_stream_kernel<64, true, 0, false>"
        traffic, tsrc = pmc_traffic(kname, B)
        roof = {"kernel": kname, "stage": dom, "bound": bound, "achieved": round(achieved, 3),
                "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 5), "traffic": traffic,
                "traffic_source": tsrc, "algorithmic_hbm_bytes_per_launch": STAGE_BYTES.get(dom, 0) * B,
                "arithmetic": arith, "ops_per_launch": ops * B, "avg_launch_ms": round(dom_ms, 5),
                "launches_timed": nprof, "timing": "HIP events around this kernel on every 8th step of the timed region"}
        if dom == "encoder" and fused_tok:
            # the launch also carries the f32 tokenizer of every frame; `achieved` counts the int8 ops only
            # (conservative).  Mixed form: minimum time = int8 ops / int8 peak + executed f32 MFMA flops / f32 peak.
            tok_flops = 2 * 128 * 64 * 52 * B
            t_min = ops * B / (peak * 1e12) + tok_flops / (157.3e12)
            roof["algorithmic_hbm_bytes_per_launch"] = ((5400 if a.image_dtype == "u8" else 21600) + 2 * 128 * 64 * 2) * B   # frame in, f16 hi/lo planes out
            roof["note"] = ("kernel = tokenizer (f32 MFMA) + int8 MHA + int8 FFN + both LayerNorms of each frame; "
                            "achieved/frac count the int8 ops only")
            roof["mixed"] = {"f32_flops_per_launch": tok_flops, "f32_peak": 157.3, "min_time_ms": round(t_min * 1e3, 5),
                             "frac": round(t_min / (max(dom_ms, 1e-9) * 1e-3), 5)}
        stages = {}
        for k, ms in per.items():
            if k not in STAGE_WORK:
                continue
            o, _, pk, ar = STAGE_WORK[k]
            ach = o * B / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            stages[k] = {"ms": round(ms, 5), "achieved_Tops": round(ach, 3), "peak_Tops": pk, "frac": round(ach / pk, 5),
                         "arithmetic": ar}
        out = {
            "metric": "frames/s on ITAViTLSTM int8, 60x90 depth input",
            "value": round(world * B * K / elapsed, 1), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int8+f32", "data": "synthetic",
            "config": {"workload": f"ITAViTLSTM int8 end-to-end forward (BASELINE config 4): {B} synthetic 60x90 "
                                   "depth frames per GPU per step, LSTM state carried, velocity all-gather",
                       "frames_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "image_dtype": a.image_dtype, "weights": "seed-0 synthetic QAT (tests/golden)",
                       "schedule": "two streams: front(t+1) overlaps back(t)" if pipelined else "one stream"},
            "roofline": roof, "stages": stages,
            "stages_note": "per-stage times from an untimed 20-step pass with events around every stage (each event "
                           "costs a ~5 us stream bubble, so they sum to more than ms_per_step)",
        }
        if not a.no_latency:      # p50 single-frame latency, host enqueue -> result ready
            e1 = host.Engine(blob, device=local_rank, reserve=1)
            i1, d1, q1 = img[:1].contiguous(), dv[:1].contiguous(), qt[:1].contiguous()
            st = (torch.zeros((3, 1, 128), device=dev), torch.zeros((3, 1, 128), device=dev))
            lat = []
            for it in range(250):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                v1, st = e1.forward(i1, d1, q1, st)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t1)
            out["p50_latency_ms_b1"] = round(float(np.median(lat[50:])) * 1e3, 4)
            # the same step replayed from a HIP graph (six launches -> one host call), in-place state
            g1 = e1.graphed_step(1)
            g1.img.copy_(i1); g1.desvel.copy_(d1.reshape(1)); g1.quat.copy_(q1)
            lat = []
            for it in range(250):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                g1()
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t1)
            out["p50_latency_ms_b1_hipgraph"] = round(float(np.median(lat[50:])) * 1e3, 4)
            del g1
            e1.close()
        out["cpu_baseline"] = cpu
    fence()
    if rank == 0:
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
