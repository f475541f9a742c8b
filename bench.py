#!/usr/bin/env python3
"""bench.py -- frames/s of ORB extract + match on 752x480 frames (BASELINE.json metric), MI355X.

Workload (config.workload = BASELINE.json configs[2]): synthetic frame stream, 752x480, 8 levels,
1000 features, FAST 20/7, lapping {0,1000}; every frame is extracted (ORBextractor::operator()) and matched against
its predecessor with the SearchByProjection core in the "1000x1000" stress setting (window = whole image, levels open,
nnratio 0.8, TH_HIGH 100, sequential claims on).

A step = one pass of the hot path over one batch of B frames that are already resident in HBM.  One process per GPU;
frames shard across ranks with no data-path collective, so scaling is weak: every rank runs B frames
per step and `value` = (world * B * steps) / max-over-ranks(time).

Extra keys on the JSON line:
  roofline      dominant kernel: algorithmic bytes per launch / average launch duration (HIP events on the launch
                stream, measured here, see also profiles/), against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (oracle/, kind "port") timed on this host, one core, bounded sample (rank 0, N=1 only)
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 480, 752
CFG = dict(nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7)
LAP = (0, 1000)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters


def algorithmic_bytes(level_shapes, n_kp):
    """each stage reads its inputs once and writes its outputs once (bytes per frame)."""
    px = [h * w for (h, w) in level_shapes]
    P, P0, P7 = sum(px), px[0], px[-1]
    stages = {
        "pyramid": (P - P7) + (P - P0),
        "fast": P,
        "blur": 2 * P,
        "describe": 749 * n_kp + 512 * n_kp + 32 * n_kp + 28 * n_kp,
        "octree": 0,  # filled by the caller from the measured candidate count
    }
    return stages


def cpu_baseline(frames, offs, budget_s=12.0):
    """Time the CPU oracle (clean-room port of the reference's algorithm) on a bounded sample, one core."""
    from oracle import oracle_py as O
    ex = O.OracleExtractor(**CFG)
    sf = ex.scale_factors
    prev = None
    t0 = time.perf_counter()
    n = 0
    for t in list(range(len(frames))) * 4:                              # cycle the stream until the time budget is used
        mono, kps, desc = ex.extract(frames[t], LAP)
        if prev is not None and len(kps):
            pk, pd, po = prev
            F = O.OracleFrame(kps["x"], kps["y"], kps["octave"], kps["angle"], desc, (0.0, float(W), 0.0, float(H)), sf)
            u = (pk["x"] + np.float32(po[0] - offs[t][0])).astype(np.float32)  # (wrap-around pair at t == 0: large shift, still a valid query set)
            v = (pk["y"] + np.float32(po[1] - offs[t][1])).astype(np.float32)
            m1 = np.full(len(pk), -1, np.int32)
            F.search_by_projection_win(pd, u, v, np.full(len(pk), 1.0e4, np.float32), m1, m1, 0.8, 100, True)
        prev = (kps, desc, offs[t])
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d frames of the same synthetic stream, extract + 1000x1000 match, oracle/liborb_oracle.so (gcc -O2), %.1f s" % (n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80)   # 80 x 256 frames: the drain of the 4-deep pipeline at the end is ~3 % of the region
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256, help="frames per step per GPU")
    ap.add_argument("--streams", type=int, default=4, help="independent pipelines on separate HIP streams (steps alternate)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-match", action="store_true", help="extract only (BASELINE configs[1])")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU; backend "nccl" (= RCCL) for the barrier / MAX-time all-reduce.  ORB_BENCH_BACKEND=gloo lets several
    # ranks rehearse the N>1 path on a box with fewer GPUs (ranks then share devices modulo the device count).
    backend = os.environ.get("ORB_BENCH_BACKEND", "nccl")
    if world > 1:   # before anything initialises the GPU runtime: the host driver only supports dmabuf IPC
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    dev = torch.device("cuda", dev_index)
    local_rank = dev_index
    pkg = importlib.import_module("3_orb_slam3_selfnote_amd")
    synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")

    B = args.batch
    frames, offs = synth.make_stream(1000 + rank, B, H, W)          # B distinct frames of one moving scene
    d_img = torch.from_numpy(frames).to(dev)                         # inputs resident in HBM before timing starts
    # shift of the scene between frame t-1 and t (pair p: queries = slot p, candidates = slot p+1)
    shift = np.zeros((B, 2), dtype=np.float32)
    for p in range(B):
        prev = offs[p - 1] if p > 0 else offs[B - 1]
        shift[p] = (prev[0] - offs[p][0], prev[1] - offs[p][1])
    d_shift = torch.from_numpy(shift).to(dev)
    C = pkg.C

    class Pipe:
        """One independent extract+match pipeline: its own handles, workspace, outputs and HIP stream.  Consecutive steps
        alternate between args.streams pipelines so that the latency-bound kernels of one step (octree, in-order match
        resolve) overlap the throughput-bound kernels of the next (separate HIP streams, no dependency between steps)."""

        def __init__(self):
            self.ex = pkg.ORBextractor(device=local_rank, **CFG)
            self.mt = pkg.ORBmatcher(0.8, True, device=local_rank)
            self.cap = cap = self.ex.configure(H, W, B)
            self.stream = torch.cuda.Stream(device=dev)
            # slot 0 = last frame of the previous step (query side of pair 0); slots 1..B = this step's frames
            self.d_kps = torch.zeros((B + 1, cap, 7), dtype=torch.float32, device=dev)
            self.d_desc = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device=dev)
            self.d_cnt = torch.zeros((B + 1, 2), dtype=torch.int32, device=dev)
            self.d_radius = torch.full((B, cap), 1.0e4, dtype=torch.float32, device=dev)
            self.d_lvl = torch.full((B, cap), -1, dtype=torch.int32, device=dev)
            self.d_slot = torch.empty((B, cap), dtype=torch.int32, device=dev)
            self.d_sobs = torch.empty((B, cap), dtype=torch.uint8, device=dev)
            self.d_moq = torch.empty((B, cap), dtype=torch.int32, device=dev)
            self.d_nm = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.kp1 = self.d_kps[1:]
            self.fs = pkg.FrameStruct(cap, self.kp1.data_ptr(), self.d_desc[1:].data_ptr(), None, 0.0, float(W), 0.0, float(H))

        def step(self):
            cap, ex, mt = self.cap, self.ex, self.mt
            d_kps, d_desc, d_cnt = self.d_kps, self.d_desc, self.d_cnt
            with torch.cuda.stream(self.stream):
                stream = self.stream.cuda_stream
                # carry the last frame of the previous pass into slot 0 (60 KB device copy)
                d_kps[0].copy_(d_kps[B]); d_desc[0].copy_(d_desc[B]); d_cnt[0].copy_(d_cnt[B])
                ex.extract_batch_device(d_img.data_ptr(), H, W, W, H * W, B, self.kp1.data_ptr(), d_desc[1:].data_ptr(), d_cnt[1:].data_ptr(), cap, LAP, stream=stream)
                if args.no_match:
                    return
                # caller-side projection of the previous frame's features into the current frame (pure shift in this stream)
                u = (d_kps[:B, :, 0] + d_shift[:, 0:1]).contiguous()
                v = (d_kps[:B, :, 1] + d_shift[:, 1:2]).contiguous()
                self.d_slot.fill_(-1); self.d_sobs.zero_()                  # Frame ctor: mvpMapPoints = NULL
                qs = pkg.QueryStruct(cap, d_desc.data_ptr(), u.data_ptr(), v.data_ptr(), self.d_radius.data_ptr(), self.d_lvl.data_ptr(), self.d_lvl.data_ptr(), None, None)
                rc = mt.L.orbm_search_by_projection_batch_device(mt.m, C.byref(self.fs), cap, C.c_void_p(d_cnt[1:].data_ptr()), 2, C.byref(qs), cap,
                                                                 C.c_void_p(d_cnt.data_ptr()), 2, B, C.c_float(0.8), 100, 1,
                                                                 C.c_void_p(self.d_slot.data_ptr()), C.c_void_p(self.d_sobs.data_ptr()), C.c_void_p(self.d_moq.data_ptr()),
                                                                 None, C.c_void_p(self.d_nm.data_ptr()), C.c_void_p(stream))
                if rc < 0:
                    raise RuntimeError("orbm_search_by_projection_batch_device rc=%d %s" % (rc, mt.L.orbm_last_error(mt.m)))

    pipes = [Pipe() for _ in range(max(1, args.streams))]
    cap = pipes[0].cap
    level_shapes = [pipes[0].ex.level_shape(l) for l in range(CFG["nlevels"])]
    # set-up: every pipeline runs once so that its lazily sized device buffers and kernel attributes exist before any step is
    # counted, whatever --warmup is (the W warm-up steps below alternate between the pipelines like the timed ones)
    for pp in pipes:
        pp.step()
    torch.cuda.synchronize()
    counter = [0]

    def step():
        pipes[counter[0] % len(pipes)].step()
        counter[0] += 1

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # HIP events inside liborbhip, on each pipeline's own launch stream, around every kernel of every timed step (a ring of
    # event sets per handle): the per-kernel averages below are measured over the timed region itself.
    for pp in pipes:
        pp.ex.set_profiling(True)
        pp.mt.set_profiling(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    acc, nsamp = {}, 0
    for pp in pipes:
        st = pp.ex.stage_ms()
        if not st:
            continue                      # a pipeline that got no step (steps < streams)
        if not args.no_match:
            st.update(pp.mt.stage_ms())
        for k, v in st.items():
            acc[k] = acc.get(k, 0.0) + v
        nsamp += 1
    acc = {k: v / nsamp for k, v in acc.items()}
    # the same kernels alone on the chip (one pipeline, synchronised between steps): isolates kernel quality from sharing
    pipes[0].ex.set_profiling(True)       # resets the event ring
    pipes[0].mt.set_profiling(True)
    for _ in range(4):
        pipes[0].step()
        torch.cuda.synchronize()
    iso = pipes[0].ex.stage_ms()
    if not args.no_match:
        iso.update(pipes[0].mt.stage_ms())
    for pp in pipes:
        pp.ex.set_profiling(False)
        pp.mt.set_profiling(False)
    d_cnt, d_nm = pipes[0].d_cnt, pipes[0].d_nm

    cnt = d_cnt[1:].cpu().numpy()
    n_kp = float(cnt[:, 0].mean())
    nm = d_nm.cpu().numpy()
    # ---- roofline of the dominant KERNEL (per launch).  Stage -> kernel: "pyramid" is nlevels-1 launches of k_resize.
    stage_bytes = algorithmic_bytes(level_shapes, n_kp)
    stage_bytes["match_scan"] = (n_kp + n_kp) * 32 + n_kp * 12        # SURVEY.md 8d, match: descriptors in, indices out
    stage_bytes["match_resolve"] = n_kp * (8 * 4 + 12)                # top-8 keys in, (index, distance, count) out
    kernels = {"pyramid": ("k_resize", CFG["nlevels"] - 1), "fast": ("k_fast", 1), "octree": ("k_octree", 1), "blur": ("k_blur", 1),
               "describe": ("k_describe", 1), "match_scan": ("k_match_scan", 1), "match_resolve": ("k_match_resolve", 1)}
    per_launch = {k: acc[k] / kernels[k][1] for k in acc}
    dom = max(per_launch, key=per_launch.get)
    kname, nl = kernels[dom]
    launch_bytes = stage_bytes.get(dom, 0) * B / nl
    ach = launch_bytes / (per_launch[dom] * 1e-3) / 1e9
    fps = world * B * args.steps / dt
    total_bytes = sum(stage_bytes[k] for k in ("pyramid", "fast", "blur", "describe")) + (0 if args.no_match else stage_bytes["match_scan"])
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")              # measured separately with rocprofv3 --pmc (tools/collect_traffic.py)
    if os.path.exists(tf) and B == 256:
        try:
            tj = json.load(open(tf))
            hit = [v for k, v in tj.items() if k.startswith(kname)]
            if hit:
                traffic = round(hit[0]["hbm_B"])
        except Exception:
            traffic = None

    # VALU issue roofline of the whole step: wave-instructions per launch from the committed SQ_INSTS_VALU profile (a separate
    # rocprofv3 --pmc run of this same workload, tools/collect_sq.py), time measured here.  A wave64 instruction occupies its
    # SIMD for 4 cycles; 256 CUs x 4 SIMDs at the 2.4 GHz peak clock issue at most 614.4 G wave-instructions/s.
    valu = None
    sf = os.path.join(ROOT, "profiles", "sq_counters.json")
    if os.path.exists(sf) and B == 256 and not args.no_match:
        try:
            sj = json.load(open(sf))
            tot = 0.0
            for stage, (kn, launches) in kernels.items():
                hit = [v for k, v in sj.items() if kn in k and "SQ_INSTS_VALU" in v]
                tot += max(h["SQ_INSTS_VALU"] for h in hit) * launches if hit else 0.0
            peak = 256 * 4 * 2.4e9 / 4
            ach_i = tot / (dt / args.steps)
            valu = {"bound": "valu-issue", "achieved": round(ach_i / 1e9, 2), "peak": round(peak / 1e9, 2), "unit": "G wave-instr/s",
                    "frac": round(ach_i / peak, 4), "wave_instructions_per_step": round(tot),
                    "source": "profiles/sq_counters.json (SQ_INSTS_VALU per launch, separate --pmc run); time measured live"}
        except Exception:
            valu = None

    if rank == 0:
        out = {
            "metric": "frames/sec ORB extract+match, 752x480" if not args.no_match else "frames/sec ORB extract, 752x480",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: extract + SearchByProjection match, 1000x1000 candidates, synthetic frame stream",
                       "image": "%dx%d" % (W, H), "nfeatures": 1000, "nlevels": 8, "frames_per_step_per_gpu": B, "streams": len(pipes),
                       "mean_keypoints_per_frame": round(n_kp, 1), "mean_matches_per_frame": round(float(nm.mean()), 1),
                       "sharding": "frames round-robin, one process per GPU, no collective"},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": round(launch_bytes), "avg_launch_ms": round(per_launch[dom], 4),
                         "stage_ms_per_step": {k: round(v, 4) for k, v in acc.items()},
                         "isolated": {"note": "same kernel with nothing else on the chip (1 stream)",
                                      "avg_launch_ms": round(iso[dom] / nl, 4),
                                      "achieved": round(launch_bytes / (iso[dom] / nl * 1e-3) / 1e9, 2),
                                      "frac": round(launch_bytes / (iso[dom] / nl * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                      "stage_ms_per_step": {k: round(v, 4) for k, v in iso.items()}},
                         "pipeline_algorithmic_GBps": round(fps / world * total_bytes / 1e9, 2)},
        }
        if valu is not None and world == 1:
            out["roofline"]["valu_issue"] = valu
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames, offs)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
