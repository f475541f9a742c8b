#!/usr/bin/env python3
"""bench.py -- frames/s of ORB extract + match (BASELINE.json metric) on MI355X.

Workloads (`--config`):
  euroc  BASELINE configs[2] (the configuration the metric is quoted on): synthetic 752x480 stream, 8 levels, 1000
         features, FAST 20/7, lapping {0,1000}; every frame is extracted (ORBextractor::operator()) and matched against its predecessor
         with the SearchByProjection core in the "1000x1000" stress setting (window = whole image, levels open, nnratio 0.8,
         TH_HIGH 100, sequential claims on).
  tumvi  BASELINE configs[4]: 512x512, 1500 features, last-frame SearchByProjection (ORBmatcher.cc:2027-2289) whose windows come
         from KannalaBrandt8::project with the TUM_512.yaml:9-19 parameters, th 15; map points resident in HBM.

A batch = B frames that are already resident in HBM: 11 extraction launches + the search launches on one stream.  A STEP =
`--batches-per-step` batches (default 224 x 256 = 57344 frames per GPU, ~0.3 s), dealt round-robin to `--streams` independent pipelines,
so that the driver's 20-step run keeps the GPU busy for more than six seconds (its utilisation sampler looks every five).  The batches of a step cycle through `--groups` different
frame sets (2048 distinct frames per GPU by default).

One process per GPU; frames shard across ranks with no data-path collective, so scaling is weak: every rank runs
the same number of frames per step and `value` = world * frames per step * steps / max-over-ranks(time).
`python bench.py --gpus N` with WORLD_SIZE unset starts the N ranks itself (fresh child processes, created before this process
touches the GPU runtime); under torch.distributed.run the ranks come from the environment.

Extra keys on the JSON line:
  roofline         dominant kernel: algorithmic bytes per launch / average launch duration (HIP events on the launch stream,
                   measured over the timed region; see also profiles/), against the 8 TB/s HBM peak; `valu_issue`: the step's vector
                   instructions per second against the MEASURED issue ceiling of its opcode mix (profiles/valu_calib.json)
  verified_frames  frames of the last timed batch of EVERY pipeline (streams x batch frames, distinct frame sets) whose keypoints,
                   descriptors and match indices are byte-equal to the CPU oracle's (non-zero exit status on any mismatch)
  cpu_baseline     the CPU oracle (oracle/, kind "port") timed natively on this host per BASELINE.md section 3: one core and all
                   cores (one frame per thread), median and mean, extract / match split (rank 0, N=1 only)
  host_fed         the same pipeline fed from pinned host memory (all frame sets in turn; H2D on a copy stream ahead of the pipelines, D2H
                   of counts, keypoints, descriptors and match indices), on EVERY rank, MAX time over ranks: whole-job frames/s and PCIe
                   GB/s per GPU; the downloaded outputs of every pipeline are compared with a resident run.  `value` stays the HBM-resident rate.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PCIE_SPEC_GBS = 63.0    # same table: PCIe Gen5 x16
LAP = (0, 1000)

CONFIGS = {
    "euroc": dict(H=480, W=752, cfg=dict(nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7),
                  workload="BASELINE configs[2]: extract + SearchByProjection match, 1000x1000 candidates, synthetic frame stream",
                  metric="frames/sec ORB extract+match, 752x480"),
    "tumvi": dict(H=512, W=512, cfg=dict(nfeatures=1500, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7),
                  workload="BASELINE configs[4]: TUM-VI 512x512, 1500 features, KannalaBrandt8 projection in the last-frame SearchByProjection",
                  metric="frames/sec ORB extract+match, 512x512 KannalaBrandt8"),
}
TUMVI_TH = 15.0         # Tracking.cc:2898: th = 15 for monocular frames
EUROC_K = np.array([458.654, 457.296, 367.215, 248.375], np.float32)                    # Examples/Monocular/EuRoC.yaml:9-12
EUROC_D = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)    # EuRoC.yaml:14-17


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="euroc")
    ap.add_argument("--batch", type=int, default=256, help="frames per batch (one launch sequence)")
    ap.add_argument("--batches-per-step", type=int, default=224, help="batches per step: a step is batch * batches_per_step frames per GPU")
    ap.add_argument("--groups", type=int, default=8, help="distinct frame sets of `batch` frames resident in HBM")
    ap.add_argument("--streams", type=int, default=4, help="independent pipelines on separate HIP streams (batches alternate)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle legs (also skips verified_frames)")
    ap.add_argument("--no-host-fed", action="store_true")
    ap.add_argument("--no-match", action="store_true", help="extract only (BASELINE configs[1])")
    ap.add_argument("--distort", choices=["auto", "none", "euroc"], default="auto",
                    help="euroc: Frame::UndistortKeyPoints with the EuRoC.yaml:9-17 calibration between extraction and search, grid on the "
                         "undistorted-corner bounds (Frame.cc:837-899) - what the reference does on this configuration; auto = euroc for --config euroc")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """`--gpus N` without a launcher: N fresh child processes, one per GPU, created before this process makes any GPU call (it
    never does).  Rank 0 inherits stdout (the JSON line); the exit status is the worst of the ranks'."""
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    return max((abs(rc) for rc in rcs), default=0)


def host_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota when there is one (a GPU box hands
    each job a 16-core share of a much larger host), else by 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except Exception:
        pass
    return max(1, min(n, quota if quota is not None else 16))


def algorithmic_bytes(level_shapes, n_kp):
    """each stage reads its inputs once and writes its outputs once (bytes per frame)."""
    px = [h * w for (h, w) in level_shapes]
    P, P0, P7 = sum(px), px[0], px[-1]
    return {"pyramid": (P - P7) + (P - P0), "fast": P, "blur": 2 * P, "describe": 749 * n_kp + 512 * n_kp + 32 * n_kp + 28 * n_kp, "octree": 0}


def stats_ms(a):
    a = np.asarray(a, dtype=np.float64)
    return {"median": round(float(np.median(a)), 3), "mean": round(float(a.mean()), 3)}


def cpu_legs(conf, groups, g_last, gpu_last, scene_host, cap, nthreads, distort=None):
    """BASELINE.md section 3 over the oracle (native driver, oracle/orb_cpu_bench.c) + the parity tie of the timed region:
    one core over the frame set the GPU's last timed batch worked on (every frame compared), all cores over every resident set."""
    from oracle import oracle_py as O
    ex = O.OracleExtractor(**conf["cfg"])
    mode = 1 if conf is CONFIGS["tumvi"] else 0

    def scene_of(g):
        if mode == 0:
            return None
        s = scene_host[g]
        return dict(cam_type=1, cam=s["cam"], Xw=s["Xw"], has_mp=s["has"], Tcw=s["Tcw"], Tlw=s["Tlw"], th=TUMVI_TH, check_ori=1,
                    bounds=(0.0, float(conf["W"]), 0.0, float(conf["H"])))

    frames, offs = groups[g_last]
    B = len(frames)
    r1 = O.bench_stream(ex, frames, offs, B, threads=1, warmup=50, lap=LAP, cap=cap, mode=mode, nnratio=0.8, th_high=100, scene=scene_of(g_last), distort=distort)
    # ---- parity tie: every pipeline's last timed batch against the oracle, frame by frame (the pipeline of the very last batch against
    # the one-core run, the others against the all-cores run of their frame set, below)
    bad = []
    verified = [0]

    def compare(r, got, tag):
        for t in range(B):
            n = int(r["counts"][t, 0])
            ok = n == int(got["cnt"][t, 0]) and int(r["counts"][t, 1]) == int(got["cnt"][t, 1])
            ok = ok and r["kps"][t, :n].tobytes() == got["kps"][t, :n].tobytes() and np.array_equal(r["desc"][t, :n], got["desc"][t, :n])
            if ok and "match" in got:
                nl = int(r["counts"][(t - 1) % B, 0])
                m = nl if mode == 0 else n           # mode 0: match_of_query of the last frame's keypoints; mode 1: slot array of the current frame
                ok = int(r["nmatch"][t]) == int(got["nm"][t]) and np.array_equal(r["moq"][t, :m], got["match"][t, :m])
            if ok:
                verified[0] += 1
            else:
                bad.append((tag, t))
    assert gpu_last[0][0] == g_last
    compare(r1, gpu_last[0][1], 0)
    tot1 = r1["ms_extract"] + r1["ms_match"]
    one = {"frames": B, "warmup_frames": 50, "extract_ms": stats_ms(r1["ms_extract"]), "match_ms": stats_ms(r1["ms_match"]), "total_ms": stats_ms(tot1),
           "fps": round(B / (r1["wall_extract"] + r1["wall_match"]), 3), "extract_fps": round(B / r1["wall_extract"], 3)}
    # ---- all cores, one frame per thread, every resident frame set (>= 2000 frames)
    nfr, we, wm = 0, 0.0, 0.0
    for g in range(len(groups)):
        fr, of = groups[g]
        ra = O.bench_stream(ex, fr, of, len(fr), threads=nthreads, warmup=nthreads if g == 0 else 0, lap=LAP, cap=cap, mode=mode, nnratio=0.8, th_high=100,
                            scene=scene_of(g), distort=distort)
        nfr += len(fr); we += ra["wall_extract"]; wm += ra["wall_match"]
        for i in range(1, len(gpu_last)):
            if gpu_last[i][0] == g:
                compare(ra, gpu_last[i][1], i)
    allc = {"threads": nthreads, "frames": nfr, "fps": round(nfr / (we + wm), 3), "extract_fps": round(nfr / we, 3), "match_fps": round(nfr / wm, 3)}
    out = {"value": one["fps"], "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "oracle/liborb_oracle.so (gcc -O3 -ffp-contract=off), native timer around each call (orb_cpu_bench.c), same synthetic stream: "
                     "1 core: %d frames after 50 warm-up frames; %d threads, one frame per thread: %d frames" % (B, nthreads, nfr),
           "one_core": one, "all_cores": allc}
    return out, verified[0], bad


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))           # parent: no torch, no HIP
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # The contract is ONE JSON line on stdout.  Native libraries write there too (Gloo: "Rank 0 is connected to ..."), so this process's
    # stdout is pointed at stderr for the whole run and the line goes to a private copy of the original descriptor.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(line_fd, (json.dumps(obj) + "\n").encode())
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or leave WORLD_SIZE unset\n"
                         % (args.gpus, world, args.gpus))
        sys.exit(2)
    conf = CONFIGS[args.config]
    H, W, CFG = conf["H"], conf["W"], conf["cfg"]
    B, S, G, NB = args.batch, max(1, args.streams), max(1, args.groups), max(1, args.batches_per_step)
    backend = os.environ.get("ORB_BENCH_BACKEND", "nccl")
    selftest = os.environ.get("ORB_BENCH_SELFTEST") == "1"   # plumbing test without a device (tests/test_bench_launch.py): no hot path runs
    if world > 1:   # before anything initialises the GPU runtime: the host driver only supports dmabuf IPC
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    shard = importlib.import_module("3_orb_slam3_selfnote_amd.shard")
    frames_per_step = B * NB

    if selftest:
        shard.init_distributed(backend if backend != "nccl" else "gloo")
        dt = shard.timed_steps(lambda: time.sleep(0.002 * (rank + 1)), args.steps, args.warmup, world=world)
        hf = None
        if not args.no_host_fed:   # the host-fed leg's plumbing: every rank runs it, MAX over ranks, whole-job aggregate
            tf = shard.timed_steps(lambda: time.sleep(0.003 * (rank + 1)), 1, 0, world=world)
            hf = {"value": round(world * 48 * B / tf, 2), "unit": "frames/s", "n_gpus": world, "seconds_max_over_ranks": round(tf, 5)}
        if rank == 0:
            emit({"selftest": True, "metric": conf["metric"], "value": round(shard.aggregate_fps(frames_per_step, args.steps, world, dt), 2),
                              "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
                              "scaling": "weak", "host_fed": hf, "data": "none (ORB_BENCH_SELFTEST: no device work, launcher / barrier / MAX plumbing only)"})
        shard.finish_distributed()
        return

    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % ndev           # ORB_BENCH_BACKEND=gloo lets several ranks rehearse the N>1 path on fewer GPUs
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    shard.init_distributed(backend, device=dev if backend == "nccl" else None)
    pkg = importlib.import_module("3_orb_slam3_selfnote_amd")
    synth = importlib.import_module("3_orb_slam3_selfnote_amd.synth")
    C = pkg.C
    tumvi = args.config == "tumvi"
    distort = (args.distort == "euroc" or (args.distort == "auto" and not tumvi)) and not tumvi and not args.no_match
    bounds = pkg.image_bounds(W, H, EUROC_K, EUROC_D) if distort else (0.0, float(W), 0.0, float(H))

    # ---- resident inputs: G frame sets of B frames each (distinct crops of G moving synthetic scenes)
    groups = [synth.make_stream(1000 + 64 * rank + g, B, H, W) for g in range(G)]
    d_img = [torch.from_numpy(f).to(dev) for f, _ in groups]
    d_shift = []
    for f, offs in groups:   # pair p: queries = frame p-1 (wrapping to B-1), candidates = frame p
        sh = np.array([(offs[p - 1][0] - offs[p][0], offs[p - 1][1] - offs[p][1]) for p in range(B)], dtype=np.float32)
        d_shift.append(torch.from_numpy(sh).to(dev))

    class Pipe:
        """One independent extract+match pipeline: its own handles, workspace, outputs and HIP stream.  Consecutive batches alternate
        between the pipelines so that the latency-bound kernels of one batch (octree, in-order match resolve) overlap the
        throughput-bound kernels of the next (separate HIP streams, no dependency between batches)."""

        def __init__(self):
            self.ex = pkg.ORBextractor(device=dev_index, **CFG)
            self.mt = pkg.ORBmatcher(0.8 if not tumvi else 0.9, True, device=dev_index)
            if not tumvi:
                self.mt.set_scan_mode(1)      # the caller knows its windows cover the frame: no grid-window walk launch (it would only vote and leave)
            self.cap = cap = self.ex.configure(H, W, B)
            self.stream = torch.cuda.Stream(device=dev)
            # slot 0 = the last frame of the set (query side of pair 0); slots 1..B = the set's frames
            self.d_kps = torch.zeros((B + 1, cap, 7), dtype=torch.float32, device=dev)
            self.d_desc = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device=dev)
            self.d_cnt = torch.zeros((B + 1, 2), dtype=torch.int32, device=dev)
            self.d_radius = torch.full((B, cap), 1.0e4, dtype=torch.float32, device=dev)
            self.d_lvl = torch.full((B, cap), -1, dtype=torch.int32, device=dev)
            self.d_slot = torch.empty((B, cap), dtype=torch.int32, device=dev)
            self.d_sobs = torch.empty((B, cap), dtype=torch.uint8, device=dev)
            self.d_moq = torch.empty((B, cap), dtype=torch.int32, device=dev)
            self.d_nm = torch.zeros((B,), dtype=torch.int32, device=dev)
            # mvKeysUn (Frame.cc:837-870): a second keypoint block when the calibration has distortion, else mvKeysUn = mvKeys
            self.d_kps_un = torch.zeros((B + 1, cap, 7), dtype=torch.float32, device=dev) if distort else self.d_kps
            self.fs = pkg.FrameStruct(cap, self.d_kps_un[1:].data_ptr(), self.d_desc[1:].data_ptr(), None, *bounds)
            self.group = -1

        def batch(self, g, images=None, scene=None):
            cap, ex, mt = self.cap, self.ex, self.mt
            d_kps, d_desc, d_cnt = self.d_kps, self.d_desc, self.d_cnt
            self.group = g
            with torch.cuda.stream(self.stream):
                stream = self.stream.cuda_stream
                img = d_img[g] if images is None else images
                ex.extract_batch_device(img.data_ptr(), H, W, W, H * W, B, d_kps[1:].data_ptr(), d_desc[1:].data_ptr(), d_cnt[1:].data_ptr(), cap, LAP, stream=stream)
                if args.no_match:
                    return
                d_kun = self.d_kps_un
                if distort:   # the Frame constructor's UndistortKeyPoints for the B frames of the batch
                    mt.undistort_batch_device(d_kps[1:].data_ptr(), cap, d_cnt[1:].data_ptr(), 2, B, EUROC_K, EUROC_D, d_kun[1:].data_ptr(), stream=stream)
                    d_kun[0].copy_(d_kun[B])
                # the set's last frame is the predecessor of its first (60 KB device copy)
                d_kps[0].copy_(d_kps[B]); d_desc[0].copy_(d_desc[B]); d_cnt[0].copy_(d_cnt[B])
                self.d_slot.fill_(-1); self.d_sobs.zero_()                  # Frame ctor: mvpMapPoints = NULL
                if not tumvi:
                    # caller-side projection of the previous frame's features into the current frame (pure shift in this stream)
                    u = (d_kun[:B, :, 0] + d_shift[g][:, 0:1]).contiguous()
                    v = (d_kun[:B, :, 1] + d_shift[g][:, 1:2]).contiguous()
                    qs = pkg.QueryStruct(cap, d_desc.data_ptr(), u.data_ptr(), v.data_ptr(), self.d_radius.data_ptr(), self.d_lvl.data_ptr(), self.d_lvl.data_ptr(), None, None)
                    rc = mt.L.orbm_search_by_projection_batch_device(mt.m, C.byref(self.fs), cap, C.c_void_p(d_cnt[1:].data_ptr()), 2, C.byref(qs), cap,
                                                                     C.c_void_p(d_cnt.data_ptr()), 2, B, C.c_float(0.8), 100, 1,
                                                                     C.c_void_p(self.d_slot.data_ptr()), C.c_void_p(self.d_sobs.data_ptr()), C.c_void_p(self.d_moq.data_ptr()),
                                                                     None, C.c_void_p(self.d_nm.data_ptr()), C.c_void_p(stream))
                else:
                    sc = scene[g]     # the map: world points, poses (resident in HBM); descriptors / octaves / angles = the last frame's own
                    last = pkg.LastFrameStruct(cap, sc["has"].data_ptr(), sc["Xw"].data_ptr(), d_desc.data_ptr(), d_kps.data_ptr(), None,
                                               sc["Tcw"].data_ptr(), sc["Tlw"].data_ptr())
                    rc = mt.L.orbm_search_by_projection_last_frame_batch_device(
                        mt.m, C.byref(self.fs), cap, C.c_void_p(d_cnt[1:].data_ptr()), 2, C.byref(last), cap, C.c_void_p(d_cnt.data_ptr()), 2, B,
                        sf_host.ctypes.data_as(C.c_void_p), len(sf_host), 1, kb8.ctypes.data_as(C.c_void_p), C.c_float(0.0), C.c_float(0.0), C.c_float(TUMVI_TH), 1, 1,
                        C.c_void_p(self.d_slot.data_ptr()), C.c_void_p(self.d_sobs.data_ptr()), C.c_void_p(self.d_moq.data_ptr()), C.c_void_p(self.d_nm.data_ptr()),
                        C.c_void_p(stream))
                if rc < 0:
                    raise RuntimeError("search rc=%d %s" % (rc, mt.L.orbm_last_error(mt.m)))

    pipes = [Pipe() for _ in range(S)]
    cap = pipes[0].cap
    level_shapes = [pipes[0].ex.level_shape(l) for l in range(CFG["nlevels"])]
    sf_host = np.ascontiguousarray(pipes[0].ex.GetScaleFactors(), dtype=np.float32)
    kb8 = np.ascontiguousarray(synth.TUMVI_KB8)

    # ---- tumvi: the map.  One untimed extraction per frame set gives the keypoints; the map points of pair p are those keypoints
    # (frame p-1) un-projected through the KannalaBrandt8 model at their position in frame p (synth.make_last_frame_scene).
    scene_dev, scene_host = None, None
    if tumvi and not args.no_match:
        scene_dev, scene_host = [], []
        pp = pipes[0]
        for g in range(G):
            saved, args.no_match = args.no_match, True
            pp.batch(g)
            args.no_match = saved
            torch.cuda.synchronize()
            cnt = pp.d_cnt[1:].cpu().numpy()
            kk = pp.d_kps[1:].cpu().numpy()
            offs = groups[g][1]
            Xw = np.zeros((B, cap, 3), np.float32); Tcw = np.zeros((B, 16), np.float32); Tlw = np.zeros((B, 16), np.float32)
            has = np.zeros((B, cap), np.uint8)
            for p in range(B):
                l = (p - 1) % B
                n = int(cnt[l, 0])
                x, T, Tl = synth.make_last_frame_scene(1, kb8, kk[l, :n, 0], kk[l, :n, 1], (offs[l][0] - offs[p][0], offs[l][1] - offs[p][1]), 7000 + 977 * g + p)
                Xw[p, :n] = x; Tcw[p] = T.reshape(-1); Tlw[p] = Tl.reshape(-1); has[p, :n] = 1
            scene_host.append(dict(cam=kb8, Xw=Xw, Tcw=Tcw, Tlw=Tlw, has=has))
            scene_dev.append({k: torch.from_numpy(v).to(dev) for k, v in (("Xw", Xw), ("Tcw", Tcw), ("Tlw", Tlw), ("has", has))})

    # set-up: every pipeline runs once so that its lazily sized device buffers exist before any step is counted
    for k, pp in enumerate(pipes):
        pp.batch(k % G, scene=scene_dev)
    torch.cuda.synchronize()
    counter = [0]

    def step():
        for _ in range(NB):
            j = counter[0]
            pipes[j % S].batch((j // S + j % S) % G, scene=scene_dev)     # neighbouring pipelines work on different frame sets
            counter[0] += 1

    def start_profiling():
        # HIP events inside liborbhip, on each pipeline's own launch stream, around every kernel of the batches of the timed region (a
        # ring of the 32 most recent event sets per handle)
        for pp in pipes:
            pp.ex.set_profiling(True)
            pp.mt.set_profiling(True)

    dt = shard.timed_steps(step, args.steps, args.warmup, sync=torch.cuda.synchronize, world=world,
                           device=dev if backend == "nccl" else None, before_timed=start_profiling)
    fps = shard.aggregate_fps(frames_per_step, args.steps, world, dt)

    acc, nsamp = {}, 0
    for pp in pipes:
        st = pp.ex.stage_ms()
        if not st:
            continue
        if not args.no_match:
            st.update(pp.mt.stage_ms())
        for k, v in st.items():
            acc[k] = acc.get(k, 0.0) + v
        nsamp += 1
    acc = {k: v / nsamp for k, v in acc.items()}
    # the last timed batch: what the parity tie compares
    j_last = counter[0] - 1
    p_last, g_last = pipes[j_last % S], (j_last // S + j_last % S) % G
    gpu_last = None      # what EVERY pipeline's last batch left in HBM: [(frame set, outputs)], the pipeline of the very last batch first
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_py as O

        def grab(pp):
            o = {"cnt": pp.d_cnt[1:].cpu().numpy(), "desc": pp.d_desc[1:].cpu().numpy(),
                 "kps": pp.d_kps[1:].cpu().numpy().view(np.uint8).reshape(B, cap * 28).view(O.KP_DTYPE).reshape(B, cap)}
            if not args.no_match:
                o["nm"] = pp.d_nm.cpu().numpy()
                o["match"] = (pp.d_slot if tumvi else pp.d_moq).cpu().numpy()
            return o
        gpu_last = [(pp.group, grab(pp)) for pp in [p_last] + [q for q in pipes if q is not p_last and q.group >= 0]]
    cnt = p_last.d_cnt[1:].cpu().numpy()
    n_kp = float(cnt[:, 0].mean())
    nm_mean = float(p_last.d_nm.cpu().numpy().mean()) if not args.no_match else 0.0

    # the same kernels alone on the chip (one pipeline, synchronised between batches): isolates kernel quality from sharing
    pipes[0].ex.set_profiling(True)       # resets the event ring
    pipes[0].mt.set_profiling(True)
    for _ in range(4):
        pipes[0].batch(0, scene=scene_dev)
        torch.cuda.synchronize()
    iso = pipes[0].ex.stage_ms()
    if not args.no_match:
        iso.update(pipes[0].mt.stage_ms())
    for pp in pipes:
        pp.ex.set_profiling(False)
        pp.mt.set_profiling(False)

    # ---- host-fed: frames in pinned host memory, H2D on a copy stream ahead of the pipelines, results back to pinned memory.
    # EVERY rank runs it (on a multi-GPU node this is the figure PCIe / NUMA / host DRAM can bend); bracketed like the timed
    # region (barrier + device sync on both sides), MAX over ranks, aggregate = world * frames / max time.
    host_fed = None
    if not args.no_host_fed:
        h_img = [torch.from_numpy(groups[g][0]).pin_memory() for g in range(G)]      # all G frame sets cycle through the link
        NS = 6                                  # staging buffers: uploads run up to six batches ahead of the pipelines
        stage = [torch.empty_like(d_img[0]) for _ in range(NS)]
        # ONE upload stream and ONE download stream: uploads run back to back at the link's full rate (two uploads in flight would
        # share it and put the pipelines in lockstep: upload, upload, compute, compute ...), PCIe is full duplex
        up_stream, down_stream = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        h_out = [dict(cnt=torch.empty((B, 2), dtype=torch.int32).pin_memory(), kps=torch.empty((B, cap, 7), dtype=torch.float32).pin_memory(),
                      desc=torch.empty((B, cap, 32), dtype=torch.uint8).pin_memory(), moq=torch.empty((B, cap), dtype=torch.int32).pin_memory(),
                      nm=torch.empty((B,), dtype=torch.int32).pin_memory()) for _ in range(S)]
        stage_free, pipe_done = [None] * NS, [None] * S
        up_events = []
        last_group = [0] * S

        def fed_batch(k):
            j, i, g = k % NS, k % S, k % G
            pp = pipes[i]
            last_group[i] = g
            with torch.cuda.stream(up_stream):
                if stage_free[j] is not None:
                    up_stream.wait_event(stage_free[j])     # the batch that read this staging buffer in place has finished
                u0 = torch.cuda.Event(enable_timing=True); u0.record(up_stream)
                stage[j].copy_(h_img[g], non_blocking=True)
                up = torch.cuda.Event(enable_timing=True); up.record(up_stream)
                up_events.append((u0, up))
            pp.stream.wait_event(up)
            if pipe_done[i] is not None:
                pp.stream.wait_event(pipe_done[i])          # the pipeline's previous outputs have been downloaded
            pp.batch(g, images=stage[j], scene=scene_dev)
            stage_free[j] = pp.stream.record_event()
            with torch.cuda.stream(down_stream):
                down_stream.wait_event(stage_free[j])
                h_out[i]["cnt"].copy_(pp.d_cnt[1:], non_blocking=True)
                h_out[i]["kps"].copy_(pp.d_kps[1:], non_blocking=True)
                h_out[i]["desc"].copy_(pp.d_desc[1:], non_blocking=True)
                if not args.no_match:
                    h_out[i]["moq"].copy_(pp.d_slot if tumvi else pp.d_moq, non_blocking=True)
                    h_out[i]["nm"].copy_(pp.d_nm, non_blocking=True)
                pipe_done[i] = down_stream.record_event()

        for k in range(2 * NS):
            fed_batch(k)
        torch.cuda.synchronize()
        nfed = 48
        up_events.clear()
        counter_k = [0]

        def fed_step():
            for _ in range(nfed):
                fed_batch(counter_k[0])
                counter_k[0] += 1

        tf = shard.timed_steps(fed_step, 1, 0, sync=torch.cuda.synchronize, world=world, device=dev if backend == "nccl" else None)
        up_b = H * W
        down_b = 8 + cap * (28 + 32) + (0 if args.no_match else cap * 4 + 4)
        ffps = world * nfed * B / tf
        up_ms = float(np.mean([a.elapsed_time(b) for a, b in up_events]))
        # what came back over the link against a resident run of the same frame set on the same pipeline: counts, keypoints,
        # descriptors and match indices of every frame
        same = True
        for i, pp in enumerate(pipes):
            pp.batch(last_group[i], scene=scene_dev)
            torch.cuda.synchronize()
            cnt_d = pp.d_cnt[1:].cpu()
            same = same and bool(torch.equal(h_out[i]["cnt"], cnt_d))
            nn = cnt_d[:, 0].clamp(max=cap)
            live = (torch.arange(cap)[None, :] < nn[:, None])
            same = same and bool(torch.equal(h_out[i]["kps"].view(torch.int32)[live], pp.d_kps[1:].cpu().view(torch.int32)[live]))
            same = same and bool(torch.equal(h_out[i]["desc"][live], pp.d_desc[1:].cpu()[live]))
            if not args.no_match:
                same = same and bool(torch.equal(h_out[i]["nm"], pp.d_nm.cpu()))
                if tumvi:
                    same = same and bool(torch.equal(h_out[i]["moq"][live], pp.d_slot.cpu()[live]))
                else:     # match_of_query is indexed by the PREVIOUS frame's keypoints (frame p - 1, wrapping)
                    nq = torch.roll(nn, 1)
                    liveq = (torch.arange(cap)[None, :] < nq[:, None])
                    same = same and bool(torch.equal(h_out[i]["moq"][liveq], pp.d_moq.cpu()[liveq]))
        if world > 1:
            import torch.distributed as dist
            tsame = torch.tensor([1 if same else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tsame, op=dist.ReduceOp.MIN)
            same = bool(tsame.item())
        host_fed = {"value": round(ffps, 2), "unit": "frames/s", "n_gpus": world, "batches_per_rank": nfed, "distinct_frame_sets": G, "staging_buffers": NS, "pipelines": S,
                    "pcie_bytes_per_frame": {"h2d": up_b, "d2h": down_b}, "pcie_GBps_per_gpu": round(ffps / world * (up_b + down_b) / 1e9, 2),
                    "h2d_GBps_per_gpu": round(ffps / world * up_b / 1e9, 2), "upload_ms_per_batch": round(up_ms, 3),
                    "upload_GBps_while_copying": round(B * up_b / (up_ms * 1e-3) / 1e9, 2), "pcie_spec_GBps": PCIE_SPEC_GBS, "outputs_equal_resident_run": same,
                    "note": "every rank: pinned host frames (all %d frame sets in turn) -> HBM on one upload stream into %d staging buffers, %d pipelines; counts, keypoints, "
                            "descriptors, match indices back to pinned memory on a download stream; MAX time over ranks; outputs compared with a resident run, every frame" % (G, NS, S)}
        if not same:
            sys.stderr.write("bench.py: rank %d: host-fed outputs differ from the resident run\n" % rank)

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
    total_bytes = sum(stage_bytes[k] for k in ("pyramid", "fast", "blur", "describe")) + (0 if args.no_match else stage_bytes["match_scan"])
    traffic = None
    tf_ = os.path.join(ROOT, "profiles", "traffic.json")              # measured separately with rocprofv3 --pmc (tools/collect_traffic.py)
    if os.path.exists(tf_) and B == 256 and not tumvi:
        try:
            tj = json.load(open(tf_))
            hit = [v for k, v in tj.items() if k.startswith(kname)]
            if hit:
                traffic = round(hit[0]["hbm_B"])
        except Exception:
            traffic = None

    # VALU issue roofline of the whole batch: wave-instructions per launch from the committed SQ_INSTS_VALU profile (a separate
    # rocprofv3 --pmc run of this workload, tools/collect_sq.py); the ceiling is MEASURED per opcode class by tools/calib/orb_calib.h
    # (profiles/valu_calib.json) and weighted with every kernel's own opcode mix (tools/valu_mix.py -> profiles/valu_mix.json).
    valu = None
    sfp, mixp = os.path.join(ROOT, "profiles", "sq_counters.json"), os.path.join(ROOT, "profiles", "valu_mix.json")
    if os.path.exists(sfp) and os.path.exists(mixp) and B == 256 and not args.no_match and not tumvi:
        try:
            sj, mj = json.load(open(sfp)), json.load(open(mixp))
            tot, tot_t = 0.0, 0.0
            for stage, (kn, launches) in kernels.items():
                hit = [v for k, v in sj.items() if kn in k and "SQ_INSTS_VALU" in v]
                n_i = max(h["SQ_INSTS_VALU"] for h in hit) * launches if hit else 0.0
                ceil = mj["kernels"].get(kn, {}).get("ceiling_G_wave_instr_per_s")
                if n_i and ceil:
                    tot += n_i
                    tot_t += n_i / (ceil * 1e9)
            peak = tot / tot_t
            ach_i = tot / (dt / (args.steps * NB))
            valu = {"bound": "valu-issue", "achieved": round(ach_i / 1e9, 2), "peak": round(peak / 1e9, 2), "unit": "G wave-instr/s",
                    "frac": round(ach_i / peak, 4), "wave_instructions_per_batch": round(tot),
                    "source": "instructions: profiles/sq_counters.json (SQ_INSTS_VALU per launch, separate --pmc run); ceiling: measured per opcode class "
                              "(profiles/valu_calib.json), weighted by each kernel's static opcode mix (profiles/valu_mix.json); time measured live"}
        except Exception:
            valu = None

    rc = 0
    if rank == 0:
        out = {
            "metric": conf["metric"] if not args.no_match else conf["metric"].replace("extract+match", "extract"),
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": conf["workload"], "image": "%dx%d" % (W, H), "nfeatures": CFG["nfeatures"], "nlevels": CFG["nlevels"],
                       "frames_per_step_per_gpu": frames_per_step, "batch": B, "batches_per_step": NB, "distinct_frames_per_gpu": G * B, "streams": S,
                       "mean_keypoints_per_frame": round(n_kp, 1), "mean_matches_per_frame": round(nm_mean, 1),
                       "frame_geometry": ("EuRoC.yaml:9-17 calibration: keypoints undistorted per batch on the device, grid bounds = undistorted corners "
                                          "(%.2f, %.2f, %.2f, %.2f)" % bounds) if distort else "no distortion: bounds = image rectangle",
                       "sharding": "frames round-robin, one process per GPU, no collective"},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": round(launch_bytes), "avg_launch_ms": round(per_launch[dom], 4),
                         "stage_ms_per_batch": {k: round(v, 4) for k, v in acc.items()},
                         "isolated": {"note": "same kernel with nothing else on the chip (1 stream)",
                                      "avg_launch_ms": round(iso[dom] / nl, 4),
                                      "achieved": round(launch_bytes / (iso[dom] / nl * 1e-3) / 1e9, 2),
                                      "frac": round(launch_bytes / (iso[dom] / nl * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                      "stage_ms_per_batch": {k: round(v, 4) for k, v in iso.items()}},
                         "pipeline_algorithmic_GBps": round(fps / world * total_bytes / 1e9, 2)},
        }
        if valu is not None and world == 1:
            out["roofline"]["valu_issue"] = valu
        if host_fed is not None:
            out["host_fed"] = host_fed
            if not host_fed["outputs_equal_resident_run"]:
                rc = 4
        if gpu_last is not None:
            nthreads = host_cores()
            cb, verified, bad = cpu_legs(conf, groups, g_last, gpu_last, scene_host, cap, nthreads, distort=(EUROC_K, EUROC_D) if distort else None)
            out["cpu_baseline"] = cb
            out["verified_frames"] = verified
            out["verified_of"] = B * len(gpu_last)        # the last timed batch of every pipeline
            if bad:
                out["verify_mismatch_frames"] = bad[:16]
                rc = 3
        emit(out)
        if rc:
            sys.stderr.write("bench.py: the GPU's last timed batch differs from the CPU oracle on %d frames\n" % len(bad))
    shard.finish_distributed()
    sys.exit(rc)


if __name__ == "__main__":
    main()
