"""Multi-GPU driver plumbing: frames are independent, so N GPUs = N processes, each with its own
extractor/matcher handle, fed a disjoint share of the frame stream.  There is no data-path collective; the only
communication is the barrier around the timed region and a MAX all-reduce of the elapsed time.  Backend "nccl" (= RCCL)
on GPUs, "gloo" in the CPU tests."""
import os
import time


def frames_for_rank(nframes, rank, world):
    """BASELINE config 4: frame i -> GPU i mod world (round-robin)."""
    return list(range(rank, nframes, world))


def chunk_for_rank(nframes, rank, world):
    """Contiguous chunks (matching frame t against t-1 needs both on one device).  Returns (lo, hi);
    rank r > 0 additionally needs frame lo-1 as the query side of its first pair."""
    base, rem = divmod(nframes, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def init_distributed(backend, device=None):
    """One process per GPU; RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run, or bench.py's own launcher).
    `device`: the rank's torch.device for backend "nccl" (= RCCL), which binds the communicator to it."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl" and device is not None:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def finish_distributed():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def timed_steps(step, steps, warmup, sync=lambda: None, world=1, device=None, before_timed=None):
    """bench.py's timing contract: W untimed warm-up steps, then exactly K steps bracketed by barrier + device sync on
    both sides; returns the MAX elapsed seconds over ranks.  `before_timed` runs once between the warm-up and the first barrier
    (bench.py switches its per-kernel event recording on there).  `device`: where the MAX all-reduce's tensor lives (the rank's
    GPU for "nccl", None = host memory for "gloo")."""
    import torch
    import torch.distributed as dist
    for _ in range(warmup):
        step()
    sync()
    if before_timed is not None:
        before_timed()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def aggregate_fps(frames_per_step_per_rank, steps, world, elapsed_max):
    """Whole-job throughput: every rank processes frames_per_step_per_rank frames per step (weak scaling)."""
    return world * frames_per_step_per_rank * steps / elapsed_max
