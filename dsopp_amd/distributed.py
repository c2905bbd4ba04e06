"""Multi-GPU glue: one process per GPU, landmarks sharded across ranks, frames / images replicated.

The hot path shards naturally by landmark: every landmark's residuals touch only its host frame and the
target frames, and landmarks interact only through the K x K sums.  Each rank therefore uploads only its landmark shard
and the C-ABI calls back (dsopp_hip_window_set_allreduce) whenever partial sums over landmarks must be combined:
   * once per linearisation: [H_pp | b_pp | H_schur | b_schur]  (2 K^2 + 2 K doubles: 51 KB at F = 7),
   * once per energy evaluation: (energy, n_valid, |idepth step|^2, idepth . step),
   * once per solve: the energy lists for the global 3rd-quartile outlier threshold.
The collective itself is torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, rank: int, world: int):
    """contiguous block partition of n items, balanced to +-1 (rank r gets [lo, hi))"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_window(win, rank: int, world: int):
    """keep this rank's slice of every frame's landmarks (in place); returns the window"""
    for f in win.frames:
        lo, hi = shard_bounds(len(f.uv), rank, world)
        f.uv, f.idepth_gt, f.idepth_init, f.patch = f.uv[lo:hi], f.idepth_gt[lo:hi], f.idepth_init[lo:hi], f.patch[lo:hi]
    return win


class _DeviceBuffer:
    """exposes a raw device pointer to torch through __cuda_array_interface__"""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (int(ptr), False), "version": 3}


def make_device_allreduce(dist, torch, stream, device_index: int):
    """callback(ptr, count, stream_ptr) for HipWindow.set_allreduce: in-place sum across ranks, ordered on `stream`.
    The callback runs between kernel launches of the solve loop, so its host cost is on the critical path when it exceeds
    the GPU time of an iteration: `stream` is made the thread's current stream ONCE here (no per-call context manager) and
    the wrapped tensors are cached per (pointer, count)."""
    cache = {}
    torch.cuda.set_stream(stream)
    all_reduce = dist.all_reduce
    as_tensor = torch.as_tensor
    dev = f"cuda:{device_index}"
    expected_stream = int(stream.cuda_stream)

    def allreduce(ptr, count, stream_ptr):
        # the collective is ordered on torch's current stream: it must be the stream the library launches on
        if stream_ptr and int(stream_ptr) != expected_stream:
            return -1
        t = cache.get((ptr, count))
        if t is None:
            t = as_tensor(_DeviceBuffer(ptr, count), device=dev)
            cache[(ptr, count)] = t
        all_reduce(t)
        return 0

    return allreduce


def allreduce_numpy(dist, torch, arr: np.ndarray) -> np.ndarray:
    """sum a host array across ranks (used by the CPU / gloo tests and for host-side statistics)"""
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).copy())
    dist.all_reduce(t)
    return t.numpy()


def gather_variable(dist, torch, values: np.ndarray, rank: int, world: int) -> np.ndarray:
    """all-gather of variable-length float64 lists with two sum-collectives (counts, then zero-padded slices) — the scheme
    the C-ABI uses for the global 3rd-quartile threshold of updatePointStatuses"""
    counts = np.zeros(world)
    counts[rank] = len(values)
    counts = allreduce_numpy(dist, torch, counts).astype(np.int64)
    total, offset = int(counts.sum()), int(counts[:rank].sum())
    buf = np.zeros(total)
    buf[offset:offset + len(values)] = values
    return allreduce_numpy(dist, torch, buf) if total else buf
