"""Multi-GPU support for the image-sharded inference path.

Images are independent, so the only communication is ONE broadcast of the prepared weight arena from rank 0
(RCCL over xGMI when the process group backend is "nccl"; "gloo" in the CPU tests), then zero steady-state traffic.
`shard_indices` is the round-robin image -> rank assignment used by drivers that own a list of images.
"""
from __future__ import annotations

import time
from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    return list(range(rank, n_items, world))


def broadcast_weights(model, src: int = 0, bucket_bytes: int = 1 << 30, direct_bytes: int = 32 << 20) -> Tuple[int, float]:
    """Broadcast every prepared weight tensor of `model.w` from `src`.  Tensors of >= `direct_bytes` (the 96 Phi / Swin matrices that
    make up ~95 % of the arena: 117 MB per fused Phi matrix in split-f16 form) are broadcast IN PLACE -- already large messages, no
    staging copy; the many small ones (biases, norms, tables, scales) are packed per dtype into flat buckets of up to `bucket_bytes`
    (few, large messages: xGMI links are per-link bound, so message size >> latency * bandwidth).  Returns (bytes moved, seconds)."""
    t0 = time.perf_counter()
    total = 0
    by_dtype = {}
    W = {}                                   # flat name -> tensor view of the prepared weights (split-f16 weights = two tensors)
    for k, v in model.w.items():
        if hasattr(v, "inv_scale"):          # hip_ops.SplitF16 (precision="f16x3")
            W[k + "#f16"], W[k + "#inv_scale"] = v.t, v.inv_scale
        else:
            W[k] = v

    class _M:                                # the bucket code below reads `model.w[k]`
        w = W
    model = _M
    for k in sorted(model.w):
        t = model.w[k]
        if t.numel() * t.element_size() >= direct_bytes and t.is_contiguous():
            dist.broadcast(t, src=src)                       # in place: no torch.cat staging copy, no copy back
            total += t.numel() * t.element_size()
            continue
        by_dtype.setdefault(t.dtype, []).append(k)
    for dt, keys in by_dtype.items():
        bucket, size = [], 0
        esz = torch.empty((), dtype=dt).element_size()

        def flush():
            nonlocal bucket, size, total
            if not bucket:
                return
            flat = torch.cat([model.w[k].reshape(-1) for k in bucket])
            dist.broadcast(flat, src=src)
            off = 0
            for k in bucket:
                n = model.w[k].numel()
                model.w[k].copy_(flat[off:off + n].view_as(model.w[k]))
                off += n
            total += flat.numel() * esz
            bucket, size = [], 0
        for k in keys:
            n = model.w[k].numel() * esz
            if size + n > bucket_bytes and bucket:
                flush()
            bucket.append(k)
            size += n
        flush()
    if model.w and next(iter(model.w.values())).is_cuda:
        torch.cuda.synchronize()
    return total, time.perf_counter() - t0


def _flat_weights(model):
    W = {}
    for k, v in model.w.items():
        if hasattr(v, "inv_scale"):          # hip_ops.SplitF16 (precision="f16x3")
            W[k + "#f16"], W[k + "#inv_scale"] = v.t, v.inv_scale
        elif torch.is_tensor(v):
            W[k] = v
    return W


def weights_checksum(model) -> int:
    """Integer checksum of every prepared weight tensor's BYTES, position dependent: per tensor  sum_i byte_i * (i mod 65521 + 1)  (int64, in
    16 MiB pieces), weighted by a hash of the tensor's name, modulo 2^61.  Two arenas that differ in one byte, or hold the same bytes in
    another order (a row permutation, a transposed layout), get different sums up to a ~2^-60 collision; not a cryptographic hash."""
    total = 0
    for k, t in sorted(_flat_weights(model).items()):
        b = t.contiguous().reshape(-1).view(torch.uint8)            # reshape first: 0-dim tensors and odd last dimensions view cleanly
        acc = 0
        for off in range(0, b.numel(), 1 << 24):
            piece = b[off:off + (1 << 24)].to(torch.int64)
            idx = (torch.arange(off, off + piece.numel(), device=b.device, dtype=torch.int64) % 65521) + 1
            acc += int((piece * idx).sum().item())
        h = sum((i + 1) * ord(c) for i, c in enumerate(k)) % 1000003 + 1
        total = (total + h * (acc % (1 << 61))) % (1 << 61)
    return total


def check_weights_identical(model, device=None, force: bool = False) -> Tuple[bool, int]:
    """All ranks hold the same weights?  MIN / MAX all-reduce of the per-rank checksum (two 31-bit halves: exact in any backend).
    `force`: run the all-reduces in a one-rank group too (bench.py --force-dist: RCCL dry run on one GPU)."""
    c = weights_checksum(model)
    if not (dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)):
        return True, c
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    halves = torch.tensor([c >> 31, c & ((1 << 31) - 1)], dtype=torch.int64, device=device)
    lo, hi = halves.clone(), halves.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi)), c


def reduce_metrics(values: torch.Tensor) -> torch.Tensor:
    """SUM all-reduce of a small vector (float32 / float64, on the device the process group's backend reduces on), e.g.
    [intersection, union, count] -- the role of AverageMeter.all_reduce in the reference's eval scripts
    (psalm/eval/referring_segmentation.py:58-79)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
    return values
