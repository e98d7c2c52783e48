"""Multi-GPU layout of the hot path: one process per GPU, units sharded, no collective inside
the path.

A unit is (request b, kv head g): its tables, K/V rows and its G query heads are self-contained
(library/lsh/lsh.cc:251-257 index everything by the kv-head group), so ranks own disjoint units:

  mode "batch"  requests are split across ranks (BASELINE cfg 3: B = 64 -> 8 requests per GPU);
  mode "head"   kv heads are split across ranks exactly like the reference's tensor-parallel
                variant (`num_key_value_heads // world_size`, evaluations/RULER/pred/
                attnserver_dist.py:252-254; BASELINE cfg 4: 1 kv head + 8 query heads per GPU).

Collectives exist only at the edges (RCCL through torch.distributed's "nccl" backend on the
GPUs, gloo in the CPU tests): a one-time broadcast of the SimHash hyperplanes so every rank hashes
with the same planes (attnserver_dist.py:279), an optional all_gather of the per-rank outputs for
a single consumer, and the max-over-ranks reduction of the bench's wall time.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass(frozen=True)
class Shard:
    mode: str
    rank: int
    world_size: int
    requests: range      # global request ids owned by this rank
    kv_heads: range      # global kv-head ids owned by this rank
    heads: range         # global query-head ids (per request) owned by this rank

    @property
    def local_batch(self) -> int:
        return len(self.requests)

    @property
    def local_kv_heads(self) -> int:
        return len(self.kv_heads)

    @property
    def local_heads(self) -> int:
        return len(self.heads)


def _block(total: int, parts: int, idx: int) -> range:
    """Contiguous block partition; the first `total % parts` blocks get one extra element."""
    base, extra = divmod(total, parts)
    start = idx * base + min(idx, extra)
    return range(start, start + base + (1 if idx < extra else 0))


def partition(batch_size: int, num_attention_heads: int, num_key_value_heads: int, world_size: int,
              rank: int, mode: str = "batch") -> Shard:
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    if num_attention_heads % num_key_value_heads:
        raise ValueError("num_attention_heads must be a multiple of num_key_value_heads")
    G = num_attention_heads // num_key_value_heads
    if mode == "batch":
        return Shard(mode, rank, world_size, _block(batch_size, world_size, rank),
                     range(num_key_value_heads), range(num_attention_heads))
    if mode == "head":
        if num_key_value_heads % world_size:
            raise ValueError("head sharding needs num_key_value_heads % world_size == 0 "
                             "(attnserver_dist.py:252-254)")
        per = num_key_value_heads // world_size
        kv = range(rank * per, (rank + 1) * per)
        return Shard(mode, rank, world_size, range(batch_size), kv, range(kv.start * G, kv.stop * G))
    raise ValueError(f"unknown sharding mode {mode!r}")


def _dist():
    import torch.distributed as dist

    return dist if (dist.is_available() and dist.is_initialized()) else None


def sync_hash_func(hash_func: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Every rank must hash with the same hyperplanes: broadcast rank `src`'s tensor
    (evaluations/RULER/pred/attnserver_dist.py:279).  bf16 travels as raw bytes (gloo has no bf16)."""
    dist = _dist()
    if dist is None:         # (a one-rank process group still runs the collective: tests/test_gpu_rccl.py)
        return hash_func
    buf = hash_func.contiguous().view(torch.uint8)
    dist.broadcast(buf, src)
    return buf.view(torch.bfloat16)


def gather_outputs(local: torch.Tensor, shard: Shard, batch_size: int, num_attention_heads: int) -> torch.Tensor:
    """all_gather the per-rank attention outputs bf16 [B_loc, H_loc, D] into [B, H, D] on every rank
    (only for a single consumer / checksums; the path itself needs no exchange)."""
    dist = _dist()
    D = local.shape[-1]
    local = local.reshape(shard.local_batch, shard.local_heads, D)
    if dist is None:
        return local
    # the shard describes the process group it was cut for: a one-rank shard inside a larger group owns everything
    # already, and a group of another size cannot be gathered with this shard's block arithmetic
    if shard.world_size != dist.get_world_size():
        if shard.world_size == 1:
            return local
        raise ValueError(f"gather_outputs: shard of {shard.world_size} ranks in a process group of {dist.get_world_size()}")
    # (a ONE-rank group still runs the collective on RCCL -- tests/test_gpu_rccl.py; gloo has no all_gather of
    # device tensors, so there a single rank keeps its block)
    if shard.world_size == 1 and local.is_cuda and dist.get_backend() != "nccl":
        return local
    full = torch.zeros((batch_size, num_attention_heads, D), dtype=local.dtype, device=local.device)
    # ragged batch blocks: gather through a padded buffer of the largest block
    if shard.mode == "batch":
        pad_b = -(-batch_size // shard.world_size)
        mine = torch.zeros((pad_b, num_attention_heads, D * local.element_size()), dtype=torch.uint8,
                           device=local.device)
        mine[:shard.local_batch] = local.contiguous().view(torch.uint8)
        parts = [torch.empty_like(mine) for _ in range(shard.world_size)]
        dist.all_gather(parts, mine)
        for r, p in enumerate(parts):
            blk = _block(batch_size, shard.world_size, r)
            full[blk.start:blk.stop] = p[:len(blk)].view(local.dtype)
    else:
        mine = local.contiguous().view(torch.uint8)
        parts = [torch.empty_like(mine) for _ in range(shard.world_size)]
        dist.all_gather(parts, mine)
        per = num_attention_heads // shard.world_size
        for r, p in enumerate(parts):
            full[:, r * per:(r + 1) * per] = p.view(local.dtype)
    return full


def gather_scalars(value: int, device=None) -> list:
    """One int64 per rank, gathered on every rank (per-rank checksums of the bench / the multi-GPU tests)."""
    dist = _dist()
    if dist is None:
        return [int(value)]
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [int(p.item()) for p in parts]


def max_over_ranks(seconds: float, device=None) -> float:
    """The bench's step time is the slowest rank's."""
    dist = _dist()
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
