"""Genome sharding across GPUs: one process per GPU, contiguous genome ranges, and the
single exchange step of the path — a sum all-reduce of the per-variant count tensor (RCCL over xGMI when
the backend is "nccl"; the same code runs under "gloo" in the CPU tests).  PyTorch is plumbing only."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class GenomeShard:
    rank: int
    genome_base: int      # global index of the shard's first genome
    n_genomes: int


def shard_genomes(total_genomes: int, world_size: int, align: int = 4) -> list[GenomeShard]:
    """Contiguous shards whose sizes differ by at most `align`; every base is a multiple of `align` so packed
    bytes (4 genomes each) never straddle two ranks.  Ranks beyond the data get empty shards."""
    if total_genomes < 0 or world_size < 1:
        raise ValueError("bad shard request")
    units = -(-total_genomes // align)                     # ceil: whole `align`-genome units
    per, extra = divmod(units, world_size)
    shards, base = [], 0
    for r in range(world_size):
        n_units = per + (1 if r < extra else 0)
        n = min(n_units * align, max(0, total_genomes - base))
        shards.append(GenomeShard(r, base, n))
        base += n
    assert base == total_genomes
    return shards


def replicate_genomes(genomes_per_rank: int, world_size: int) -> list[GenomeShard]:
    """Weak scaling: every rank holds `genomes_per_rank` genomes of one ever larger population."""
    return [GenomeShard(r, r * genomes_per_rank, genomes_per_rank) for r in range(world_size)]


def allreduce_counts(counts, world_size: int):
    """Sum the [V][4] per-variant count tensor over all genome shards, in place.

    counts holds uint32 bit patterns in an int32 tensor; sums stay below 2^31 for < 2^31 genomes, so the
    signed add is exact.  With one rank this is a no-op (no collective is issued)."""
    if world_size > 1:
        import torch.distributed as dist

        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def allreduce_counts_async(counts, world_size: int):
    """allreduce_counts issued asynchronously: returns a work handle whose wait() makes the CURRENT stream (not the
    host) wait for the sums, or None with one rank.  The caller launches the next batch's sweep before waiting, so
    the exchange over xGMI runs beside it (two count buffers alternate)."""
    if world_size > 1:
        import torch.distributed as dist

        return dist.all_reduce(counts, op=dist.ReduceOp.SUM, async_op=True)
    return None


def gather_by_genome(local_rows, shards: list[GenomeShard], world_size: int):
    """Per-genome results need no reduction, only concatenation in shard order (all_gather of ragged rows)."""
    if world_size == 1:
        return local_rows
    import torch
    import torch.distributed as dist

    sizes = [s.n_genomes for s in shards]
    width = local_rows.shape[1:]
    padded = torch.zeros((max(sizes),) + tuple(width), dtype=local_rows.dtype, device=local_rows.device)
    padded[: local_rows.shape[0]] = local_rows
    out = [torch.empty_like(padded) for _ in range(world_size)]
    dist.all_gather(out, padded)
    return torch.cat([o[:n] for o, n in zip(out, sizes)], dim=0)
