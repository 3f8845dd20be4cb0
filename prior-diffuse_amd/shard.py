"""Multi-GPU: utterances are independent units (eval-mode networks are per-sample), so the
path shards by a contiguous batch split with no data-path collective.
One process per GPU; torch.distributed (RCCL on ROCm, gloo in CPU tests) is used only to
gather results / reduce timings."""
import torch


def shard_range(total, world, rank):
    """Contiguous split: rank r owns utterances [lo, hi); remainders go to the first ranks."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def enhance_sharded(enhance_fn, wav, x_T, group=None, gather=True):
    """Run ``enhance_fn(wav_shard, x_T_shard) -> out_shard`` on this rank's utterances of the
    *global* batch (wav [B,L], x_T [B,2,T,161], identical on every rank — x_T is drawn for
    the full batch so results do not depend on the number of ranks) and all-gather the
    waveforms.  Returns the full [B,L] tensor on every rank (or the local shard if
    ``gather`` is False)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    B = wav.shape[0]
    lo, hi = shard_range(B, world, rank)
    out = enhance_fn(wav[lo:hi], x_T[lo:hi]) if hi > lo else wav.new_zeros((0, wav.shape[1]))
    if world == 1 or not gather:
        return out
    return gather_shards(out, B, group)


def gather_shards(out, B, group=None):
    """All-gather the per-rank shards ``out`` ([hi - lo, L] of this rank's contiguous range of a global batch of B) into the full
    [B, L] tensor on every rank.  Ragged shards are padded to the largest shard: one bulk message per peer.  RCCL ("nccl") moves
    the device buffers peer to peer over xGMI; gloo (CPU rehearsals, several ranks sharing one GPU) gathers host copies.  Also
    valid for a group of one rank (the collective then runs on that rank alone)."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(B, world, rank)
    m = (B + world - 1) // world
    pad = out.new_zeros((m, out.shape[1]))
    pad[: hi - lo] = out
    via_host = pad.is_cuda and dist.get_backend(group) != "nccl"
    send = pad.cpu() if via_host else pad
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send, group=group)
    res = []
    for r in range(world):
        a, b = shard_range(B, world, r)
        res.append(parts[r][: b - a])
    full = torch.cat(res, dim=0)
    return full.to(out.device) if via_host else full
