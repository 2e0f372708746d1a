"""Inter-GPU exchange for collaborative perception on one MI355X node (new design: the reference simulates every exchange
through its dataloader / a disk database and has no feature collectives -- SURVEY.md F5, section 8(e)).

One process per GPU; `torch.distributed` backend "nccl" is RCCL over the xGMI mesh (gloo on CPU for tests).  Message sizes are
small (<= 1.9 MB of points per agent, 8.4 MB per compressed BEV map), so the collectives are latency bound: a single padded
all-gather per exchange, no bucketing.

* early fusion (config 4):  exact agent sharding is only possible BEFORE the pillar reductions (mean / max are over the union
  of all agents' points), so the exchanged quantity is the raw ego-frame points: all_gather_v_rows.
* mid fusion  (config 5):  GPU a runs agent a's frozen BEV maker and the shared compressor; the (128, H, W) compressed maps
  are gathered on every rank that hosts an ego for some frame: all_gather_maps (bit-identical to single-device execution
  because the compressor precedes the warp and is applied per map -- v2x_fusion_disco.py:85).
"""
import torch
import torch.distributed as dist


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def all_gather_v_rows(rows, group=None):
    """rows: (n_r, C) tensor, n_r may differ per rank (0 allowed).  Returns (cat over ranks in rank order, list of n_r).
    Two collectives: sizes (world int64) and one padded payload all-gather."""
    world, _rank = _world(group)
    if world == 1:
        return rows, [rows.shape[0]]
    assert rows.dim() == 2
    n_local = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    counts = [int(s.item()) for s in sizes]
    cap = max(max(counts), 1)
    pad = rows.new_zeros((cap, rows.shape[1]))
    pad[:rows.shape[0]] = rows
    recv = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(recv, pad.contiguous(), group=group)
    return torch.cat([recv[r][:counts[r]] for r in range(world)], dim=0), counts


def all_gather_maps(local_map, group=None):
    """local_map: (..., H, W, C) tensor of identical shape on every rank -> list of per-rank maps (rank order)."""
    world, _rank = _world(group)
    if world == 1:
        return [local_map]
    recv = [torch.empty_like(local_map) for _ in range(world)]
    dist.all_gather(recv, local_map.contiguous(), group=group)
    return recv


def gather_maps_to(local_map, dst, group=None):
    """gather on one rank (the ego's GPU); other ranks get None.  On the xGMI mesh the ego pulls from 5 peers over 5
    different links concurrently (RCCL send/recv pairs), ~0.06 ms for 8.4 MB maps at link rate."""
    world, rank = _world(group)
    if world == 1:
        return [local_map]
    if rank == dst:
        out = [torch.empty_like(local_map) for _ in range(world)]
        out[rank] = local_map
        reqs = [dist.irecv(out[r], src=r, group=group) for r in range(world) if r != dst]
        for q in reqs:
            q.wait()
        return out
    dist.send(local_map.contiguous(), dst=dst, group=group)
    return None


def shard_frames(num_frames, world, rank):
    """frame indices rank `rank` detects on when frames are dealt round-robin (DistributedSampler order,
    pcdet/datasets/__init__.py:31-51 of the reference)."""
    return list(range(rank, num_frames, world))
