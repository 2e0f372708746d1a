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
import os

import torch
import torch.distributed as dist

# PCP_FORCE_COLLECTIVES=1: issue every collective even in a ONE-rank group (default: a single rank short-cuts them).  On a one-GPU box
# this is how the RCCL code path itself -- librccl loaded, device buffers, the library's kernels on the device -- is exercised
# (tests/test_gpu_dist.py::test_rccl_*); results are identical by construction.
def force_collectives():
    return os.environ.get('PCP_FORCE_COLLECTIVES', '0') == '1'


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _single(world):
    return world == 1 and not (force_collectives() and dist.is_available() and dist.is_initialized())


def all_gather_v_rows(rows, group=None):
    """rows: (n_r, C) tensor, n_r may differ per rank (0 allowed).  Returns (cat over ranks in rank order, list of n_r).
    Two collectives: sizes (world int64) and one padded payload all-gather."""
    world, _rank = _world(group)
    if _single(world):
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
    if _single(world):
        return [local_map]
    recv = [torch.empty_like(local_map) for _ in range(world)]
    dist.all_gather(recv, local_map.contiguous(), group=group)
    return recv


class PendingMaps:
    """an all-gather of per-rank maps in flight (RCCL runs it on its own stream); wait() orders the current stream behind it"""

    def __init__(self, recv, work):
        self.recv, self.work = recv, work

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.recv


def all_gather_maps_async(local_map, group=None):
    """all_gather_maps that returns at once: the caller queues independent work (the ego branch up to the fusion module) and calls
    .wait() where the maps are consumed"""
    world, _rank = _world(group)
    if _single(world):
        return PendingMaps([local_map], None)
    recv = [torch.empty_like(local_map) for _ in range(world)]
    return PendingMaps(recv, dist.all_gather(recv, local_map.contiguous(), group=group, async_op=True))


def gather_maps_to(local_map, dst, group=None):
    """gather on one rank (the ego's GPU); other ranks get None.  On the xGMI mesh the ego pulls from 5 peers over 5
    different links concurrently (RCCL send/recv pairs), ~0.06 ms for 8.4 MB maps at link rate."""
    world, rank = _world(group)
    if world == 1:
        return [local_map]
    # `dst` and the loop index are ranks INSIDE `group`; torch.distributed's point-to-point calls take GLOBAL ranks
    glob = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    if rank == dst:
        out = [torch.empty_like(local_map) for _ in range(world)]
        out[rank] = local_map
        reqs = [dist.irecv(out[r], src=glob(r), group=group) for r in range(world) if r != dst]
        for q in reqs:
            q.wait()
        return out
    dist.send(local_map.contiguous(), dst=glob(dst), group=group)
    return None


def shard_frames(num_frames, world, rank):
    """frame indices rank `rank` detects on when frames are dealt round-robin (DistributedSampler order,
    pcdet/datasets/__init__.py:31-51 of the reference)."""
    return list(range(rank, num_frames, world))


# ---- lately fusion (config 3): MoDAR exchange, GPU to GPU ---------------------------------------------------------------------

def ingest_modar(ego_points13, modar, foreground, target_se3_lidar, max_sweep_idx=None):
    """Ego-side ingestion of one remote agent's message (reference: the dataloader does this on the host from a disk database,
    pcdet/datasets/v2x_sim/v2x_sim_dataset_ego.py:196-232): shifts each MoDAR box by twice the mean flow of the foreground points
    inside it, maps it to the ego frame and appends the 13-column rows to the ego cloud.  Everything stays on the device.
    ego_points13: (N, 13) CUDA rows WITHOUT the frame index column [x,y,z,i,t, dx,dy,dz,heading,score,label, sweep_idx, inst_idx];
    modar: (n, 9) CUDA; foreground: (m, 13) CUDA or None; target_se3_lidar: (4, 4) float64 numpy."""
    from pcp_amd import ops
    if max_sweep_idx is None:
        max_sweep_idx = float(ego_points13[:, -2].max().item())
    rows = ops.modar_ingest(modar, foreground, target_se3_lidar, max_sweep_idx)
    return torch.cat([ego_points13, rows], dim=0)


def gather_modar(modar, foreground, dst, group=None):
    """RCCL gather of the (<= 83 x 9) MoDAR rows and the foreground rows of every agent on the ego's rank (README: 0.02 MB per agent):
    two ragged row gathers; returns lists indexed by rank on `dst`, (None, None) elsewhere."""
    world, rank = _world(group)
    if _single(world):
        return [modar], [foreground]
    allm, cm = all_gather_v_rows(modar, group)
    allf, cf = all_gather_v_rows(foreground, group)
    if rank != dst:
        return None, None
    om, of, a, b = [], [], 0, 0
    for r in range(world):
        om.append(allm[a:a + cm[r]])
        of.append(allf[b:b + cf[r]])
        a += cm[r]
        b += cf[r]
    return om, of
