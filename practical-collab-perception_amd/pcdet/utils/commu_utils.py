"""Process-group helpers with the reference's names (pcdet/utils/commu_utils.py:13-182) on torch.distributed; on ROCm the
'nccl' backend is RCCL over xGMI."""
import pickle

import torch
import torch.distributed as dist


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_main_process():
    return get_rank() == 0


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


def all_gather(data):
    """gather arbitrary picklable data from every rank (list ordered by rank)."""
    world = get_world_size()
    if world == 1:
        return [data]
    out = [None] * world
    dist.all_gather_object(out, data)
    return out


def average_reduce_value(data):
    vals = all_gather(data)
    return sum(vals) / len(vals)


def all_reduce(data, op='sum', average=False):
    world = get_world_size()
    if world == 1:
        return data
    t = data.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 'sum' else dist.ReduceOp.MAX)
    return t / world if average else t
