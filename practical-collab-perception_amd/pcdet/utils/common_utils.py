"""Logger / seeding / distributed bootstrap with the reference's names (pcdet/utils/common_utils.py:90-289)."""
import logging
import os
import pickle
import random
import shutil
import subprocess

import numpy as np
import torch
import torch.distributed as dist


def create_logger(log_file=None, rank=0, log_level=logging.INFO):
    logger = logging.getLogger(__name__)
    logger.setLevel(log_level if rank == 0 else 'ERROR')
    fmt = logging.Formatter('%(asctime)s  %(levelname)5s  %(message)s')
    if not logger.handlers:
        console = logging.StreamHandler()
        console.setLevel(log_level if rank == 0 else 'ERROR')
        console.setFormatter(fmt)
        logger.addHandler(console)
        if log_file is not None:
            fh = logging.FileHandler(filename=log_file)
            fh.setLevel(log_level if rank == 0 else 'ERROR')
            fh.setFormatter(fmt)
            logger.addHandler(fh)
    logger.propagate = False
    return logger


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)


def mask_points_by_range(points, limit_range):
    return (points[:, 0] >= limit_range[0]) & (points[:, 0] < limit_range[3]) & (points[:, 1] >= limit_range[1]) & \
           (points[:, 1] < limit_range[4]) & (points[:, 2] >= limit_range[2]) & (points[:, 2] < limit_range[5])


def init_dist_pytorch(tcp_port=None, local_rank=None, backend='nccl'):
    """one process per GPU; reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun)."""
    if local_rank is None:
        local_rank = int(os.environ.get('LOCAL_RANK', 0))
    # one Python host per GPU: confine this rank to its own physical cores BEFORE the first GPU call (in-process, no taskset / numactl hop)
    from pcp_amd.hostcpu import pin_rank_to_cpus
    pin_rank_to_cpus(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1'))))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if not dist.is_initialized():
        if tcp_port is not None and 'MASTER_PORT' not in os.environ:
            os.environ['MASTER_PORT'] = str(tcp_port)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=backend)
    return dist.get_world_size(), dist.get_rank()


def init_dist_slurm(tcp_port, local_rank, backend='nccl'):
    proc_id = int(os.environ['SLURM_PROCID'])
    ntasks = int(os.environ['SLURM_NTASKS'])
    node_list = os.environ['SLURM_NODELIST']
    addr = subprocess.getoutput('scontrol show hostname {} | head -n1'.format(node_list))
    os.environ.update(MASTER_PORT=str(tcp_port), MASTER_ADDR=addr, WORLD_SIZE=str(ntasks), RANK=str(proc_id))
    if torch.cuda.is_available():
        torch.cuda.set_device(proc_id % torch.cuda.device_count())
    dist.init_process_group(backend=backend)
    return dist.get_world_size(), dist.get_rank()


def get_dist_info(return_gpu_per_machine=False):
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    if return_gpu_per_machine:
        return rank, world, max(torch.cuda.device_count(), 1)
    return rank, world


def merge_results_dist(result_part, size, tmpdir=None):
    """gathers per-rank result lists in dataset order (interleaved, like the reference's tmpdir-pickle merge
    common_utils.py:223-244) through all_gather_object instead of the file system."""
    rank, world = get_dist_info()
    if world == 1:
        return result_part[:size]
    parts = [None] * world
    dist.all_gather_object(parts, result_part)
    if rank != 0:
        return None
    # rank-interleaved = dataset order when frames are dealt round-robin (DistributedSampler pads every rank to equal length and the
    # [:size] cut drops its repeats; an unpadded dealing leaves the last ranks one item short, which the index guard covers)
    ordered = [p[i] for i in range(max(len(p) for p in parts)) for p in parts if i < len(p)]
    return ordered[:size]


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
