"""world_size = 2 tests of the N > 1 path on CPU (gloo): the exchange collectives used by early / mid fusion, the frame
sharding + result merge used by replicas, and that agent-sharded early fusion reproduces single-process pillarisation."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
PKG = os.path.join(REPO, 'practical-collab-perception_amd')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn_name, ret):
    for p in (REPO, PKG, os.path.join(REPO, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ret[rank] = globals()[fn_name](rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn_name, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn_name, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


# ---- bodies executed inside the ranks ------------------------------------------------------------------------------------

def _body_gather_rows(rank, world):
    from pcdet.utils import v2x_exchange as ex
    n = [5, 0][rank]                                              # ragged, one rank empty
    rows = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3) + 100 * rank
    out, counts = ex.all_gather_v_rows(rows)
    return out.numpy().tolist(), counts


def _body_early_fusion(rank, world):
    from pcdet.utils import v2x_exchange as ex
    from pcp_amd import synth
    from oracle import pillars as opil
    agents = [[0, 2, 4], [1, 3, 5]][rank]                         # three agents per rank
    mine = np.concatenate([synth.agent_cloud(a, 4000, 'early') for a in agents], 0)
    mine = np.concatenate([np.zeros((mine.shape[0], 1), np.float32), mine], 1)
    union, counts = ex.all_gather_v_rows(torch.from_numpy(mine))
    vox = opil.voxelize(union.numpy(), 5, [-51.2, -51.2, -8, 51.2, 51.2, 0], [0.2, 0.2, 8.0], [512, 512, 1])
    return vox['coords'].tobytes(), np.sort(vox['cnt']).tobytes(), counts


def _body_maps(rank, world):
    from pcdet.utils import v2x_exchange as ex
    m = torch.full((2, 4, 4, 8), float(rank + 1))
    allm = ex.all_gather_maps(m)
    got = ex.gather_maps_to(m, dst=1)
    return [float(x.mean()) for x in allm], None if got is None else [float(x.mean()) for x in got]


def _body_merge(rank, world):
    from pcdet.utils import common_utils, v2x_exchange as ex
    frames = ex.shard_frames(7, world, rank)
    part = [{'frame': f} for f in frames]
    merged = common_utils.merge_results_dist(part, 7)
    return frames, merged


# ---- tests ---------------------------------------------------------------------------------------------------------------

def test_all_gather_v_rows_ragged():
    r0, r1 = _run('_body_gather_rows')
    assert r0 == r1
    rows, counts = r0
    assert counts == [5, 0] and len(rows) == 5 and rows[0] == [0.0, 1.0, 2.0]


def test_agent_sharded_early_fusion_equals_single_process():
    from pcp_amd import synth
    from oracle import pillars as opil
    r0, r1 = _run('_body_early_fusion')
    assert r0[0] == r1[0] and r0[2] == [12000, 12000]
    # single process, agents in the gathered order (rank 0's agents first): identical pillar set, bit for bit
    order = [0, 2, 4, 1, 3, 5]
    pts = np.concatenate([synth.agent_cloud(a, 4000, 'early') for a in order], 0)
    pts = np.concatenate([np.zeros((pts.shape[0], 1), np.float32), pts], 1)
    vox = opil.voxelize(pts, 5, [-51.2, -51.2, -8, 51.2, 51.2, 0], [0.2, 0.2, 8.0], [512, 512, 1])
    assert vox['coords'].tobytes() == r0[0]
    # and the pillar set does not depend on the agent order at all
    pts2 = np.concatenate([synth.agent_cloud(a, 4000, 'early') for a in range(6)], 0)
    pts2 = np.concatenate([np.zeros((pts2.shape[0], 1), np.float32), pts2], 1)
    vox2 = opil.voxelize(pts2, 5, [-51.2, -51.2, -8, 51.2, 51.2, 0], [0.2, 0.2, 8.0], [512, 512, 1])
    assert vox2['coords'].tobytes() == r0[0] and np.sort(vox2['cnt']).tobytes() == r0[1]


def test_map_gathers():
    r0, r1 = _run('_body_maps')
    assert r0[0] == [1.0, 2.0] and r1[0] == [1.0, 2.0]
    assert r0[1] is None and r1[1] == [1.0, 2.0]


def test_frame_sharding_and_result_merge_keep_dataset_order():
    r0, r1 = _run('_body_merge')
    assert r0[0] == [0, 2, 4, 6] and r1[0] == [1, 3, 5]
    assert [d['frame'] for d in r0[1]] == list(range(7))
    assert r1[1] is None


def _body_flat_gradient_allreduce(rank, world):
    """data-parallel training contract: every rank ends an iteration with identical parameters, equal to a single process that saw
    the mean gradient (the fused optimizer folds 1 / world into its grad_scale)"""
    sys.path.insert(0, os.path.join(PKG, 'tools'))
    from train_utils.optimization import all_reduce_flat_gradient
    from oracle import train as otr
    from pcp_amd import synth
    n = 1000
    p0 = torch.from_numpy(synth.uniform(77, 1, n, -1, 1))
    grads = [torch.from_numpy(synth.uniform(77, 10 + r, n, -1, 1)) for r in range(world)]
    flat = grads[rank].clone()
    scale = all_reduce_flat_gradient(flat)
    st = {'w': p0.clone()}
    opt = otr.AdamOneCycle(['w'])
    opt.step(st, {'w': flat * scale}, 1e-3, 0.95)
    ref = {'w': p0.clone()}
    opt2 = otr.AdamOneCycle(['w'])
    opt2.step(ref, {'w': sum(grads) / world}, 1e-3, 0.95)
    return float((st['w'] - ref['w']).abs().max()), float(scale), st['w'].numpy().tolist()[:8]


def test_flat_gradient_allreduce_gives_identical_replicas():
    out = _run('_body_flat_gradient_allreduce')
    assert out[0][1] == 0.5 and out[1][1] == 0.5
    assert out[0][0] < 1e-7 and out[1][0] < 1e-7
    assert out[0][2] == out[1][2]
