"""world_size = 2 tests of the N > 1 path on CPU (gloo): the exchange collectives used by early / mid fusion, the frame
sharding + result merge used by replicas, and that agent-sharded early fusion reproduces single-process pillarisation."""
import os
import socket
import sys

import numpy as np
import pytest
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


def _body_world8(rank, world):
    """the layout the driver's 8-GPU run produces: 6 agents and 4 frames on 8 ranks -- ranks 6, 7 hold no rows, ranks 5 .. 7 encode no agent,
    ranks 4 .. 7 detect on no frame"""
    from pcdet.models.sharded import AgentShardedMidFusion
    from pcdet.utils import common_utils, v2x_exchange as ex
    n = 3 + rank if rank < 6 else 0                               # rows of agent `rank` (agent ids 0 .. 5 dealt agent % world)
    rows = torch.full((n, 4), float(rank))
    rows[:, 0] = torch.arange(n) % 4                              # frame index column
    union, counts = ex.all_gather_v_rows(rows)
    frames = ex.shard_frames(4, world, rank)
    merged = common_utils.merge_results_dist([{'frame': f, 'rank': rank} for f in frames], 4)
    mine = AgentShardedMidFusion.agents_of_rank([0, 1, 2, 3, 4, 5], world, rank)
    stack = torch.full((1, 2, 2, 2, 3), float(rank))              # the fixed-shape map stack every rank contributes, agents or not
    maps = ex.all_gather_maps_async(stack).wait()
    return counts, union[:, 1].tolist(), frames, merged, mine, [float(m.mean()) for m in maps]


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


def test_world_8_layout_with_ranks_that_own_no_agent_and_no_frame():
    """every multi-process test was world 2 with agents AND frames on each rank; the 8-GPU run has neither on some"""
    out = _run('_body_world8', world=8)
    counts = [3, 4, 5, 6, 7, 8, 0, 0]
    for rank, (cnt, col, frames, merged, mine, maps) in enumerate(out):
        assert cnt == counts
        assert col == [float(r) for r in range(6) for _ in range(3 + r)]          # rank-major union, empty ranks contribute nothing
        assert frames == ([rank] if rank < 4 else [])
        assert maps == [float(r) for r in range(8)]
        if rank == 0:
            assert [(d['frame'], d['rank']) for d in merged] == [(f, f) for f in range(4)]      # dataset order, one frame per rank 0 .. 3
        else:
            assert merged is None
    assert [o[4] for o in out] == [[0], [2], [3], [4], [5], [], [], []]          # remote agents round-robin; the ego (1) is nobody's


@pytest.mark.parametrize('shard', ['frame', 'agent'])
def test_bench_gpus_8_dry_run(shard):
    """`python bench.py --gpus 8` (the driver's scaling run) starts eight ranks itself; with --shard agent the dry run also walks the
    agent % world row split and the ragged gather with the two ranks that get no agent"""
    r, line = _bench(['--gpus', '8', '--steps', '2', '--warmup', '1', '--shard', shard] + (['--config', 'disco'] if shard == 'agent' else []),
                     {'PCP_BENCH_DRY_RUN': '1'}, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    assert line['n_gpus'] == 8 and line['ranks_seen_by_collective'] == 8 and 'DRY RUN' in line['data']
    assert line['ms_per_step'] >= 2 * 8 * 0.9                                    # the slowest rank sleeps 2 ms x 8 per step: MAX over ranks
    # every rank confined itself to its own slice of the host's CPUs (in-process sched_setaffinity from LOCAL_RANK, before any GPU call)
    masks = line['config']['rank_cpu_affinity']
    assert len(masks) == 8 and all(masks)
    if len(os.sched_getaffinity(0)) >= 8:
        cpu_sets = []
        for m in masks:
            cpus = set()
            for part in m.split(','):
                a, _, b = part.partition('-')
                cpus.update(range(int(a), int(b or a) + 1))
            cpu_sets.append(cpus)
        assert all(not (cpu_sets[i] & cpu_sets[j]) for i in range(8) for j in range(i + 1, 8)), masks
        assert len({len(c) for c in cpu_sets}) == 1
    if shard == 'agent':
        assert line['config']['rows_per_rank'] == [4, 4, 4, 4, 4, 4, 0, 0] and line['config']['frames_per_rank'] == [1, 1, 1, 1, 0, 0, 0, 0]


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


def _body_overlapped_reduce(rank, world):
    """the two-bucket reduction against the single all-reduce with DIFFERENT gradients per rank -- one backward per
    step, two backward passes before a step (gradient accumulation), a step that is skipped (zero_grad in between), and two optimizers in
    one process whose hooks must not fire each other's reduction."""
    sys.path.insert(0, os.path.join(PKG, 'tools'))
    from train_utils.optimization import OverlappedFlatReduce
    from pcdet.models.detectors.centerpoint import run_tape
    from pcp_amd import synth
    n, tail = 1000, 640
    grad = lambda step, part: torch.from_numpy(synth.uniform(5, 100 * step + 10 * part + rank, n, -1, 1))      # rank-dependent
    want = lambda step, parts: sum(torch.from_numpy(synth.uniform(5, 100 * step + 10 * p_ + r, n, -1, 1)) for p_ in parts for r in range(world))
    out = {}
    flat = torch.zeros(n)
    red = OverlappedFlatReduce(flat, tail)
    # (1) one backward per step: the tail is ready first (and reduced while the "backbone" still writes the head)
    flat.zero_()
    g = grad(1, 0)
    flat[tail:] += g[tail:]
    red.grad_ready()
    flat[:tail] += g[:tail]
    scale = red.finish()
    out['one_backward'] = bool(torch.allclose(flat, want(1, [0]), atol=1e-6)) and scale == 1.0 / world and red.started == 1
    # (2) gradient accumulation: a second backward adds to the tail AFTER its first reduction was issued
    flat.zero_()
    for part in (0, 1):
        g = grad(2, part)
        flat[tail:] += g[tail:]
        red.grad_ready()
        flat[:tail] += g[:tail]
    red.finish()
    out['two_backwards'] = bool(torch.allclose(flat, want(2, [0, 1]), atol=1e-6)) and red.started == 2
    # (3) a skipped step: backward, zero_grad (abandon), backward, step
    flat.zero_()
    flat += grad(3, 0)
    red.grad_ready()
    red.abandon()
    flat.zero_()
    flat += grad(3, 1)
    red.grad_ready()
    red.finish()
    out['skipped_step'] = bool(torch.allclose(flat, want(3, [1]), atol=1e-6))
    # (4) the hook travels with the loss node of ITS model: running model B's tape fires B's reducer only
    fa, fb = torch.zeros(n), torch.zeros(n)
    ra, rb = OverlappedFlatReduce(fa, tail), OverlappedFlatReduce(fb, tail)
    fired = []
    hook_a = lambda name: (fired.append(('a', name)), ra.grad_ready())[0] if name == 'dense_head' else None
    hook_b = lambda name: (fired.append(('b', name)), rb.grad_ready())[0] if name == 'dense_head' else None
    fb += grad(4, 0)
    run_tape([('vfe', lambda g_: None), ('dense_head', lambda g_: None)], hook_b)
    out['other_models_tape_leaves_this_one_alone'] = ra.work is None and rb.work is not None and fired == [('b', 'dense_head')]
    rb.finish()
    fa += grad(4, 1)
    run_tape([('vfe', lambda g_: None), ('dense_head', lambda g_: None)], hook_a)
    ra.finish()
    out['both_correct'] = bool(torch.allclose(fb, want(4, [0]), atol=1e-6) and torch.allclose(fa, want(4, [1]), atol=1e-6))
    return out


def test_overlapped_two_bucket_reduce_equals_the_single_all_reduce():
    for out in _run('_body_overlapped_reduce'):
        assert all(out.values()), out


# ---- bench.py --gpus N: the launcher itself ------------------------------------------------------------------------------

def _bench(args, env_extra, timeout=240):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` outside any launcher must start two ranks (the flag was parsed and ignored).  Dry-run mode
    keeps the launcher / rendezvous / barrier / max-over-ranks aggregation and swaps the kernels for a sleep (gloo, no GPU)."""
    r, line = _bench(['--gpus', '2', '--steps', '3', '--warmup', '1'], {'PCP_BENCH_DRY_RUN': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    assert line['n_gpus'] == 2 and line['ranks_seen_by_collective'] == 2
    assert line['steps'] == 3 and line['warmup'] == 1 and 'DRY RUN' in line['data']
    # rank 1 sleeps twice as long per step: the reported time is the MAX over ranks (>= 3 x 4 ms)
    assert line['ms_per_step'] >= 3.9


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    r, line = _bench(['--gpus', '4', '--steps', '1', '--warmup', '0'],
                     {'PCP_BENCH_DRY_RUN': '1', 'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and line is None and 'WORLD_SIZE=1' in (r.stderr + r.stdout)


def test_bench_under_an_external_launcher_uses_its_world():
    """the driver's own form: python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2"""
    import json
    import subprocess
    env = dict(os.environ, PCP_BENCH_DRY_RUN='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(_free_port()), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                         # rank 0 only
    assert json.loads(lines[0])['n_gpus'] == 2


# ---- rank-0-only checkpointing, cfg.LOCAL_RANK from the launcher, group-relative ranks -------------------------------

class _StubOpt:
    lr = 1e-3

    def __init__(self, model):
        self.opt = torch.optim.SGD(model.parameters(), lr=1e-3)

    def zero_grad(self):
        self.opt.zero_grad()

    def clip_grad_norm(self, max_norm):
        pass

    def step(self):
        self.opt.step()

    def state_dict(self):
        return {'t': 1}


class _StubSched:
    def step(self, it):
        pass


def _body_train_model_ckpt_plain(rank, world):
    import argparse
    sys.path.insert(0, os.path.join(PKG, 'tools'))
    import train as train_tool
    from pcdet.config import EasyDict, cfg
    from train_utils.train_utils import train_model
    os.environ['LOCAL_RANK'] = str(rank)
    os.environ['PCP_DIST_BACKEND'] = 'gloo'
    args = argparse.Namespace(launcher='pytorch', tcp_port=int(os.environ['MASTER_PORT']), local_rank=0)
    dist_train, total = train_tool.init_distributed(args, cfg)
    assert dist_train and total == world and cfg.LOCAL_RANK == rank
    ckpt_dir = os.environ['PCP_TEST_CKPT_DIR']
    model = torch.nn.Linear(4, 2)
    loader = [torch.ones(3, 4) for _ in range(2)]

    def model_func(m, batch):
        loss = m(batch).sum()
        return loss, {'loss_total': float(loss)}, {}
    train_model(model, _StubOpt(model), loader, model_func, _StubSched(), EasyDict(GRAD_NORM_CLIP=10), start_epoch=0, total_epochs=4,
                start_iter=0, rank=cfg.LOCAL_RANK, tb_log=None, ckpt_save_dir=ckpt_dir, max_ckpt_save_num=2)
    dist.barrier()
    cfg.LOCAL_RANK = 0
    return sorted(os.listdir(ckpt_dir))


def test_train_model_saves_checkpoints_on_rank_0_only(tmp_path):
    os.environ['PCP_TEST_CKPT_DIR'] = str(tmp_path / 'ckpt')
    try:
        out = _run('_body_train_model_ckpt_plain')
    finally:
        os.environ.pop('PCP_TEST_CKPT_DIR')
    # 4 epochs, keep 2: the two newest files, complete (loadable), no temp leftovers
    assert out[0] == out[1] == ['checkpoint_epoch_3.pth', 'checkpoint_epoch_4.pth']
    ck = torch.load(str(tmp_path / 'ckpt' / 'checkpoint_epoch_4.pth'), weights_only=False)
    assert ck['epoch'] == 4 and ck['it'] == 8 and set(ck['model_state']) == {'weight', 'bias'}


def test_save_is_skipped_when_a_caller_passes_rank_0_on_every_rank(tmp_path):
    """the round-1 bug shape: rank argument 0 everywhere.  The process group's own rank gates the write as well."""
    os.environ['PCP_TEST_CKPT_DIR'] = str(tmp_path / 'ckpt')
    try:
        out = _run('_body_rank_arg_zero')
    finally:
        os.environ.pop('PCP_TEST_CKPT_DIR')
    assert out == [True, False]


def _body_rank_arg_zero(rank, world):
    sys.path.insert(0, os.path.join(PKG, 'tools'))
    from train_utils.train_utils import _is_main_process
    return _is_main_process(0)


def _body_subgroup_gather(rank, world):
    from pcdet.utils import v2x_exchange as ex
    sub = dist.new_group([1, 2])                                  # group rank 0 = global rank 1
    if rank == 0:
        return None
    m = torch.full((2, 2), float(rank))
    got = ex.gather_maps_to(m, dst=0, group=sub)                  # dst is a rank INSIDE the group
    return None if got is None else [float(g[0, 0]) for g in got]


def test_gather_maps_to_translates_group_ranks():
    out = _run('_body_subgroup_gather', world=3)
    assert out[0] is None and out[1] == [1.0, 2.0] and out[2] is None


def test_reference_optimizer_parameter_order():
    """fastai's OptimWrapper numbers tensors leaf module by leaf module, non-BatchNorm leaves first (fastai_optim.py:16-27,104-122)"""
    sys.path.insert(0, os.path.join(PKG, 'tools'))
    from train_utils.optimization import reference_param_order
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 3, 1), torch.nn.BatchNorm2d(3), torch.nn.Sequential(torch.nn.Linear(3, 1), torch.nn.BatchNorm1d(1)))
    net[2][0].bias.requires_grad_(False)
    params = [p for p in net.parameters() if p.requires_grad]
    names = [n for n, p in net.named_parameters() if p.requires_grad]
    order = reference_param_order(net, params)
    assert [names[i] for i in order] == ['0.weight', '0.bias', '2.0.weight', '1.weight', '1.bias', '2.1.weight', '2.1.bias']
