"""Agent-sharded execution through the REAL HIP path with two processes on one MI355X (gloo moves the tensors; on a
multi-GPU node the same code runs over RCCL): results must be bit-identical to single-process execution on the union."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
PKG = os.path.join(REPO, 'practical-collab-perception_amd')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup_paths():
    for p in (REPO, PKG, os.path.join(REPO, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)


def _build(g):
    from pcdet.models import build_network_from_meta
    from pcp_amd import synth
    model = build_network_from_meta(g['meta'])
    st = synth.fill_state_dict(g['meta']['state_shapes'])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    return model.cuda().eval()


def _build_conditioned(case):
    """the mini model of tests/golden/g13_conditioned.npz: weights that keep an O(1) spatial signal and a SCORE_THRESH in a wide score gap, so the
    final set is the same for every fp32 implementation -- including this build's own kernels chosen differently by launch size"""
    from helpers import load_golden
    from pcdet.models import build_network_from_meta
    from pcp_amd import synth
    g = load_golden('g13_conditioned.npz')
    meta = g['meta']['cases'][case]
    model = build_network_from_meta(meta)
    model.dense_head.model_cfg.POST_PROCESSING.SCORE_THRESH = float(g[case + '_score_thresh'])
    st = synth.fill_state_dict(meta['state_shapes'], scheme=str(g[case + '_weight_scheme']))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    return model.cuda().eval()


def _disco_inputs(g):
    metadata = [{'se3_from_ego': {0: g['pose_0'], 2: g['pose_2']}}, {'se3_from_ego': {0: g['pose_0']}}]
    return g['points'], metadata


def _worker(rank, world, port, kind, ret):
    _setup_paths()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from helpers import load_golden
        from pcdet.models import sharded
        if kind == 'disco':
            g = load_golden('g1_disco.npz')
            pts, metadata = _disco_inputs(g)
            mine = pts[(pts[:, -1] == 2.0) if rank == 1 else (pts[:, -1] != 2.0)]       # rank 1 owns agent 2, rank 0 agents 0 and 1 (ego)
            runner = sharded.AgentShardedMidFusion(_build(g))
        else:
            g = load_golden('g1_early.npz')
            pts, metadata = g['points'], [{}, {}]
            mine = pts[rank::world]                                                      # any partition of the rows works
            runner = sharded.AgentShardedEarlyFusion(_build(g))
        frames, preds = runner(torch.from_numpy(np.ascontiguousarray(mine)).cuda(), 2, metadata)
        ret[rank] = (frames, [(p['pred_boxes'].cpu().numpy(), p['pred_scores'].cpu().numpy(), p['pred_labels'].cpu().numpy()) for p in preds])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind', ['disco', 'early'])
def test_agent_sharded_equals_single_process(kind):
    _setup_paths()
    from helpers import load_golden
    g = load_golden('g1_%s.npz' % kind)
    model = _build(g)
    if kind == 'disco':
        pts, metadata = _disco_inputs(g)
    else:
        pts, metadata = g['points'], [{}, {}]
    with torch.no_grad():
        single, _ = model({'points': torch.from_numpy(pts).cuda(), 'batch_size': 2, 'metadata': metadata})
    single = [(p['pred_boxes'].cpu().numpy(), p['pred_scores'].cpu().numpy(), p['pred_labels'].cpu().numpy()) for p in single]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), kind, ret), nprocs=2, join=True)
    seen = []
    for rank in range(2):
        frames, preds = ret[rank]
        assert frames == list(range(rank, 2, 2))
        for f, (b, s, l) in zip(frames, preds):
            seen.append(f)
            assert b.shape == single[f][0].shape, (kind, f, b.shape, single[f][0].shape)
            assert np.array_equal(b, single[f][0]) and np.array_equal(s, single[f][1]) and np.array_equal(l, single[f][2])
    assert sorted(seen) == [0, 1]
    assert sum(x[0].shape[0] for x in single) > 0


# ---- the layout of the driver's 8-GPU run: 6 agents and 4 frames on 8 ranks ---------------------------------------------------

def _six_agent_batch(kind, g):
    """mini geometry (the golden's +-6.4 m range), B = 4 frames, 6 agents x 1500 points; DiscoNet: agent 4 is absent from frame 2's
    metadata (its rows stay in the cloud: the ego branch sees every point)"""
    from pcp_amd import synth
    frames = []
    for b in range(4):
        clouds = []
        for a in range(6):
            c = synth.agent_cloud(a + 10 * b, 1500, 'disco' if kind == 'disco' else 'early', xy_half=6.6)
            if kind == 'disco':
                c[:, -1] = a
            clouds.append(c)
        frames.append(np.concatenate(clouds, 0))
    pts = synth.collate(frames)
    agent_of_row = np.tile(np.repeat(np.arange(6), 1500), 4)
    metadata = [{} for _ in range(4)]
    if kind == 'disco':
        for b in range(4):
            poses = {}
            for a in (0, 2, 3, 4, 5):
                if a == 4 and b == 2:
                    continue
                T = np.eye(4)
                yaw = 0.3 * a
                T[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
                T[:3, 3] = [0.3 * a, -0.2 * a, 0.0]
                poses[a] = T
            metadata[b] = {'se3_from_ego': poses}
    return pts, agent_of_row, metadata


def _worker8(rank, world, port, kind, ret, conditioned=False):
    _setup_paths()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from helpers import load_golden
        from pcdet.models import sharded
        g = load_golden('g1_%s.npz' % kind)
        pts, agent_of_row, metadata = _six_agent_batch(kind, g)
        mine = pts[agent_of_row % world == rank]                                          # bench.py --shard agent's split: ranks 6, 7 hold nothing
        runner = (sharded.AgentShardedMidFusion if kind == 'disco' else sharded.AgentShardedEarlyFusion)(_build_conditioned(kind) if conditioned else _build(g))
        frames, preds = runner(torch.from_numpy(np.ascontiguousarray(mine)).cuda(), 4, metadata)
        ret[rank] = (frames, [(p['pred_boxes'].cpu().numpy(), p['pred_scores'].cpu().numpy(), p['pred_labels'].cpu().numpy()) for p in preds])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind', ['disco', 'early'])
def test_agent_sharded_on_8_ranks_equals_single_process(kind, monkeypatch):
    """8 gloo ranks share this GPU: ranks 5 - 7 encode no agent (6 agents, the ego is nobody's), ranks 6 - 7 hold no rows, ranks 4 - 7 detect
    on no frame (B = 4) -- the branches of sharded.py the world-2 tests never enter.  Bitwise the single-process detections.
    The convolution kernel of a layer is chosen by launch size (auto dispatch), and here the ego branch runs one frame per rank against four
    in the single process: under auto dispatch the two agree to fp32 rounding (1e-7 measured), bitwise once the algorithm is pinned -- so
    it is pinned (PCP_CONV_ALGO=direct, inherited by the ranks) and equality is demanded to the bit."""
    monkeypatch.setenv('PCP_CONV_ALGO', 'direct')
    _setup_paths()
    from helpers import load_golden
    g = load_golden('g1_%s.npz' % kind)
    pts, _agents, metadata = _six_agent_batch(kind, g)
    model = _build(g)
    with torch.no_grad():
        single, _ = model({'points': torch.from_numpy(pts).cuda(), 'batch_size': 4, 'metadata': metadata})
    single = [(p['pred_boxes'].cpu().numpy(), p['pred_scores'].cpu().numpy(), p['pred_labels'].cpu().numpy()) for p in single]
    del model
    torch.cuda.empty_cache()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker8, args=(8, _free_port(), kind, ret), nprocs=8, join=True)
    seen = []
    for rank in range(8):
        frames, preds = ret[rank]
        assert frames == ([rank] if rank < 4 else []) and len(preds) == len(frames)
        for f, (b, s, l) in zip(frames, preds):
            seen.append(f)
            assert b.shape == single[f][0].shape, (kind, f, b.shape, single[f][0].shape)
            assert np.array_equal(b, single[f][0]) and np.array_equal(s, single[f][1]) and np.array_equal(l, single[f][2])
    assert sorted(seen) == [0, 1, 2, 3]
    assert sum(x[0].shape[0] for x in single) > 0


@pytest.mark.parametrize('kind', ['disco', 'early'])
def test_agent_sharded_on_8_ranks_under_auto_dispatch_stays_inside_the_tolerance(kind, monkeypatch):
    """the same 8-rank layout WITHOUT pinning the convolution algorithm: the ego branch runs one frame per rank against four
    in the single process, so `auto` may pick another kernel for the same layer -- ranks with unequal local batches must still return the
    single process's detections: same count per frame, every box within 1e-5 and every score within 1e-6 (fp32 rounding of two summation
    orders; the reference tolerance is 1e-3), labels equal.  Well-conditioned weights (g13), so the SET is not a function of that rounding."""
    monkeypatch.delenv('PCP_CONV_ALGO', raising=False)
    _setup_paths()
    from helpers import load_golden, match_boxes
    g = load_golden('g1_%s.npz' % kind)
    pts, _agents, metadata = _six_agent_batch(kind, g)
    model = _build_conditioned(kind)
    with torch.no_grad():
        single, _ = model({'points': torch.from_numpy(pts).cuda(), 'batch_size': 4, 'metadata': metadata})
    single = [(p['pred_boxes'].cpu().numpy(), p['pred_scores'].cpu().numpy(), p['pred_labels'].cpu().numpy()) for p in single]
    del model
    torch.cuda.empty_cache()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker8, args=(8, _free_port(), kind, ret, True), nprocs=8, join=True)
    seen = []
    for rank in range(8):
        frames, preds = ret[rank]
        assert frames == ([rank] if rank < 4 else []) and len(preds) == len(frames)
        for f, (b, s, l) in zip(frames, preds):
            seen.append(f)
            assert b.shape == single[f][0].shape, (kind, f, b.shape, single[f][0].shape)
            n, worst = match_boxes(single[f][0], single[f][1], b, s, tol=1e-5)
            assert n == b.shape[0], (kind, f, n, b.shape[0], worst)
            assert float(np.abs(np.sort(s) - np.sort(single[f][1])).max(initial=0.0)) <= 1e-6
            assert np.array_equal(np.sort(l), np.sort(single[f][2]))
    assert sorted(seen) == [0, 1, 2, 3]
    assert sum(x[0].shape[0] for x in single) > 0


@pytest.mark.parametrize('yaml_name', ['v2x_pointpillar_basic_ego.yaml', 'v2x_pointpillar_disco.yaml'])
def test_train_py_two_ranks_share_one_gpu(tmp_path, yaml_name):
    """tools/train.py under `torch.distributed.run --nproc-per-node 2 ... --launcher pytorch` (the reference's tools/scripts/dist_train.sh)
    through the real HIP path, both ranks on this GPU with gloo carrying the flat-gradient all-reduce (PCP_DIST_BACKEND=gloo; RCCL on a
    multi-GPU node): the run finishes, ONLY rank 0 logs and writes checkpoints, the checkpoint loads, and the loss falls"""
    import re
    import subprocess
    tools = os.path.join(PKG, 'tools')
    env = dict(os.environ, PCP_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
           str(_free_port()), 'train.py', '--launcher', 'pytorch', '--cfg_file', 'cfgs/v2x_sim_models/' + yaml_name, '--batch_size', '2',
           '--epochs', '2', '--output_dir', str(tmp_path), '--set', 'DATA_CONFIG.SYNTHETIC.POINTS_PER_AGENT', '3000',
           'DATA_CONFIG.SYNTHETIC.NUM_FRAMES', '8', 'OPTIMIZATION.LR', '0.003']
    sync_bn = 'disco' in yaml_name                        # the DiscoNet run also exercises --sync_bn (reference tools/train.py:37,128-129)
    if sync_bn:
        cmd.insert(cmd.index('--set'), '--sync_bn')
    r = subprocess.run(cmd, cwd=tools, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2500:]
    out = r.stdout + r.stderr
    assert ('cross-rank BatchNorm statistics: on (2 ranks)' in out) == sync_bn
    losses = [float(m) for m in re.findall(r'loss ([0-9.]+)  lr', out)]
    # --batch_size is the TOTAL batch (reference tools/train.py:86-88): 1 frame per rank and iteration, 4 iterations per epoch, first and
    # last of each epoch logged -- by rank 0 only
    assert len(losses) == 4 and losses[-1] < losses[0], losses
    files = sorted(os.listdir(os.path.join(str(tmp_path), 'ckpt')))
    assert files == ['checkpoint_epoch_1.pth', 'checkpoint_epoch_2.pth'], files
    ck = torch.load(os.path.join(str(tmp_path), 'ckpt', 'checkpoint_epoch_2.pth'), map_location='cpu', weights_only=False)
    assert ck['epoch'] == 2 and ck['it'] == 8
    assert all(torch.isfinite(v).all() for v in ck['model_state'].values() if v.dtype.is_floating_point)


def test_test_py_two_ranks_merge_results_in_dataset_order():
    """tools/test.py --launcher pytorch with two ranks on this GPU (gloo): frames sharded by the DistributedSampler, per-rank detections
    merged on rank 0 (common_utils.merge_results_dist) -- the report covers every frame once"""
    import re
    import subprocess
    tools = os.path.join(PKG, 'tools')
    env = dict(os.environ, PCP_DIST_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
           str(_free_port()), 'test.py', '--launcher', 'pytorch', '--fast', '--cfg_file', 'cfgs/v2x_sim_models/v2x_pointpillar_disco.yaml',
           '--batch_size', '2', '--set', 'DATA_CONFIG.SYNTHETIC.POINTS_PER_AGENT', '3000', 'DATA_CONFIG.SYNTHETIC.NUM_FRAMES', '7']
    r = subprocess.run(cmd, cwd=tools, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2500:]
    out = r.stdout + r.stderr
    reports = re.findall(r'(\d+) detections over (\d+) frames', out)
    assert len(reports) == 1 and int(reports[0][1]) == 7, (reports, out[-1500:])          # one report (rank 0), 7 frames (ragged shard 4 + 3)


def _sync_bn_worker(rank, world, port, ret):
    _setup_paths()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pcp_amd import train_ops as tops
        tops.SYNC_BN = True
        x, dy, gamma, beta = _sync_bn_case()
        rows = [slice(0, 700), slice(700, 1800)][rank]                           # ragged split of the rows over the two ranks
        xs, dys = x[rows].cuda().contiguous(), dy[rows].cuda().contiguous()
        rm, rv = torch.zeros(32).cuda(), torch.ones(32).cuda()
        vec = tops.bn_train_stats(xs, 32, gamma.cuda(), beta.cuda(), 1e-3, 0.01, rm, rv)
        out = torch.empty_like(xs)
        tops.scale_shift_act(xs, 32, vec, True, out)
        dg, db = torch.zeros(32).cuda(), torch.zeros(32).cuda()
        dx = tops.bn_act_backward(dys.clone(), xs, 32, vec, True, dg, db)
        torch.cuda.synchronize()
        ret[rank] = dict(out=out.cpu().numpy(), dx=dx.cpu().numpy(), dg=dg.cpu().numpy(), db=db.cpu().numpy(), rm=rm.cpu().numpy(),
                         rv=rv.cpu().numpy(), mean=vec.mean.cpu().numpy())
    finally:
        dist.destroy_process_group()


def _sync_bn_case():
    g = torch.Generator().manual_seed(41)
    x = torch.randn((1800, 32), generator=g) * 2.0 + 0.5
    x[:700] += 1.5                                                                # the two shards have DIFFERENT statistics
    dy = torch.randn((1800, 32), generator=g)
    return x, dy, torch.rand(32, generator=g) + 0.5, torch.rand(32, generator=g) - 0.5


def test_sync_bn_two_ranks_equal_one_rank_on_the_union():
    """tools/train.py --sync_bn (reference tools/train.py:128-129, nn.SyncBatchNorm): training-mode BatchNorm + ReLU, forward and backward,
    on two ranks holding a ragged split of the rows (all-reduce of the per-channel float64 sums, pcp_bn_*_sums) against ONE rank on all
    rows: activations, running statistics and dx agree to float rounding; the ranks' dgamma / dbeta add up to the single-rank ones"""
    _setup_paths()
    from pcp_amd import train_ops as tops
    assert tops.SYNC_BN is False
    x, dy, gamma, beta = _sync_bn_case()
    xs, dys = x.cuda().contiguous(), dy.cuda().contiguous()
    rm, rv = torch.zeros(32).cuda(), torch.ones(32).cuda()
    vec = tops.bn_train_stats(xs, 32, gamma.cuda(), beta.cuda(), 1e-3, 0.01, rm, rv)
    out = torch.empty_like(xs)
    tops.scale_shift_act(xs, 32, vec, True, out)
    dg, db = torch.zeros(32).cuda(), torch.zeros(32).cuda()
    dx = tops.bn_act_backward(dys.clone(), xs, 32, vec, True, dg, db)
    torch.cuda.synchronize()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sync_bn_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    got_out = np.concatenate([ret[0]['out'], ret[1]['out']], 0)
    got_dx = np.concatenate([ret[0]['dx'], ret[1]['dx']], 0)
    np.testing.assert_allclose(ret[0]['mean'], vec.mean.cpu().numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(got_out, out.cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(got_dx, dx.cpu().numpy(), rtol=0, atol=2e-6)
    for r in (0, 1):                                                              # both ranks track the GLOBAL running statistics
        np.testing.assert_allclose(ret[r]['rm'], rm.cpu().numpy(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(ret[r]['rv'], rv.cpu().numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(ret[0]['dg'] + ret[1]['dg'], dg.cpu().numpy(), rtol=0, atol=2e-4)
    np.testing.assert_allclose(ret[0]['db'] + ret[1]['db'], db.cpu().numpy(), rtol=0, atol=2e-4)
    # and the shards' OWN statistics differ: without the exchange the result would not match
    assert float(np.abs(x[:700].mean(0).numpy() - x.mean(0).numpy()).max()) > 0.5


# ---------------------------------------------------------------------------------------------------------------------
# bench.py --gpus 2 through the REAL kernels: two ranks share this box's one MI355X (PCP_BENCH_BACKEND=gloo), so the
# line is a functional check of the N > 1 path -- launcher, rendezvous, device binding, collectives, max-over-ranks timing, JSON contract
# ---------------------------------------------------------------------------------------------------------------------
def _bench(args, env_extra=None, timeout=900):
    import json
    import subprocess
    env = dict(os.environ, **(env_extra or {}))
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + args, cwd=REPO, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                    # rank 0 prints ONE JSON line
    return json.loads(lines[0])


_BENCH_N1 = {}


def _bench_n1(extra):
    key = tuple(extra)
    if key not in _BENCH_N1:
        _BENCH_N1[key] = _bench(['--gpus', '1', '--steps', '3', '--warmup', '1', '--no-cpu-baseline'] + list(extra))
    return _BENCH_N1[key]


@pytest.mark.parametrize('extra', [[], ['--train'], ['--shard', 'agent'], ['--shard', 'agent', '--config', 'early']],
                         ids=['replicas', 'train', 'agent_disco', 'agent_early'])
def test_bench_two_ranks_through_the_real_kernels(extra):
    line = _bench(['--gpus', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline'] + extra, {'PCP_BENCH_BACKEND': 'gloo'})
    cfg = line['config']
    assert line['n_gpus'] == 2 and cfg['ranks_seen_by_collective'] == 2 and cfg['backend'] == 'gloo'
    assert line['steps'] == 3 and line['warmup'] == 1 and line['cpu_baseline'] is None
    assert 'FUNCTIONAL CHECK ONLY' in line['data']
    assert len(cfg['per_rank_ms_per_step']) == 2 and cfg['rank_devices'] == [0, 0]          # both ranks on this box's one device
    assert max(cfg['per_rank_ms_per_step']) <= line['ms_per_step'] * 1.001 + 1e-3           # the line carries the MAX over ranks
    assert line['scaling'] == ('strong' if '--shard' in extra else 'weak')
    if '--shard' in extra:
        return
    # two replicas time-share one GPU: the aggregate stays near the single-rank rate (a rank that silently did nothing, or did its work
    # twice, would double or halve it)
    one = _bench_n1([e for e in extra])
    assert one['n_gpus'] == 1 and one['config']['rank_devices'] == [0]
    ratio = line['value'] / one['value']
    assert 0.55 < ratio < 1.6, (line['value'], one['value'])
    if '--train' not in extra:
        assert cfg['final_boxes_last_step'] > 0


def test_bench_pipelined_default_and_batch_by_batch_agree(tmp_path):
    """bench.py's default inference mode queues step i+1 before it reads step i's box counts (pcdet/models/pipelined.py); --no-pipeline runs
    batch by batch.  Same workload, same final boxes; the line says which mode it measured; --layer-table writes the per-shape kernel table."""
    table = tmp_path / 'layers.txt'
    piped = _bench(['--gpus', '1', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--layer-table', str(table)])
    plain = _bench(['--gpus', '1', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-pipeline'])
    assert 'software-pipelined' in piped['config']['mode'] and 'software-pipelined' not in plain['config']['mode']
    assert piped['config']['final_boxes_last_step'] == plain['config']['final_boxes_last_step'] > 0
    assert piped['roofline']['kernel'] == plain['roofline']['kernel'] and piped['roofline']['next_mfma_kernels']
    # per-launch times are HIP-event brackets less what an empty event pair reads in the same process; the raw values ride along
    for line in (piped, plain):
        r, h = line['roofline'], line['roofline_hbm']
        assert 0.5 < r['event_bracket_overhead_us'] < 25.0
        assert abs((r['avg_launch_us_uncorrected'] - r['avg_launch_us']) - r['event_bracket_overhead_us']) < 0.05
        assert r['frac_uncorrected'] < r['frac'] < 1.0 and h['frac_uncorrected'] < h['frac'] < 1.0
        assert h['ms_per_step'] < h['ms_per_step_uncorrected']
    rows = table.read_text().splitlines()
    assert rows[0].split()[:2] == ['kernel', 'shape'] and any(r.startswith('k_wino4c') for r in rows[1:])


# ---------------------------------------------------------------------------------------------------------------------
# RCCL itself: a ONE-rank `nccl` process group on this box's MI355X.  PCP_FORCE_COLLECTIVES=1 makes every exchange of
# the N > 1 path issue its collective although one rank could short-cut it: librccl is loaded, the payloads are device tensors, RCCL's
# kernels run on the device.  A one-rank collective is an identity, so every result must equal the path without a process group.
# ---------------------------------------------------------------------------------------------------------------------
_RCCL_CHILD = r'''
import json, os, sys
import numpy as np
import torch
import torch.distributed as dist
REPO, PKG = sys.argv[1], sys.argv[2]
for p in (REPO, PKG, os.path.join(REPO, 'tests'), os.path.join(PKG, 'tools')):
    sys.path.insert(0, p)
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
out = {}
try:
    from helpers import load_golden
    from pcdet.models import build_network_from_meta, sharded
    from pcdet.utils import v2x_exchange as ex
    from pcp_amd import synth
    assert ex.force_collectives()
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(5)
    rows = torch.randn((1234, 6), generator=g).to(dev)
    got, counts = ex.all_gather_v_rows(rows)
    out['rows_equal'] = bool(torch.equal(got, rows)) and counts == [1234] and got.data_ptr() != rows.data_ptr()
    maps = torch.randn((2, 16, 16, 8), generator=g).to(dev)
    rec = ex.all_gather_maps(maps)
    out['maps_equal'] = len(rec) == 1 and bool(torch.equal(rec[0], maps)) and rec[0].data_ptr() != maps.data_ptr()
    pend = ex.all_gather_maps_async(maps)
    busy = torch.ones(1 << 20, device=dev).sum()                      # independent work queued between issue and wait
    rec = pend.wait()
    out['async_equal'] = bool(torch.equal(rec[0], maps)) and float(busy.item()) == float(1 << 20)
    m, f = ex.gather_modar(torch.randn((7, 9), generator=g).to(dev), torch.randn((40, 13), generator=g).to(dev), 0)
    out['modar_shapes'] = [list(m[0].shape), list(f[0].shape)]
    # the flat-gradient all-reduce of a data-parallel training step (tools/train_utils/optimization)
    from train_utils.optimization import all_reduce_flat_gradient
    flat = torch.randn(4_800_000, generator=g).to(dev)                # 19 MB: config 5's gradient buffer
    want = flat.clone()
    scale = all_reduce_flat_gradient(flat)
    out['allreduce_equal'] = scale == 1.0 and bool(torch.equal(flat, want))
    # agent-sharded mid fusion and early fusion through the real kernels, collectives forced
    for kind in ('disco', 'early'):
        gold = load_golden('g1_%s.npz' % kind)
        model = build_network_from_meta(gold['meta'])
        st = synth.fill_state_dict(gold['meta']['state_shapes'])
        model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
        model = model.cuda().eval()
        if kind == 'disco':
            metadata = [{'se3_from_ego': {0: gold['pose_0'], 2: gold['pose_2']}}, {'se3_from_ego': {0: gold['pose_0']}}]
            runner = sharded.AgentShardedMidFusion(model)
        else:
            metadata = [{}, {}]
            runner = sharded.AgentShardedEarlyFusion(model)
        pts = torch.from_numpy(gold['points']).cuda()
        with torch.no_grad():
            single, _ = model({'points': pts.clone(), 'batch_size': 2, 'metadata': metadata})
        frames, preds = runner(pts.clone(), 2, metadata)
        same = frames == [0, 1] and len(preds) == 2
        for a, b in zip(single, preds):
            for k in ('pred_boxes', 'pred_scores', 'pred_labels'):
                same = same and a[k].shape == b[k].shape and bool(torch.equal(a[k], b[k]))
        out[kind + '_equal'] = bool(same)
        out[kind + '_boxes'] = int(sum(p['pred_boxes'].shape[0] for p in single))
    # a training step whose gradient all-reduce runs in two buckets, the head + fusion bucket asynchronously under the backbone's backward
    from pcdet.config import EasyDict
    from train_utils.optimization import build_optimizer
    from pcdet.models.detectors import centerpoint
    gold = load_golden('g7_train.npz')
    grads = {}
    for overlap in (True, False):
        model = build_network_from_meta(gold['meta'])
        st = synth.fill_state_dict(gold['meta']['state_shapes'])
        model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
        model = model.cuda()
        opt = build_optimizer(model, EasyDict(gold['meta']['optimization']))
        if overlap:
            out['overlap_attached'] = opt._tail_off is not None and 0 < opt._tail_off < opt.flat_g.numel()
        else:
            opt.detach_overlap(model)
        model.train()
        opt.zero_grad()
        metadata = [{'se3_from_ego': {0: gold['pose_0'], 2: gold['pose_2']}}, {'se3_from_ego': {0: gold['pose_0']}}]
        ret, tb, _ = model({'points': torch.from_numpy(gold['points']).cuda(), 'batch_size': 2, 'metadata': metadata,
                            'gt_boxes': torch.from_numpy(gold['gt_boxes']).cuda()})
        ret['loss'].backward()
        if overlap:
            out['overlap_in_flight'] = opt._overlap.work is not None and getattr(opt, 'overlapped_reductions', 0) == 1
        opt.clip_grad_norm(10.0)
        g_before = opt.flat_g.clone()
        opt.step()
        torch.cuda.synchronize()
        grads[overlap] = (g_before, opt.flat_p.clone())
    out['overlap_same_update'] = bool(torch.equal(grads[True][1], grads[False][1]))
    torch.cuda.synchronize()
    with open('/proc/self/maps') as fh:
        out['librccl_mapped'] = any('librccl' in ln for ln in fh)
    out['backend'] = dist.get_backend()
finally:
    dist.destroy_process_group()
print('RCCL_RESULT ' + json.dumps(out))
'''


def test_rccl_one_rank_group_runs_every_collective_of_the_sharded_paths_on_device_tensors():
    import json
    import subprocess
    env = dict(os.environ, PCP_FORCE_COLLECTIVES='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', _RCCL_CHILD, REPO, PKG], capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('RCCL_RESULT ')][0][len('RCCL_RESULT '):])
    assert res['backend'] == 'nccl' and res['librccl_mapped'], res
    for k in ('rows_equal', 'maps_equal', 'async_equal', 'allreduce_equal', 'disco_equal', 'early_equal', 'overlap_attached', 'overlap_in_flight',
              'overlap_same_update'):
        assert res[k] is True, (k, res)
    assert res['modar_shapes'] == [[7, 9], [40, 13]]
    assert res['disco_boxes'] > 0 and res['early_boxes'] > 0


def test_rccl_one_rank_training_step_allreduces_the_flat_gradient_on_the_device(tmp_path):
    """tools/train.py --launcher pytorch with ONE rank and the default backend (nccl = RCCL), collectives forced: the optimizer's
    all-reduce runs on the device-resident flat gradient; the run finishes and the loss falls as in the gloo two-rank test"""
    import re
    import subprocess
    tools = os.path.join(PKG, 'tools')
    env = dict(os.environ, PCP_FORCE_COLLECTIVES='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('PCP_DIST_BACKEND', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port',
           str(_free_port()), 'train.py', '--launcher', 'pytorch', '--cfg_file', 'cfgs/v2x_sim_models/v2x_pointpillar_disco.yaml',
           '--batch_size', '2', '--epochs', '2', '--output_dir', str(tmp_path), '--set', 'DATA_CONFIG.SYNTHETIC.POINTS_PER_AGENT', '3000',
           'DATA_CONFIG.SYNTHETIC.NUM_FRAMES', '8', 'OPTIMIZATION.LR', '0.003']
    r = subprocess.run(cmd, cwd=tools, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2500:]
    out = r.stdout + r.stderr
    losses = [float(m) for m in re.findall(r'loss ([0-9.]+)  lr', out)]
    assert len(losses) == 4 and losses[-1] < losses[0], losses
    assert 'backend nccl' in out, out[-1500:]
