"""Config-5 (DiscoNet) training step on MI355X against the reference's own train loop (tests/golden/g7_train.npz) and the CPU
oracle (oracle/train.py, float64 = exact gradients, float32 = the noise floor of this ill-conditioned fixture)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from helpers import arch_of, load_golden
from pcp_amd import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(g):
    from pcdet.models import build_network_from_meta
    model = build_network_from_meta(g['meta'])
    st = synth.fill_state_dict(g['meta']['state_shapes'])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    return model.to(DEV)


def _batch(g):
    metadata = [{'se3_from_ego': {0: g['pose_0'], 2: g['pose_2']}}, {'se3_from_ego': {0: g['pose_0']}}]
    return {'points': torch.from_numpy(g['points']).to(DEV), 'batch_size': 2, 'metadata': metadata,
            'gt_boxes': torch.from_numpy(g['gt_boxes']).to(DEV)}, metadata


def _sample(t, cap=4096):
    a = t.detach().reshape(-1)
    if a.numel() <= cap:
        return a.cpu().numpy()
    return a[::a.numel() // 1024][:1024].cpu().numpy()


def _oracle_grads(g, metadata, dtype):
    from oracle import train as otr
    meta = g['meta']
    arch = otr.add_train_arch(arch_of(meta), meta['model'])
    st = otr.make_state(synth.fill_state_dict(meta['state_shapes']))
    if dtype == torch.float64:
        st = {k: (v.detach().double().requires_grad_(v.requires_grad) if (v.dtype == torch.float32 and not k.startswith('bev_maker')) else v)
              for k, v in st.items()}
    loss, tb, aux = otr.train_forward(g['points'], g['gt_boxes'], metadata, st, arch)
    loss.backward()
    names = [str(n) for n in g['trainable']]
    return {n: st[n].grad.detach().double() for n in names}, float(loss.detach()), tb, st


def test_disco_train_step_matches_reference_and_oracle():
    sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd', 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from pcdet.config import EasyDict
    g = load_golden('g7_train.npz')
    meta = g['meta']
    names = [str(n) for n in g['trainable']]
    model = _build(g)
    ocfg = EasyDict(meta['optimization'])
    opt = build_optimizer(model, ocfg)
    sched, _ = build_scheduler(opt, meta['total_it_each_epoch'], ocfg.NUM_EPOCHS, -1, ocfg)
    params = dict(model.named_parameters())
    assert set(names) == set(n for n, p in params.items() if p.requires_grad)
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    batch, metadata = _batch(g)
    g64, loss64, _tb64, st64 = _oracle_grads(g, metadata, torch.float64)
    g32, _loss32, _tb32, _ = _oracle_grads(g, metadata, torch.float32)

    for it in range(2):
        sched.step(it)
        assert abs(opt.lr - float(g['it%d_lr' % it])) < 1e-12 and abs(opt.mom - float(g['it%d_mom' % it])) < 1e-12
        model.train()
        opt.zero_grad()
        batch, _ = _batch(g)
        ret, tb, _disp = model(batch)
        loss = ret['loss']
        model.update_global_step()
        loss.backward()
        ref_tb = json.loads(str(g['it%d_tb_json' % it]))
        tol = 2e-5 if it == 0 else 2e-3
        lv = float(loss.detach())
        assert abs(lv - float(g['it%d_loss' % it])) <= tol * abs(float(g['it%d_loss' % it])), (lv, float(g['it%d_loss' % it]))
        for k, v in ref_tb.items():
            assert abs(tb[k] - v) <= max(tol, 2e-4) * abs(v) + 1e-9, (k, tb[k], v)
        if it == 0:
            assert abs(lv - loss64) <= 1e-5 * abs(loss64)
            td = model.dense_head.forward_ret_dict['target_dicts']
            np.testing.assert_allclose(td['heatmaps'][0].cpu().numpy(), g['tgt_heatmap'], atol=1e-6)
            assert np.array_equal(td['inds'][0].cpu().numpy(), g['tgt_inds']) and np.array_equal(td['masks'][0].cpu().numpy(), g['tgt_mask'])
            np.testing.assert_allclose(td['target_boxes'][0].cpu().numpy(), g['tgt_boxes'], atol=2e-6)
            # Gradients vs the float64 oracle (exact).  Tensors downstream of the last BatchNorm+ReLU of the graph (head, last fusion
            # conv) must agree to 1e-4.  Further upstream the comparison is bounded by the fixture, not the kernels: a forward
            # difference of 1e-5 flips the ReLU mask of the few pre-activations that sit within 1e-5 of zero, and with only
            # 2 x 32 x 32 samples per channel one flipped element moves a BatchNorm bias gradient by ~1 %; the float32 CPU oracle
            # shows the same effect (profiles/scripts/debug/dbg_train_grad_bisect.py prints both columns).  There: 3e-2 per tensor, 5e-3 global relative L2.
            num = den = 0.0
            gmax = max(float(v.abs().max()) for v in g64.values())
            for n in names:
                mine = params[n].grad.detach().double().cpu()
                exact = g64[n]
                scale = max(float(exact.abs().max()), 1e-4 * gmax)
                err = float((mine - exact).abs().max())
                floor = float((g32[n] - exact).abs().max())
                tight = n.startswith('dense_head.') or n.startswith('v2x_mid_fusion.decompressor.3')
                assert err <= (max(1e-4 * scale, 3.0 * floor) if tight else 3e-2 * scale), (n, err, scale, floor)
                num += float(((mine - exact) ** 2).sum())
                den += float((exact ** 2).sum())
                ref = g['g0/' + n]
                assert np.abs(_sample(params[n].grad) - ref).max() <= 3e-2 * max(float(np.abs(ref).max()), 1e-4 * gmax), n
            assert num <= (5e-3 ** 2) * den, (num / den) ** 0.5
            print('global relative L2 gradient error vs float64 oracle: %.3e' % ((num / den) ** 0.5))
        opt.clip_grad_norm(ocfg.GRAD_NORM_CLIP)
        opt.step()
        if it == 0:
            assert abs(opt.grad_norm() - float(g['it0_grad_norm'])) <= 2e-4 * float(g['it0_grad_norm'])
            for n in names:
                assert np.abs(_sample(params[n]) - g['p1/' + n]).max() <= 2.1 * opt.lr, n
            sd = model.state_dict()
            for i, k in enumerate(str(k) for k in g['bn_keys']):
                if 'num_batches' in k:
                    continue
                a = sd[k].double()
                d = np.array([float(a.norm()), float(a.sum()), float(a.abs().max())])
                np.testing.assert_allclose(d, g['it0_bn_digest'][i], rtol=2e-4, atol=1e-6, err_msg=k)
    # the trained model still runs the inference path
    model.eval()
    batch, _ = _batch(g)
    batch.pop('gt_boxes')
    with torch.no_grad():
        pred, _ = model(batch)
    assert len(pred) == 2 and all(torch.isfinite(p['pred_boxes']).all() for p in pred)


def test_mixed_precision_bf16_training_step_tracks_the_fp32_reference(monkeypatch):
    """PCP_CONV_ALGO=bf16 (BASELINE.json config 5: "training loop bf16", include/pcp_hip_mp.h): bf16 activation / gradient storage in the
    conv stacks, forward / data-gradient / WEIGHT-gradient 3x3 convs on the bf16 matrix cores -- the frozen teachers' forward too --,
    fp32 master weights, BatchNorm, losses and optimizer.  Not a parity mode: the first-iteration loss of the reference's train loop
    (golden g7) must be reproduced to 1 %, its clipped-gradient norm to 5 %, and two optimizer steps must stay finite."""
    monkeypatch.setenv('PCP_CONV_ALGO', 'bf16')
    sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd', 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from pcdet.config import EasyDict
    from pcp_amd import train_ops as tops
    calls = {'conv': [], 'wgrad': 0}
    orig_c, orig_w = tops.mp_conv3x3, tops.mp_conv3x3_wgrad
    monkeypatch.setattr(tops, 'mp_conv3x3', lambda x, *a, **k: (calls['conv'].append(x.dtype), orig_c(x, *a, **k))[1])
    monkeypatch.setattr(tops, 'mp_conv3x3_wgrad', lambda *a, **k: (calls.__setitem__('wgrad', calls['wgrad'] + 1), orig_w(*a, **k))[1])
    g = load_golden('g7_train.npz')
    meta = g['meta']
    model = _build(g)
    ocfg = EasyDict(meta['optimization'])
    opt = build_optimizer(model, ocfg)
    sched, _ = build_scheduler(opt, meta['total_it_each_epoch'], ocfg.NUM_EPOCHS, -1, ocfg)
    for it in range(2):
        sched.step(it)
        model.train()
        opt.zero_grad()
        batch, _ = _batch(g)
        ret, tb, _disp = model(batch)
        model.update_global_step()
        ret['loss'].backward()
        lv, ref = float(ret['loss'].detach()), float(g['it%d_loss' % it])
        assert np.isfinite(lv) and abs(lv - ref) <= (1e-2 if it == 0 else 5e-2) * abs(ref), (it, lv, ref)
        opt.clip_grad_norm(ocfg.GRAD_NORM_CLIP)
        opt.step()
        if it == 0:
            assert abs(opt.grad_norm() - float(g['it0_grad_norm'])) <= 5e-2 * float(g['it0_grad_norm']), opt.grad_norm()
    # the bf16 kernels ran: every conv launch read bf16 activations, and every 3x3 layer's weight gradient came from the bf16 GEMM
    assert len(calls['conv']) > 40 and all(d == torch.bfloat16 for d in calls['conv']) and calls['wgrad'] >= 2 * 20, calls
    assert all(torch.isfinite(p).all() for p in model.parameters())


def _full_size_batch(g):
    """(points, metadata, batch size) of a full-size training fixture: one frame of agents 0 - 5 (g7_train_full) or, for the b4 fixture, the
    batch bench.py --train times (bench.make_points(CONFIGS['disco'], 4, 0); the fixture's gt_boxes are bench.make_gt_boxes(4, 0))"""
    if 'batch' in g and int(g['batch']) == 4:
        import bench
        pts, metas = bench.make_points(bench.CONFIGS['disco'], 4, 0)
        assert np.array_equal(g['gt_boxes'], bench.make_gt_boxes(4, 0))
        return pts, metas, 4
    agents = (0, 1, 2, 3, 4, 5)
    clouds = []
    for a in agents:
        c = synth.agent_cloud(agent=a, n_points=60000, layout='disco')
        c[:, -1] = float(a)
        clouds.append(c)
    return synth.collate([np.concatenate(clouds, axis=0)]), [{'se3_from_ego': {a: g['pose_%d' % a] for a in agents if a != 1}}], 1


def _full_size_disco(monkeypatch, algo, fixture='g7_train_full.npz'):
    sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd', 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from pcdet.config import EasyDict, cfg_from_yaml_file
    from pcdet.models import DatasetInfo, build_network
    monkeypatch.setenv('PCP_CONV_ALGO', algo)
    g = load_golden(fixture)
    cfg = cfg_from_yaml_file(os.path.join(REPO, 'practical-collab-perception_amd', 'tools', 'cfgs', 'v2x_sim_models', 'v2x_pointpillar_disco.yaml'),
                             EasyDict())
    for key in ('BEV_MAKER_RSU', 'BEV_MAKER_CAR', 'BEV_MAKER_EARLY'):
        cfg.MODEL[key].CKPT = None
    vs = [p.VOXEL_SIZE for p in cfg.DATA_CONFIG.DATA_PROCESSOR if 'VOXEL_SIZE' in p][0]
    ds = DatasetInfo(cfg.CLASS_NAMES, cfg.DATA_CONFIG.POINT_CLOUD_RANGE, vs, len(cfg.DATA_CONFIG.POINT_FEATURE_ENCODING.used_feature_list))
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
    st = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model = model.to(DEV)
    ocfg = EasyDict(json.loads(str(g['optimization_json'])))
    opt = build_optimizer(model, ocfg)
    pts, metas, B = _full_size_batch(g)
    batch = lambda: {'points': torch.from_numpy(pts).to(DEV), 'batch_size': B, 'metadata': metas,
                     'gt_boxes': torch.from_numpy(g['gt_boxes']).to(DEV)}
    return g, model, opt, ocfg, batch, build_scheduler


@pytest.mark.parametrize('fixture', ['g7_train_full.npz', 'g7_train_full_b4.npz'])
def test_bf16_full_size_first_iteration_tracks_the_reference(monkeypatch, fixture):
    """config 5 at BASELINE's full size in the bf16 loop -- the first-iteration loss of the reference's own train step
    (golden g7_train_full; round 6: g7_train_full_b4 = the four-frame batch bench.py --train --conv-algo bf16 times) to 1 %, the
    clipped-gradient norm to 5 %"""
    g, model, opt, ocfg, batch, build_scheduler = _full_size_disco(monkeypatch, 'bf16', fixture)
    sched, _ = build_scheduler(opt, 5, ocfg.NUM_EPOCHS, -1, ocfg)
    sched.step(0)
    model.train()
    opt.zero_grad()
    ret, tb, _disp = model(batch())
    ret['loss'].backward()
    lv = float(ret['loss'].detach())
    assert abs(lv - float(g['loss'])) <= 1e-2 * abs(float(g['loss'])), (lv, float(g['loss']))
    ref = g['grad_digest']                                   # per tensor: [L2 norm, sum, max |.|] of the reference's gradients
    ref_norm = float(np.sqrt((ref[:, 0] ** 2).sum()))
    opt.clip_grad_norm(ocfg.GRAD_NORM_CLIP)
    opt.step()
    assert abs(opt.grad_norm() - ref_norm) <= 5e-2 * ref_norm, (opt.grad_norm(), ref_norm)
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_bf16_loss_curve_tracks_the_fp32_loop_over_20_iterations(monkeypatch):
    """the same 20 iterations (full size, one frame, the recipe's one-cycle schedule) in the fp32 loop, in the fp32 loop with ANOTHER
    summation order (direct kernels instead of Winograd: both exact fp32 arithmetic) and in the bf16 loop.  This short, steep schedule is
    chaotic: the two fp32 loops drift up to ~15 % apart per iteration (profiles/r04_loss_curves.txt), so "within 2 % per iteration" is not
    a property even fp32 has.  What is asserted: the first three iterations (before the drift amplifies) within 3.5 % of fp32 (measured
    0.08 / 1.4 / 2.5 %; every run gives the same curves since the pillar rows are sorted, pcp_voxelize_sort_pillar_rows); the bf16
    loop's largest deviation from the fp32 loop at most 2 x the deviation between the two fp32 loops (5 % floor); the mean of the last five
    iterations no further from the fp32 loop's than 2 x the other fp32 loop's (15 % floor); and the loss falls by > 10 x like the fp32 loop's."""
    curves = {}
    for algo in ('auto', 'direct', 'bf16'):
        g, model, opt, ocfg, batch, build_scheduler = _full_size_disco(monkeypatch, algo)
        sched, _ = build_scheduler(opt, 20, 1, -1, ocfg)
        losses = []
        for it in range(20):
            sched.step(it)
            model.train()
            opt.zero_grad()
            ret, tb, _disp = model(batch())
            model.update_global_step()
            ret['loss'].backward()
            opt.clip_grad_norm(ocfg.GRAD_NORM_CLIP)
            opt.step()
            losses.append(float(ret['loss'].detach()))
        curves[algo] = np.array(losses)
        del model, opt
        torch.cuda.empty_cache()
    a, d, b = curves['auto'], curves['direct'], curves['bf16']
    for k in ('auto', 'direct', 'bf16'):
        print('%-6s loop:' % k, np.round(curves[k], 4).tolist())
    assert np.isfinite(b).all()
    dev_fp32 = float((np.abs(d - a) / a).max())
    dev_bf16 = float((np.abs(b - a) / a).max())
    print('max relative deviation from the fp32 loop: fp32 direct kernels %.4f, bf16 %.4f' % (dev_fp32, dev_bf16))
    assert float((np.abs(b[:3] - a[:3]) / a[:3]).max()) <= 3.5e-2, (a[:3], b[:3])
    assert dev_bf16 <= max(2.0 * dev_fp32, 5e-2), (dev_bf16, dev_fp32)
    a5, d5, b5 = a[-5:].mean(), d[-5:].mean(), b[-5:].mean()
    assert abs(b5 - a5) <= max(2.0 * abs(d5 - a5), 0.15 * a5), (a5, d5, b5)
    assert b[-1] < 0.1 * b[0] and a[-1] < 0.1 * a[0]


@pytest.mark.parametrize('algo', ['auto', 'bf16'])
def test_training_iterations_are_bitwise_reproducible(monkeypatch, algo):
    """two runs of the first two full-size iterations from the same state: the same losses and the same gradient bits.  The pillariser hands a
    pillar's points out in atomic order; the training path sorts them (pcp_voxelize_sort_pillar_rows) because its per-point GEMMs sum over the
    rows -- without the sort the PFN's weight gradients differ in their last bits from run to run (and a chaotic schedule amplifies that)."""
    runs = []
    for _rep in range(2):
        g, model, opt, ocfg, batch, build_scheduler = _full_size_disco(monkeypatch, algo)
        sched, _ = build_scheduler(opt, 20, 1, -1, ocfg)
        out = []
        for it in range(2):
            sched.step(it)
            model.train()
            opt.zero_grad()
            ret, tb, _disp = model(batch())
            model.update_global_step()
            ret['loss'].backward()
            out.append((float(ret['loss'].detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
            opt.clip_grad_norm(ocfg.GRAD_NORM_CLIP)
            opt.step()
        runs.append(out)
        del model, opt
        torch.cuda.empty_cache()
    for (la, ga), (lb, gb) in zip(*runs):
        assert la == lb
        bad = [n for n in ga if not torch.equal(ga[n], gb[n])]
        assert not bad, bad[:8]


def test_single_model_train_step_matches_reference():
    """configs 3 / 4 (no fusion): VFE (11 raw features -> 17-d rows, padded to 32 floats) -> scatter -> backbone 64/128/256 -> CenterHead,
    two iterations of the reference's train loop (tests/golden/g7b_train_ego.npz).  Loss / lr / momentum as the reference; gradients
    and updated parameters inside the fixture's ReLU-mask noise band (see the disco test), global relative L2 of the sampled
    gradients 2e-2."""
    sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd', 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from pcdet.config import EasyDict
    g = load_golden('g7b_train_ego.npz')
    meta = g['meta']
    names = [str(n) for n in g['trainable']]
    model = _build(g)
    ocfg = EasyDict(meta['optimization'])
    opt = build_optimizer(model, ocfg)
    sched, _ = build_scheduler(opt, meta['total_it_each_epoch'], ocfg.NUM_EPOCHS, -1, ocfg)
    params = dict(model.named_parameters())
    assert set(names) == set(n for n, p in params.items() if p.requires_grad)
    for it in range(2):
        sched.step(it)
        assert abs(opt.lr - float(g['it%d_lr' % it])) < 1e-12 and abs(opt.mom - float(g['it%d_mom' % it])) < 1e-12
        model.train()
        opt.zero_grad()
        batch = {'points': torch.from_numpy(g['points']).to(DEV), 'batch_size': 2, 'metadata': [{}, {}],
                 'gt_boxes': torch.from_numpy(g['gt_boxes']).to(DEV)}
        ret, tb, _disp = model(batch)
        loss = ret['loss']
        model.update_global_step()
        loss.backward()
        ref_tb = json.loads(str(g['it%d_tb_json' % it]))
        tol = 2e-5 if it == 0 else 3e-3
        lv = float(loss.detach())
        assert abs(lv - float(g['it%d_loss' % it])) <= tol * abs(float(g['it%d_loss' % it])), (it, lv, float(g['it%d_loss' % it]))
        for k, v in ref_tb.items():
            assert abs(tb[k] - v) <= max(tol, 2e-4) * abs(v) + 1e-9, (k, tb[k], v)
        if it == 0:
            np.testing.assert_allclose(batch['spatial_features_2d'].detach().cpu().numpy()[:, ::8], g['map_probe'], rtol=0, atol=1e-4)
            gmax = max(float(np.abs(g['g0/' + n]).max()) for n in names)
            num = den = 0.0
            for n in names:
                ref = g['g0/' + n]
                mine = _sample(params[n].grad)
                scale = max(float(np.abs(ref).max()), 1e-4 * gmax)
                assert np.abs(mine - ref).max() <= 3e-2 * scale, (n, float(np.abs(mine - ref).max()), scale)
                num += float(((mine.astype(np.float64) - ref) ** 2).sum())
                den += float((ref.astype(np.float64) ** 2).sum())
            assert num <= (2e-2 ** 2) * den, (num / den) ** 0.5
        opt.clip_grad_norm(ocfg.GRAD_NORM_CLIP)
        opt.step()
        if it == 0:
            assert abs(opt.grad_norm() - float(g['it0_grad_norm'])) <= 5e-4 * float(g['it0_grad_norm'])
            for n in names:
                assert np.abs(_sample(params[n]) - g['p1/' + n]).max() <= 2.1 * opt.lr, n
            sd = model.state_dict()
            for i, k in enumerate(str(k) for k in g['bn_keys']):
                a = sd[k].double()
                d = np.array([float(a.norm()), float(a.sum()), float(a.abs().max())])
                np.testing.assert_allclose(d, g['it0_bn_digest'][i], rtol=2e-4, atol=1e-6, err_msg=k)


@pytest.mark.parametrize('yaml_name', ['v2x_pointpillar_disco.yaml', 'v2x_pointpillar_basic_ego_early.yaml', 'v2x_pointpillar_basic_ego.yaml',
                                       'v2x_pointpillar_anchor.yaml', 'v2x_pointpillar_basic_car.yaml', 'v2x_pointpillar_basic_rsu.yaml'])
def test_train_py_runs_and_loss_decreases(tmp_path, yaml_name):
    """tools/train.py (reference command line) on a small synthetic set: 2 epochs x 4 iterations, checkpoint written and loadable,
    loss of the last iteration below the first (same frames every epoch).  DiscoNet (config 5), the two fusion-free configs, the
    anchor-head PointPillar and the HunterJr model of configs 1 / 2."""
    import re
    import subprocess
    tools = os.path.join(REPO, 'practical-collab-perception_amd', 'tools')
    cmd = [sys.executable, 'train.py', '--cfg_file', 'cfgs/v2x_sim_models/' + yaml_name, '--batch_size', '2', '--epochs', '2',
           '--output_dir', str(tmp_path), '--set', 'DATA_CONFIG.SYNTHETIC.POINTS_PER_AGENT', '4000', 'DATA_CONFIG.SYNTHETIC.NUM_FRAMES', '8',
           'OPTIMIZATION.LR', '0.003']
    r = subprocess.run(cmd, cwd=tools, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    losses = [float(m) for m in re.findall(r'loss ([0-9.]+)  lr', r.stdout + r.stderr)]
    assert len(losses) >= 2 and losses[-1] < losses[0], losses
    ck = torch.load(os.path.join(str(tmp_path), 'ckpt', 'checkpoint_epoch_2.pth'), map_location='cpu', weights_only=False)
    assert ck['epoch'] == 2 and ck['it'] == 8 and 'vfe.pfn_layers.0.linear.weight' in ck['model_state']
    assert all(torch.isfinite(v).all() for v in ck['model_state'].values() if v.dtype.is_floating_point)


def test_training_step_degenerate_batches():
    """frames without ground truth, without remote agents, and a batch of one frame: the step runs, losses and gradients stay finite"""
    sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd', 'tools'))
    from train_utils.optimization import build_optimizer
    from pcdet.config import EasyDict
    g = load_golden('g7_train.npz')
    model = _build(g)
    opt = build_optimizer(model, EasyDict(g['meta']['optimization']))
    opt.lr, opt.mom = 1e-4, 0.9
    pts = g['points']
    cases = []
    gt0 = g['gt_boxes'].copy()
    gt0[1] = 0.0                                                        # frame 1 has no boxes at all
    cases.append((pts, gt0, [{'se3_from_ego': {0: g['pose_0'], 2: g['pose_2']}}, {'se3_from_ego': {0: g['pose_0']}}], 2))
    cases.append((pts, np.zeros_like(g['gt_boxes']), [{'se3_from_ego': {0: g['pose_0'], 2: g['pose_2']}}, {'se3_from_ego': {0: g['pose_0']}}], 2))
    one = pts[pts[:, 0] == 0]
    cases.append((one, g['gt_boxes'][:1], [{'se3_from_ego': {0: g['pose_0'], 2: g['pose_2']}}], 1))
    ego_only = pts[pts[:, -1] == 1.0]                                   # no remote agent sent anything: fusion over the ego map alone
    cases.append((ego_only, g['gt_boxes'], [{'se3_from_ego': {}}, {'se3_from_ego': {}}], 2))
    for pts_c, gt_c, md, bs in cases:
        model.train()
        opt.zero_grad()
        batch = {'points': torch.from_numpy(np.ascontiguousarray(pts_c)).to(DEV), 'batch_size': bs, 'metadata': md,
                 'gt_boxes': torch.from_numpy(np.ascontiguousarray(gt_c)).to(DEV)}
        ret, tb, _ = model(batch)
        ret['loss'].backward()
        assert np.isfinite(tb['loss_total']), tb
        assert bool(torch.isfinite(opt.flat_g).all())
        opt.clip_grad_norm(10.0)
        opt.step()
        assert bool(torch.isfinite(opt.flat_p).all())


@pytest.mark.parametrize('fixture', ['g7_train_full.npz', 'g7_train_full_b4.npz'])
def test_disco_full_size_training_iteration_matches_the_reference(fixture):
    """(g7_train_full_b4, round 6: the four-frame batch and the GT boxes bench.py --train times, through the reference's own train step)
    Config 5 at BASELINE's full size (6 agents x 60 000 points, 512 x 512 grid): one iteration of the reference's own train step
    (tests/golden/g7_train_full.npz) -- loss terms to 5e-4 (fp32 summation order over 16 384 heat-map cells and train-mode BatchNorm
    statistics over one frame: the CPU reference itself is no more reproducible than that), clipped-gradient norm to 5e-3, per-tensor
    gradient norms to 3e-2 of the largest (the per-element comparison against the float64 oracle lives on the mini fixture)."""
    sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd', 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from pcdet.config import EasyDict, cfg_from_yaml_file
    from pcdet.models import DatasetInfo, build_network
    g = load_golden(fixture)
    cfg = cfg_from_yaml_file(os.path.join(REPO, 'practical-collab-perception_amd', 'tools', 'cfgs', 'v2x_sim_models', 'v2x_pointpillar_disco.yaml'),
                             EasyDict())
    for key in ('BEV_MAKER_RSU', 'BEV_MAKER_CAR', 'BEV_MAKER_EARLY'):
        cfg.MODEL[key].CKPT = None
    vs = [p.VOXEL_SIZE for p in cfg.DATA_CONFIG.DATA_PROCESSOR if 'VOXEL_SIZE' in p][0]
    ds = DatasetInfo(cfg.CLASS_NAMES, cfg.DATA_CONFIG.POINT_CLOUD_RANGE, vs, len(cfg.DATA_CONFIG.POINT_FEATURE_ENCODING.used_feature_list))
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
    st = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model = model.to(DEV)
    ocfg = EasyDict(json.loads(str(g['optimization_json'])))
    opt = build_optimizer(model, ocfg)
    sched, _ = build_scheduler(opt, 5, ocfg.NUM_EPOCHS, -1, ocfg)
    pts, metas, B = _full_size_batch(g)
    assert pts.shape[0] == int(g['N'])
    sched.step(0)
    model.train()
    opt.zero_grad()
    batch = {'points': torch.from_numpy(pts).to(DEV), 'batch_size': B, 'metadata': metas, 'gt_boxes': torch.from_numpy(g['gt_boxes']).to(DEV)}
    ret, tb, _disp = model(batch)
    ret['loss'].backward()
    lv = float(ret['loss'].detach())
    assert abs(lv - float(g['loss'])) <= 5e-4 * abs(float(g['loss'])), (lv, float(g['loss']))
    for k, v in json.loads(str(g['tb_json'])).items():
        assert abs(tb[k] - v) <= 1e-3 * abs(v) + 2e-6, (k, tb[k], v)
    names = [str(n) for n in g['trainable']]
    params = dict(model.named_parameters())
    ref = g['grad_digest']                                   # per tensor: [L2 norm, sum, max |.|]
    biggest = float(ref[:, 0].max())
    for i, n in enumerate(names):
        mine = float(params[n].grad.detach().double().norm())
        assert abs(mine - ref[i, 0]) <= 3e-2 * max(ref[i, 0], 1e-3 * biggest), (n, mine, ref[i, 0])
    opt.clip_grad_norm(ocfg.GRAD_NORM_CLIP)
    opt.step()
    assert abs(opt.grad_norm() - float(g['grad_norm'])) <= 5e-3 * float(g['grad_norm']), (opt.grad_norm(), float(g['grad_norm']))


def test_pointpillar_anchor_head_train_step_matches_reference():
    """MODEL.NAME PointPillar + AnchorHeadSingle (three anchor classes, direction classifier): two iterations of the reference's own train
    loop (tests/golden/g11_anchor_train.npz).  Targets bit exact, loss terms / lr / momentum as the reference; gradients inside the
    fixture's noise band (the oracle's own float32 run is 6e-2 from its float64 run in backbone block 2, tests/test_oracle_pins.py),
    global relative L2 of the sampled gradients 2e-2."""
    sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd', 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from pcdet.config import EasyDict
    g = load_golden('g11_anchor_train.npz')
    meta = g['meta']
    names = [str(n) for n in g['trainable']]
    model = _build(g)
    ocfg = EasyDict(meta['optimization'])
    opt = build_optimizer(model, ocfg)
    sched, _ = build_scheduler(opt, meta['total_it_each_epoch'], ocfg.NUM_EPOCHS, -1, ocfg)
    params = dict(model.named_parameters())
    assert set(names) == set(n for n, p in params.items() if p.requires_grad)
    for it in range(2):
        sched.step(it)
        assert abs(opt.lr - float(g['it%d_lr' % it])) < 1e-12 and abs(opt.mom - float(g['it%d_mom' % it])) < 1e-12
        model.train()
        opt.zero_grad()
        batch = {'points': torch.from_numpy(g['points']).to(DEV), 'batch_size': 2, 'metadata': [{}, {}],
                 'gt_boxes': torch.from_numpy(g['gt_boxes']).to(DEV)}
        ret, tb, _disp = model(batch)
        loss = ret['loss']
        model.update_global_step()
        loss.backward()
        ref_tb = json.loads(str(g['it%d_tb_json' % it]))
        tol = 2e-5 if it == 0 else 3e-3
        lv = float(loss.detach())
        assert abs(lv - float(g['it%d_loss' % it])) <= tol * abs(float(g['it%d_loss' % it])), (it, lv, float(g['it%d_loss' % it]))
        for k, v in ref_tb.items():
            assert abs(tb[k] - v) <= max(tol, 2e-4) * abs(v) + 1e-9, (k, tb[k], v)
        if it == 0:
            fr = model.dense_head.forward_ret_dict
            assert np.array_equal(fr['box_cls_labels'].cpu().numpy(), g['box_cls_labels'])
            assert np.array_equal(fr['reg_weights'].cpu().numpy(), g['reg_weights'])
            np.testing.assert_allclose(fr['box_reg_targets'].cpu().numpy(), g['box_reg_targets'], rtol=0, atol=1e-6)
            for k in ('cls_preds', 'box_preds', 'dir_cls_preds'):
                np.testing.assert_allclose(fr[k].detach().cpu().numpy(), g[k], rtol=0, atol=1e-3, err_msg=k)
            np.testing.assert_allclose(batch['spatial_features_2d'].detach().cpu().numpy()[:, ::8], g['map_probe'], rtol=0, atol=1e-4)
            gmax = max(float(np.abs(g['g0/' + n]).max()) for n in names)
            num = den = 0.0
            for n in names:
                ref = g['g0/' + n]
                mine = _sample(params[n].grad)
                scale = max(float(np.abs(ref).max()), 1e-4 * gmax)
                assert np.abs(mine - ref).max() <= 1e-1 * scale, (n, float(np.abs(mine - ref).max()), scale)
                num += float(((mine.astype(np.float64) - ref) ** 2).sum())
                den += float((ref.astype(np.float64) ** 2).sum())
            assert num <= (2e-2 ** 2) * den, (num / den) ** 0.5
            # the head's own parameters see no BatchNorm noise: tight
            for n in names:
                if n.startswith('dense_head.'):
                    ref = g['g0/' + n]
                    assert np.abs(_sample(params[n].grad) - ref).max() <= 2e-3 * float(np.abs(ref).max()), n
        opt.clip_grad_norm(ocfg.GRAD_NORM_CLIP)
        opt.step()
        if it == 0:
            assert abs(opt.grad_norm() - float(g['it0_grad_norm'])) <= 2e-3 * float(g['it0_grad_norm'])
            for n in names:
                assert np.abs(_sample(params[n]) - g['p1/' + n]).max() <= 2.1 * opt.lr, n
            sd = model.state_dict()
            for i, k in enumerate(str(k) for k in g['bn_keys']):
                a = sd[k].double()
                d = np.array([float(a.norm()), float(a.sum()), float(a.abs().max())])
                np.testing.assert_allclose(d, g['it0_bn_digest'][i], rtol=2e-4, atol=1e-6, err_msg=k)


def test_hunter_jr_train_step_matches_reference():
    """configs 1 / 2 (VFE -> scatter -> backbone -> HunterJr -> CenterHead): two iterations of the reference's own train loop on
    v2x_pointpillar_basic_car.yaml (tests/golden/g12_hunter_train.npz).  Locals / instances and targets exact, the eleven loss terms as the
    reference (1e-4 at iteration 0), gradients of every parameter tensor inside the fixture's noise band with a global relative L2 of 2e-2,
    the point-wise in-place correction and the filtered ground truth as the reference leaves them."""
    sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd', 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from pcdet.config import EasyDict
    g = load_golden('g12_hunter_train.npz')
    meta = g['meta']
    names = [str(n) for n in g['trainable']]
    model = _build(g)
    ocfg = EasyDict(meta['optimization'])
    opt = build_optimizer(model, ocfg)
    sched, _ = build_scheduler(opt, meta['total_it_each_epoch'], ocfg.NUM_EPOCHS, -1, ocfg)
    params = dict(model.named_parameters())
    assert set(names) == set(n for n, p in params.items() if p.requires_grad)
    for it in range(2):
        sched.step(it)
        assert abs(opt.lr - float(g['it%d_lr' % it])) < 1e-12 and abs(opt.mom - float(g['it%d_mom' % it])) < 1e-12
        model.train()
        opt.zero_grad()
        batch = {'points': torch.from_numpy(g['points']).to(DEV), 'batch_size': 2, 'metadata': [{}, {}],
                 'gt_boxes': torch.from_numpy(g['gt_boxes']).to(DEV), 'instances_tf': torch.from_numpy(g['instances_tf']).to(DEV)}
        ret, tb, _disp = model(batch)
        loss = ret['loss']
        model.update_global_step()
        loss.backward()
        ref_tb = json.loads(str(g['it%d_tb_json' % it]))
        tol = 1e-4 if it == 0 else 1e-2
        lv = float(loss.detach())
        for k, v in ref_tb.items():
            assert abs(tb[k] - v) <= tol * abs(v) + 1e-9, (it, k, tb[k], v)
        assert abs(lv - float(g['it%d_loss' % it])) <= tol * abs(float(g['it%d_loss' % it])), (it, lv, float(g['it%d_loss' % it]))
        if it == 0:
            fr = model.corrector.forward_return_dict
            m = fr['meta']
            assert np.array_equal(m.fg_local[:m.n_fg].cpu().numpy(), g['meta/locals2fg'])
            assert np.array_equal(m.local_key[:m.n_local].cpu().numpy(), g['meta/locals_bis'])
            assert np.array_equal(fr['points_cls_target'].cpu().numpy(), np.argmax(g['tgt/points_cls'], axis=1))
            for k in ('points_cls_logit', 'points_flow3d', 'points_embedding', 'locals_tf'):
                np.testing.assert_allclose(fr['prediction'][k].detach().cpu().numpy(), g['pred/' + k], rtol=0, atol=1e-3, err_msg=k)
            np.testing.assert_allclose(batch['points'].cpu().numpy(), g['points_after'], rtol=0, atol=1e-3)
            ga = g['gt_boxes_after']
            gb = batch['gt_boxes'].cpu().numpy()
            assert np.array_equal(gb[:, :ga.shape[1]], ga) and not gb[:, ga.shape[1]:].any()
            np.testing.assert_allclose(batch['spatial_features_2d'].detach().cpu().numpy()[:, ::8], g['map_probe'], rtol=0, atol=2e-4)
            gmax = max(float(np.abs(g['g0/' + n]).max()) for n in names)
            num = den = 0.0
            for n in names:
                ref = g['g0/' + n]
                assert params[n].grad is not None, n
                mine = _sample(params[n].grad)
                scale = max(float(np.abs(ref).max()), 1e-4 * gmax)
                assert np.abs(mine - ref).max() <= 1e-1 * scale, (n, float(np.abs(mine - ref).max()), scale)
                num += float(((mine.astype(np.float64) - ref) ** 2).sum())
                den += float((ref.astype(np.float64) ** 2).sum())
            assert num <= (2e-2 ** 2) * den, (num / den) ** 0.5
        opt.clip_grad_norm(ocfg.GRAD_NORM_CLIP)
        opt.step()
        if it == 0:
            assert abs(opt.grad_norm() - float(g['it0_grad_norm'])) <= 5e-3 * float(g['it0_grad_norm'])
            for n in names:
                assert np.abs(_sample(params[n]) - g['p1/' + n]).max() <= 2.1 * opt.lr, n
            sd = model.state_dict()
            for i, k in enumerate(str(k) for k in g['bn_keys']):
                a = sd[k].double()
                d = np.array([float(a.norm()), float(a.sum()), float(a.abs().max())])
                np.testing.assert_allclose(d, g['it0_bn_digest'][i], rtol=5e-4, atol=1e-6, err_msg=k)


def test_hunter_jr_teacher_bev_term_is_a_reported_value_without_a_gradient():
    """hunter_jr.py:352-365: with batch_dict['teacher_spatial_features_2d'] the corrector records loss_dtl_bev_img = mean over the pixels the
    teacher covers (row norm > 1e-3) of the per-pixel sum of smooth_l1(corrected map - teacher).  The reference never adds it to the training
    loss (hunter_jr.py:490-494): the loss and every gradient must be those of the run without a teacher, and the value must equal a torch
    fp32 CPU evaluation of the reference's expression on the map the model produced"""
    import torch.nn.functional as F
    g = load_golden('g12_hunter_train.npz')
    outs = []
    for with_teacher in (False, True):
        model = _build(g)
        model.train()
        batch = {'points': torch.from_numpy(g['points']).to(DEV), 'batch_size': 2, 'metadata': [{}, {}],
                 'gt_boxes': torch.from_numpy(g['gt_boxes']).to(DEV), 'instances_tf': torch.from_numpy(g['instances_tf']).to(DEV)}
        if with_teacher:
            C, H, W = model.corrector.num_points_feat, g['map_probe'].shape[2], g['map_probe'].shape[3]
            gen = torch.Generator().manual_seed(3)
            teacher = torch.randn((2, C, H, W), generator=gen) * 0.5
            teacher[:, :, : H // 3] = 0.0                                     # a region the teacher does not cover: masked out
            teacher[0, :, H // 2, :] *= 1e-6                                  # row norms below the 1e-3 threshold
            batch['teacher_spatial_features_2d'] = teacher.to(DEV)
        ret, tb, _ = model(batch)
        ret['loss'].backward()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        outs.append((model, batch, float(ret['loss'].detach()), grads))
    (m0, b0, l0, g0), (m1, b1, l1, g1) = outs
    # no gradient comes from the term: the two runs agree to the run-to-run noise of the float atomics in the point <-> BEV backward kernels
    assert abs(l0 - l1) <= 1e-6 * abs(l0) and set(g0) == set(g1)
    gmax = max(float(v.abs().max()) for v in g0.values())
    for n in g0:
        assert float((g0[n] - g1[n]).abs().max()) <= 1e-4 * max(float(g0[n].abs().max()), 1e-3 * gmax), n
    assert 'loss_dtl_bev_img' not in m0.corrector.forward_return_dict
    got = float(m1.corrector.forward_return_dict['loss_dtl_bev_img'])
    fused = b1['spatial_features_2d'].detach().cpu().float()
    t = b1['teacher_spatial_features_2d'].cpu()
    f2, t2 = fused.permute(0, 2, 3, 1).reshape(-1, fused.shape[1]), t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    mask = torch.linalg.norm(t2, dim=1) > 1e-3
    want = float(F.smooth_l1_loss(f2[mask], t2[mask], reduction='none').sum(dim=1).mean())
    assert 0 < int(mask.sum()) < mask.numel()
    assert abs(got - want) <= 1e-5 * abs(want), (got, want)


def _hunter_full_inputs():
    """the inputs of tests/golden/make_golden.py::hunter_full_inputs, regenerated (closed-form streams: nothing large is committed)"""
    s0 = synth.SEED_BASE + 960
    n = 12
    gt = np.zeros((1, n, 8), dtype=np.float32)
    for col, (lo, hi) in enumerate([(-48.0, 48.0), (-48.0, 48.0), (-3.0, -1.0), (3.0, 5.5), (1.5, 2.5), (1.4, 2.0), (-3.14159, 3.14159)]):
        gt[0, :, col] = synth.uniform(s0, col + 1, n, lo, hi)
    gt[0, :, 7] = 1.0
    fg, tf = synth.instance_foreground(77, gt[0], per_local=40)
    cloud = np.concatenate([synth.agent_cloud(agent=0, n_points=60000, layout='car'), fg], axis=0)
    return synth.collate([cloud]), gt, tf[None]


def test_hunter_jr_full_size_training_iteration_matches_the_reference():
    """Configs 1 / 2 at BASELINE's full size (60 000 points + 1 440 foreground rows, 512 x 512 grid, 128 x 128 x 384 BEV map): one iteration of
    the reference's own train step (tests/golden/g12_hunter_train_full.npz) -- locals / instances counts and the number of rows the flow
    head moved exact, all thirteen loss entries to 1e-3, clipped-gradient norm to 5e-3, per-tensor gradient norms to 3e-2 of the largest."""
    sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd', 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from pcdet.config import EasyDict, cfg_from_yaml_file
    from pcdet.models import DatasetInfo, build_network
    g = load_golden('g12_hunter_train_full.npz')
    cfg = cfg_from_yaml_file(os.path.join(REPO, 'practical-collab-perception_amd', 'tools', 'cfgs', 'v2x_sim_models', 'v2x_pointpillar_basic_car.yaml'),
                             EasyDict())
    vs = [p.VOXEL_SIZE for p in cfg.DATA_CONFIG.DATA_PROCESSOR if 'VOXEL_SIZE' in p][0]
    ds = DatasetInfo(cfg.CLASS_NAMES, cfg.DATA_CONFIG.POINT_CLOUD_RANGE, vs, len(cfg.DATA_CONFIG.POINT_FEATURE_ENCODING.used_feature_list))
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
    st = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model = model.to(DEV)
    ocfg = EasyDict(json.loads(str(g['optimization_json'])))
    opt = build_optimizer(model, ocfg)
    sched, _ = build_scheduler(opt, 5, ocfg.NUM_EPOCHS, -1, ocfg)
    pts, gt, tf = _hunter_full_inputs()
    assert pts.shape[0] == int(g['N'])
    sched.step(0)
    model.train()
    opt.zero_grad()
    batch = {'points': torch.from_numpy(pts).to(DEV), 'batch_size': 1, 'metadata': [{}], 'gt_boxes': torch.from_numpy(gt).to(DEV),
             'instances_tf': torch.from_numpy(tf).to(DEV)}
    ret, tb, _disp = model(batch)
    ret['loss'].backward()
    m = model.corrector.forward_return_dict['meta']
    assert [m.n_fg, m.n_local, m.n_inst] == [int(v) for v in g['counts']]
    moved = int((np.abs(batch['points'].cpu().numpy() - pts).max(1) > 0).sum())
    assert abs(moved - int(g['moved_rows'])) <= 3, (moved, int(g['moved_rows']))          # rows within 1e-6 of the 0.3 probability threshold
    np.testing.assert_allclose(batch['spatial_features_2d'].detach().cpu().numpy()[0, ::8, ::16, ::16], g['map_probe'], rtol=0, atol=1e-3)
    lv = float(ret['loss'].detach())
    assert abs(lv - float(g['loss'])) <= 1e-3 * abs(float(g['loss'])), (lv, float(g['loss']))
    for k, v in json.loads(str(g['tb_json'])).items():
        assert abs(tb[k] - v) <= 1e-3 * abs(v) + 2e-6, (k, tb[k], v)
    names = [str(n) for n in g['trainable']]
    params = dict(model.named_parameters())
    ref = g['grad_digest']
    biggest = float(ref[:, 0].max())
    for i, n in enumerate(names):
        mine = float(params[n].grad.detach().double().norm())
        assert abs(mine - ref[i, 0]) <= 3e-2 * max(ref[i, 0], 1e-3 * biggest), (n, mine, ref[i, 0])
    opt.clip_grad_norm(ocfg.GRAD_NORM_CLIP)
    opt.step()
    assert abs(opt.grad_norm() - float(g['grad_norm'])) <= 5e-3 * float(g['grad_norm']), (opt.grad_norm(), float(g['grad_norm']))


def test_pointpillar_anchor_full_size_training_iteration_matches_the_reference():
    """PointPillar + AnchorHeadSingle at full geometry (60 000 points, 128 x 128 x 6 = 98 304 anchors, 24 boxes of three classes): one
    iteration of the reference's own train step (tests/golden/g11_anchor_train_full.npz) -- the 98 304 labels bit exact (SHA-256), the
    loss terms to 1e-3, clipped-gradient norm to 5e-3, per-tensor gradient norms to 3e-2 of the largest."""
    import hashlib
    sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd', 'tools'))
    from train_utils.optimization import build_optimizer, build_scheduler
    from pcdet.config import EasyDict
    g = load_golden('g11_anchor_train_full.npz')
    meta = g['meta']
    model = _build(g)
    ocfg = EasyDict(meta['optimization'])
    opt = build_optimizer(model, ocfg)
    sched, _ = build_scheduler(opt, 5, ocfg.NUM_EPOCHS, -1, ocfg)
    pts = synth.collate([synth.agent_cloud(agent=1, n_points=60000, layout='lately')])
    assert pts.shape[0] == int(g['N'])
    sched.step(0)
    model.train()
    opt.zero_grad()
    batch = {'points': torch.from_numpy(pts).to(DEV), 'batch_size': 1, 'metadata': [{}], 'gt_boxes': torch.from_numpy(g['gt_boxes']).to(DEV)}
    ret, tb, _disp = model(batch)
    ret['loss'].backward()
    fr = model.dense_head.forward_ret_dict
    lab = fr['box_cls_labels'].cpu().numpy().astype(np.int32)
    assert np.array_equal(np.bincount(lab.reshape(-1) + 1, minlength=5), g['label_hist'])
    assert hashlib.sha256(np.ascontiguousarray(lab).tobytes()).hexdigest() == str(g['labels_sha'])
    a = fr['box_reg_targets'].double()
    np.testing.assert_allclose([float(a.norm()), float(a.sum()), float(a.abs().max())], g['reg_targets_digest'], rtol=1e-5, atol=1e-5)
    lv = float(ret['loss'].detach())
    assert abs(lv - float(g['loss'])) <= 1e-3 * abs(float(g['loss'])), (lv, float(g['loss']))
    for k, v in json.loads(str(g['tb_json'])).items():
        assert abs(tb[k] - v) <= 1e-3 * abs(v) + 2e-6, (k, tb[k], v)
    names = [str(n) for n in g['trainable']]
    params = dict(model.named_parameters())
    ref = g['grad_digest']
    biggest = float(ref[:, 0].max())
    for i, n in enumerate(names):
        mine = float(params[n].grad.detach().double().norm())
        assert abs(mine - ref[i, 0]) <= 3e-2 * max(ref[i, 0], 1e-3 * biggest), (n, mine, ref[i, 0])
    opt.clip_grad_norm(ocfg.GRAD_NORM_CLIP)
    opt.step()
    assert abs(opt.grad_norm() - float(g['grad_norm'])) <= 5e-3 * float(g['grad_norm']), (opt.grad_norm(), float(g['grad_norm']))
