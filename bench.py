"""bench.py -- frames/s of the PointPillars collaborative-perception hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config car|ego|early|disco] [--batch B]

A "step" is one pass of the hot path (points resident in HBM -> final boxes) over one batch of B synthetic 60k-point clouds
per GPU.  Default workload = BASELINE.json configs[1]: v2x_pointpillar_basic_car.yaml, single-agent inference (VFE -> scatter ->
BEV backbone -> HunterJr -> CenterHead -> decode -> rotated NMS), B = BATCH_SIZE_PER_GPU = 4.
N > 1: launched by torch.distributed.run, one rank per GPU; frames are independent so ranks are replicas on different frames
(weak scaling, no data-path collective); value = all ranks' frames / max-over-ranks time.

Rank 0 prints ONE JSON line with the contract keys plus
  roofline     : the dominant kernel (3x3 conv, fp32 MFMA implicit GEMM) timed live with HIP events on its stream
  cpu_baseline : the oracle (CPU restatement of the reference modules) timed on this box's host cores, bounded sample.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, 'practical-collab-perception_amd')
for _p in (REPO, PKG, os.path.join(REPO, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIGS = {
    'car': dict(yaml='v2x_pointpillar_basic_car.yaml', layout='car', agents_in_cloud=1,
                name='v2x_pointpillar_basic_car single-agent inference (VFE+scatter+backbone+HunterJr+CenterHead+NMS)'),
    'ego': dict(yaml='v2x_pointpillar_basic_ego.yaml', layout='lately', agents_in_cloud=1,
                name='v2x_pointpillar_basic_ego lately-fusion ego pass (60k points incl. 300 MoDAR rows)'),
    'early': dict(yaml='v2x_pointpillar_basic_ego_early.yaml', layout='early', agents_in_cloud=6,
                  name='v2x_pointpillar_basic_ego_early early fusion (6 x 60k points merged)'),
    'disco': dict(yaml='v2x_pointpillar_disco.yaml', layout='disco', agents_in_cloud=6,
                  name='v2x_pointpillar_disco mid fusion (3 BEV makers + warp/fuse, 6 x 60k points)'),
}
MFMA_F32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)


def load_cfg(yaml_name):
    from pcdet.config import EasyDict, cfg_from_yaml_file
    cfg = cfg_from_yaml_file(os.path.join(PKG, 'tools', 'cfgs', 'v2x_sim_models', yaml_name), EasyDict())
    for key in ('BEV_MAKER_RSU', 'BEV_MAKER_CAR', 'BEV_MAKER_EARLY'):
        if cfg.MODEL.get(key, None) is not None:
            cfg.MODEL[key].CKPT = None          # random-init weights: no checkpoints offline
    return cfg


def build_model(cfg):
    from pcdet.models import DatasetInfo, build_network
    from pcp_amd import synth
    enc = cfg.DATA_CONFIG.POINT_FEATURE_ENCODING
    vs = [p.VOXEL_SIZE for p in cfg.DATA_CONFIG.DATA_PROCESSOR if 'VOXEL_SIZE' in p][0]
    ds = DatasetInfo(cfg.CLASS_NAMES, cfg.DATA_CONFIG.POINT_CLOUD_RANGE, vs, len(enc.used_feature_list))
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), ds)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    state = synth.fill_state_dict(shapes)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    return model, state, ds


def make_points(conf, batch, rank):
    """B frames per rank; frame f of rank r uses agent streams 100*r + 10*f + a (distinct data on every rank)."""
    from pcp_amd import synth
    clouds, metas = [], []
    for f in range(batch):
        parts = []
        for a in range(conf['agents_in_cloud']):
            c = synth.agent_cloud(agent=1000 * rank + 10 * f + a, n_points=60000, layout=conf['layout'])
            if conf['layout'] == 'disco':
                c[:, -1] = float(a)
            parts.append(c)
        clouds.append(np.concatenate(parts, 0))
        metas.append({'se3_from_ego': {a: synth.agent_pose(a) for a in range(conf['agents_in_cloud']) if a != 1}})
    return synth.collate(clouds), metas


def cpu_baseline(conf, cfg, state, batch_points, metas, budget_s=25.0):
    """oracle forward on the host cores: 1 frame per run (B=1), as many runs as fit the budget (at least 1)."""
    from oracle import model as omodel
    from pcdet.config import EasyDict

    def plain(d):
        if isinstance(d, dict):
            return {k: plain(v) for k, v in d.items()}
        if isinstance(d, (list, tuple)):
            return [plain(v) for v in d]
        return d
    vs = [p.VOXEL_SIZE for p in cfg.DATA_CONFIG.DATA_PROCESSOR if 'VOXEL_SIZE' in p][0]
    arch = omodel.arch_from_cfg(plain(cfg.MODEL), list(cfg.DATA_CONFIG.POINT_CLOUD_RANGE), list(vs))
    pts = batch_points[batch_points[:, 0] == 0].copy()
    # threads actually used: the CPUs this process may run on, capped at 32 (oneDNN convs at these sizes stop scaling
    # well before that and oversubscribing a shared host makes the baseline meaningless)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))
    torch.set_num_threads(cores)
    t0 = time.time()
    omodel.forward(pts, state, arch, metadata=metas[:1])          # warm-up (also builds the C NMS oracle)
    warm = time.time() - t0
    runs, spent = 0, 0.0
    while runs < 1 or (spent + spent / max(runs, 1) < budget_s - warm and runs < 20):
        t1 = time.time()
        omodel.forward(pts, state, arch, metadata=metas[:1])
        spent += time.time() - t1
        runs += 1
    return dict(value=round(runs / spent, 4), unit='frames/s', cores=cores, kind='port',
                sample='%d x 1 frame (%d points), oracle/model.py forward incl. decode+NMS, torch CPU threads=%d' % (runs, pts.shape[0], cores))


class ConvTimer:
    """HIP events around every pcp_conv3x3 launch on the launch stream (instrumented pass, outside the timed region)."""

    def __init__(self):
        self.records = []

    def install(self):
        from pcp_amd import ops
        self._orig = ops.conv3x3
        timer = self

        def timed(x, packed, bias, cin, cout, cout_pad, stride=1, relu=True, out=None, in_ch_off=0, out_ch_off=0):
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            r = timer._orig(x, packed, bias, cin, cout, cout_pad, stride=stride, relu=relu, out=out, in_ch_off=in_ch_off,
                            out_ch_off=out_ch_off)
            e1.record(s)
            B, H, W, _ = x.shape
            Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
            timer.records.append((e0, e1, 2.0 * B * Ho * Wo * cout * 9 * cin, stride, cout_pad))
            return r
        ops.conv3x3 = timed
        self._orig_w = ops.conv3x3_winograd

        def timed_w(x, packed, bias, cin, cout, cout_pad, relu=True, out=None, in_ch_off=0, out_ch_off=0):
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            r = timer._orig_w(x, packed, bias, cin, cout, cout_pad, relu=relu, out=out, in_ch_off=in_ch_off, out_ch_off=out_ch_off)
            e1.record(s)
            B, H, W, _ = x.shape
            # same dispatch rule as pcp_conv3x3_winograd (csrc/wino.hip): 64-tile instantiation <2> for the long-K layers
            big = cin >= 256 and B * ((H + 15) // 16) * ((W + 15) // 16) * (cout_pad // 64) >= 256
            timer.records.append((e0, e1, 2.0 * B * H * W * cout * 9 * cin, 1, -2 if big else -1))
            return r
        ops.conv3x3_winograd = timed_w
        self._orig_b3 = ops.conv3x3_bf16x3

        def timed_b3(x, packed, bias, cin, cout, cout_pad, stride=1, relu=True, out=None, in_ch_off=0, out_ch_off=0, plain=False):
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            r = timer._orig_b3(x, packed, bias, cin, cout, cout_pad, stride=stride, relu=relu, out=out, in_ch_off=in_ch_off,
                               out_ch_off=out_ch_off, plain=plain)
            e1.record(s)
            B, H, W, _ = x.shape
            Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
            timer.records.append((e0, e1, 2.0 * B * Ho * Wo * cout * 9 * cin, stride, -3))
            return r
        ops.conv3x3_bf16x3 = timed_b3
        self._orig_w4 = ops.conv3x3_winograd4
        self.w4 = []

        def timed_w4(x, packed, bias, cin, cout, cout_pad, relu=True, out=None, in_ch_off=0, out_ch_off=0):
            # three launches (input transform, batched GEMM, output transform): the library's measurement entry point brackets each with
            # HIP events on the launch stream (pcp_conv3x3_winograd4_timed)
            st = []
            r = timer._orig_w4(x, packed, bias, cin, cout, cout_pad, relu=relu, out=out, in_ch_off=in_ch_off, out_ch_off=out_ch_off,
                               stage_times=st)
            B, H, W, _ = x.shape
            timer.w4.append(st[0] + (2.0 * B * H * W * cout * 9 * cin,))
            return r
        ops.conv3x3_winograd4 = timed_w4

    def remove(self):
        from pcp_amd import ops
        ops.conv3x3 = self._orig
        ops.conv3x3_winograd = self._orig_w
        ops.conv3x3_bf16x3 = self._orig_b3
        ops.conv3x3_winograd4 = self._orig_w4

    def summary(self):
        torch.cuda.synchronize()
        # dominant kernel = the fused Winograd 3x3 kernel (k_conv3x3_wino); falls back to the direct stride-1 instantiation
        self.peak, self.exec_mult = MFMA_F32_PEAK_TFLOPS, 4.0 / 9.0
        sel = [(e0.elapsed_time(e1) * 1e-3, fl) for (e0, e1, fl, st, cp) in self.records if cp == -3 and st == 1]
        self.dominant = 'k_conv3x3_bf16x3 (3x3 implicit GEMM, split-bf16 operands, 3 x v_mfma_f32_32x32x16_bf16 per product) [opt-in mode]'
        if sel:
            self.peak, self.exec_mult = 2500.0, 3.0                     # dense bf16 MFMA peak; three MFMAs per algorithmic product
        else:
            sel = [(e0.elapsed_time(e1) * 1e-3, fl) for (e0, e1, fl, st, cp) in self.records if cp == -2]
            self.dominant = 'k_conv3x3_wino<2> (3x3 s1 fused Winograd F(2x2,3x3), 64-tile workgroups, v_mfma_f32_32x32x2_f32)'
        if not sel:
            sel = [(e0.elapsed_time(e1) * 1e-3, fl) for (e0, e1, fl, st, cp) in self.records if st == 1 and cp % 64 == 0]
            self.dominant = 'k_conv3x3<1,8,16,64,2,2> (3x3 s1 implicit GEMM, v_mfma_f32_32x32x2_f32)'
            self.exec_mult = 1.0
        allc = [(e0.elapsed_time(e1) * 1e-3, fl) for (e0, e1, fl, st, cp) in self.records]
        allc += [((a + b + c) * 1e-3, alg) for (a, b, c, _gf, alg) in self.w4]
        extra = {}
        if self.w4 and self.peak == MFMA_F32_PEAK_TFLOPS:
            # dominant kernel = the batched GEMM of the F(4x4,3x3) path: its executed flops ARE its algorithmic work
            sel = [(b * 1e-3, gf) for (_a, b, _c, gf, _alg) in self.w4]
            self.dominant = ('k_w4_gemm (36 batched GEMMs [tiles x cin] x [cin x cout] of the Winograd F(4x4,3x3) wide-layer convolutions, '
                             '128x128x32 LDS tiles, v_mfma_f32_32x32x2_f32)')
            self.exec_mult = 1.0
            n = len(self.w4)
            extra = {'winograd4_avg_us': {'input_transform': round(1e3 * sum(r[0] for r in self.w4) / n, 2),
                                          'gemm': round(1e3 * sum(r[1] for r in self.w4) / n, 2),
                                          'output_transform': round(1e3 * sum(r[2] for r in self.w4) / n, 2)},
                     'winograd4_conv_algorithmic_tflops': round(sum(r[4] for r in self.w4) / sum(r[0] + r[1] + r[2] for r in self.w4) / 1e9, 3)}
        t, f = sum(a for a, _ in sel), sum(b for _, b in sel)
        ta, fa = sum(a for a, _ in allc), sum(b for _, b in allc)
        return dict(launches=len(sel), avg_us=1e6 * t / max(len(sel), 1), tflops=f / t / 1e12 if t > 0 else 0.0,
                    all_conv_launches=len(allc), all_conv_tflops=fa / ta / 1e12 if ta > 0 else 0.0, all_conv_ms=1e3 * ta, extra=extra)


def pmc_traffic(kernel_key):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (separate FETCH_SIZE and
    WRITE_SIZE runs of this same command; FETCH_SIZE doubled per MI355X_MICROARCH.md for 16-B/lane streaming reads)."""
    path = os.path.join(REPO, 'profiles', 'r01_pmc_traffic.json')
    if not os.path.isfile(path):
        return None
    with open(path) as f:
        d = json.load(f)
    e = d.get(kernel_key)
    return None if e is None else e['bytes_per_launch']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='car', choices=sorted(CONFIGS))
    ap.add_argument('--batch', type=int, default=0, help='frames per GPU per step (0 = BATCH_SIZE_PER_GPU of the YAML)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dense-first-layer', action='store_true', help='A/B switch: always write the dense canvas and run the first backbone '
                    'layer as the dense stride-2 conv (default in pipeline mode: from the pillar list when the cloud is sparse)')
    ap.add_argument('--no-optin', action='store_true', help='skip the informational opt-in (bf16x3) pass after the fp32 measurement (clean '
                    'rocprofv3 kernel statistics of the headline path)')
    ap.add_argument('--graph', action='store_true', help='replay the whole forward as one hipGraph (launch-bound small batches)')
    ap.add_argument('--conv-algo', default=None, choices=['auto', 'direct', 'winograd', 'winograd4', 'bf16x3', 'bf16'],
                    help='3x3 convolution arithmetic (default auto = fp32 MFMA: direct / Winograd).  bf16x3 is the OPT-IN split-bf16 mode '
                         '(three bf16 MFMAs per product, fp32 accumulate, ~1e-5 relative error); the JSON line then says so in `dtype`')
    ap.add_argument('--shard', default='frame', choices=['frame', 'agent'], help="frame (default): every rank is a replica on its own frames; "
                    "agent: configs early / disco only -- each rank holds the points of ITS agents, one all-gather of raw points (early) or "
                    "of compressed BEV maps (disco) per step, then the frames of the batch are dealt to the ranks (pcdet/models/sharded.py; "
                    "strong scaling: the group processes ONE batch per step)")
    ap.add_argument('--train', action='store_true', help='configs ego / early / disco: time full training iterations (forward + backward + '
                    'clip + fused Adam one-cycle step; data parallel over ranks with one RCCL all-reduce of the flat gradient)')
    args = ap.parse_args()

    if args.conv_algo is not None:
        if args.conv_algo == 'bf16' and not args.train:
            raise SystemExit('--conv-algo bf16 (plain bf16 products) is the mixed-precision TRAINING mode: add --train')
        os.environ['PCP_CONV_ALGO'] = args.conv_algo
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (the hot path has no CPU fallback)'
    # one rank per GPU; PCP_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than ranks
    # (ranks then share devices round-robin -- a functional check, not a measurement)
    dev_index = local_rank % torch.cuda.device_count() if world > 1 else 0
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(dev_index)
        dist.init_process_group(backend=os.environ.get('PCP_BENCH_BACKEND', 'nccl'))
    dev = torch.device('cuda', dev_index)

    conf = CONFIGS[args.config]
    cfg = load_cfg(conf['yaml'])
    batch = args.batch or int(cfg.OPTIMIZATION.BATCH_SIZE_PER_GPU)
    model, state, ds = build_model(cfg)
    if args.train and args.config == 'car':
        raise SystemExit('--train: HunterJr (basic_car) has no training kernels; use --config ego | early | disco')
    model = model.to(dev).eval()
    for m in model.modules():
        if hasattr(m, 'materialize_pillars'):
            m.materialize_pillars = False       # per-pillar API tensors are not consumed downstream
            m.reuse_buffers = True
            m.sparse_first_layer = not args.dense_first_layer   # sparse clouds: first backbone layer from the pillar list, no dense canvas
    pts_np, metas = make_points(conf, batch, rank)
    pristine = torch.from_numpy(pts_np).to(dev)
    work = torch.empty_like(pristine)

    graphed = None
    if args.graph:
        from pcdet.models.graphed import GraphedDetector
        graphed = GraphedDetector(model, pristine, batch, metas)

    train_state = None
    if args.train:
        sys.path.insert(0, os.path.join(PKG, 'tools'))
        from train_utils.optimization import build_optimizer, build_scheduler
        from pcp_amd import synth
        opt = build_optimizer(model, cfg.OPTIMIZATION)
        sched, _ = build_scheduler(opt, 1000, cfg.OPTIMIZATION.NUM_EPOCHS, -1, cfg.OPTIMIZATION)
        gt = np.zeros((batch, 40, 8), dtype=np.float32)
        for f in range(batch):
            n = 40 - 3 * f
            sd = synth.SEED_BASE + 900 + 100 * rank + f
            gt[f, :n, 0] = synth.uniform(sd, 1, n, -50.0, 50.0)
            gt[f, :n, 1] = synth.uniform(sd, 2, n, -50.0, 50.0)
            gt[f, :n, 2] = synth.uniform(sd, 3, n, -3.0, -1.0)
            gt[f, :n, 3] = synth.uniform(sd, 4, n, 3.0, 5.5)
            gt[f, :n, 4] = synth.uniform(sd, 5, n, 1.5, 2.5)
            gt[f, :n, 5] = synth.uniform(sd, 6, n, 1.4, 2.0)
            gt[f, :n, 6] = synth.uniform(sd, 7, n, -3.14159, 3.14159)
            gt[f, :n, 7] = 1.0
        train_state = dict(opt=opt, sched=sched, gt=torch.from_numpy(gt).to(dev), it=0)

    def train_step():
        ts = train_state
        ts['sched'].step(ts['it'])
        model.train()
        ts['opt'].zero_grad()
        bd = {'points': pristine, 'batch_size': batch, 'metadata': metas, 'gt_boxes': ts['gt']}
        ret, tb, _ = model(bd)
        model.update_global_step()
        ret['loss'].backward()
        ts['opt'].clip_grad_norm(cfg.OPTIMIZATION.GRAD_NORM_CLIP)
        ts['opt'].step()
        ts['it'] += 1
        ts['last_loss'] = tb['loss_total']
        return []

    sharded_runner = None
    if args.shard == 'agent':
        if args.config not in ('early', 'disco') or args.train or args.graph:
            raise SystemExit('--shard agent: inference of --config early | disco')
        from pcdet.models import sharded
        sharded_runner = (sharded.AgentShardedEarlyFusion if args.config == 'early' else sharded.AgentShardedMidFusion)(model)
        # every rank generated the SAME batch (rank 0's streams); it keeps only the rows of its agents (round-robin over agents)
        pts_np, metas = make_points(conf, batch, 0)
        agent_of_row = pts_np[:, -1] if conf['layout'] == 'disco' else np.repeat(np.arange(conf['agents_in_cloud']), 60000)[None].repeat(batch, 0).reshape(-1)
        mine = pts_np[(agent_of_row.astype(np.int64) % world) == rank]
        pristine = torch.from_numpy(np.ascontiguousarray(mine)).to(dev)

    def step():
        if sharded_runner is not None:
            _frames, preds_local = sharded_runner(pristine, batch, metas)
            return preds_local
        if train_state is not None:
            return train_step()
        if graphed is not None:
            return graphed(pristine)            # copy-in + every kernel of the path = one graph replay
        work.copy_(pristine)                    # HunterJr corrects xyz in place: every step starts from the same bits
        bd = {'points': work, 'batch_size': batch, 'metadata': metas}
        with torch.no_grad():
            pred_dicts, _ = model(bd)
        return pred_dicts

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        preds = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_boxes = int(sum(p['pred_boxes'].shape[0] for p in preds))

    # instrumented pass (HIP events around the conv launches).  Training and agent-sharded steps contain collectives, so in those modes
    # every rank has to take part in the three extra steps; only rank 0 records.
    timer = None
    if rank == 0:
        graphed = None                          # the instrumented pass runs eagerly (events around individual launches)
        timer = ConvTimer()
        timer.install()
    if rank == 0 or ((args.train or args.shard == 'agent') and world > 1):
        for _ in range(3):
            step()
    if rank == 0:
        cs = timer.summary()
        timer.remove()
        frames = (world if args.shard == 'frame' else 1) * batch * args.steps
        line = {
            'metric': 'frames/sec (60k-pt cloud per agent) through the PointPillars hot path', 'value': round(frames / elapsed, 3),
            'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 4), 'higher_is_better': True, 'scaling': 'weak' if args.shard == 'frame' else 'strong', 'vs_baseline': None,
            'dtype': {'bf16x3': 'f32 tensors; 3x3 conv products as split bf16 (3 MFMAs, 16 mantissa bits), f32 accumulate [opt-in]',
                      'bf16': 'mixed precision [opt-in, --train only]: forward / data-gradient 3x3 conv products in bf16 (8 mantissa bits), f32 accumulate, '
                              'f32 master weights, weight gradients, BatchNorm, losses and optimizer'}.get(os.environ.get('PCP_CONV_ALGO', 'auto'), 'f32'),
            'data': 'synthetic',
            'config': {'workload': conf['name'] if not args.train else ('v2x_pointpillar_disco TRAINING iteration (3 frozen BEV makers + '
                       'trainable VFE/backbone/fusion/head forward+backward, CenterNet + distillation losses, clip, Adam one-cycle)'
                       if args.config == 'disco' else conf['name'] + ' -- TRAINING iteration (VFE/backbone/head forward+backward, CenterNet '
                       'losses, clip, Adam one-cycle)'),
                       'yaml': conf['yaml'], 'frames_per_gpu_per_step': batch,
                       'points_per_frame': int(pts_np.shape[0] // batch), 'parallelism': ('agent-sharded x%d: ragged all-gather of points%s, frames dealt to ranks' % (world, ' + all-gather of compressed BEV maps' if args.config == 'disco' else ''))
                       if args.shard == 'agent' else ('replicas x%d (frame-sharded)' % world) if not args.train else
                       ('data parallel x%d, one RCCL all-reduce of the flat fp32 gradient per step' % world), 'hipgraph': bool(args.graph),
                       'pipeline_mode': 'no per-pillar API tensors (their host sync), buffers kept across frames, first backbone layer ' +
                                        ('as the dense stride-2 conv on the canvas' if args.dense_first_layer else
                                         'from the pillar list when points <= 0.35 x cells (no dense canvas), dense otherwise') +
                                        '; outputs equal to the plugin-default path (tests/test_gpu_e2e.py::test_pipeline_mode_*)',
                       'peak_device_memory_mb': round(torch.cuda.max_memory_allocated(dev) / 2 ** 20, 1),
                       'final_boxes_last_step': n_boxes, **({'loss_last_step': train_state['last_loss']} if args.train else {})},
            'roofline': {'bound': 'mfma', 'kernel': timer.dominant,
                         'achieved': round(cs['tflops'], 3), 'peak': timer.peak, 'unit': 'TFLOP/s',
                         'frac': round(cs['tflops'] / timer.peak, 4),
                         'traffic': (pmc_traffic('k_w4_gemm' if 'k_w4_gemm' in timer.dominant else 'k_conv3x3_wino<2>')
                                     if (args.config == 'car' and ('wino' in timer.dominant or 'k_w4_gemm' in timer.dominant)) else None),
                         # the fused Winograd F(2x2) kernel executes 16/36 of the direct convolution's multiply-adds: fraction of the MFMA
                         # peak in EXECUTED flops (what the matrix pipe actually sustains); 1:1 for the GEMM kernel
                         'executed_frac': round(cs['tflops'] * timer.exec_mult / timer.peak, 4), **cs['extra'],
                         'avg_launch_us': round(cs['avg_us'], 2), 'launches_per_step': cs['launches'] // 3,
                         'all_conv3x3_tflops': round(cs['all_conv_tflops'], 3), 'all_conv3x3_ms_per_step': round(cs['all_conv_ms'] / 3, 3)},
        }
        if (world == 1 and not args.train and not args.graph and args.shard == 'frame' and os.environ.get('PCP_CONV_ALGO', 'auto') == 'auto'
                and not args.no_optin):
            # informational: the same workload with the OPT-IN split-bf16 convolution arithmetic (never part of `value`)
            os.environ['PCP_CONV_ALGO'] = 'bf16x3'
            for m in model.modules():
                if hasattr(m, 'invalidate_packed'):
                    m.invalidate_packed()
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            alt = time.perf_counter() - t1
            os.environ['PCP_CONV_ALGO'] = 'auto'
            for m in model.modules():
                if hasattr(m, 'invalidate_packed'):
                    m.invalidate_packed()
            line['optin_bf16x3'] = {'value': round(batch * args.steps / alt, 3), 'unit': 'frames/s', 'ms_per_step': round(1e3 * alt / args.steps, 4),
                                    'note': 'NOT the headline: 3x3 conv products as split bf16 (hi + lo, 16 mantissa bits; 3 bf16 MFMAs per '
                                            'product, f32 accumulate), ~1e-5 relative error, all parity tests pass at unchanged tolerances; '
                                            'enable with --conv-algo bf16x3 / PCP_CONV_ALGO=bf16x3'}
        if not args.no_cpu_baseline and world == 1:                    # rank 0 at N = 1 only (the contract); N > 1 lines carry null
            line['cpu_baseline'] = cpu_baseline(conf, cfg, state, pts_np, metas)
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
