"""bench.py -- frames/s of the PointPillars collaborative-perception hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config disco|car|ego|early|lately6] [--batch B] [--dist uniform|ring]
                    [--no-pipeline] [--pipeline-replicas R] [--layer-table FILE] [--host-input] [--no-secondary]

A "step" is one pass of the hot path (points resident in HBM -> final boxes) over one batch of B synthetic frames per GPU.
Default workload = the one BASELINE.json's metric is quoted on ("60k-pt cloud, 6 agents"): v2x_pointpillar_disco.yaml, mid fusion of
6 agents x 60 000 points per frame (3 BEV makers -> ego VFE/backbone -> compress / warp / fuse -> CenterHead -> decode -> rotated NMS),
B = BATCH_SIZE_PER_GPU = 4 frames.  It fits one GPU (5.7 GB).  The other BASELINE configs are --config car | ego | early | lately6.

Inference steps are software-pipelined by default (pcdet/models/pipelined.py, lately_chain.PipelinedChain): the kernels of step i+1 are
queued before the box counts of step i are read on the host, the BEV-maker streams of step i+1 start at that step's points, and steps
alternate between --pipeline-replicas (2) copies of the model on their own HIP streams.  All K steps AND their K host reads lie inside the
timed region (the last read is flushed before the closing synchronize); every step's detections are bit-identical to the batch-by-batch
ones (GPU tests).  --no-pipeline measures batch by batch: every step ends in its own host read.  --layer-table FILE writes the per-shape
kernel table of the instrumented pass.  --host-input uploads the batch from pinned host memory every step (the PCIe-inclusive rate,
informational); the default line also carries `secondary_ring`, the same command on the LiDAR-like cloud (--no-secondary skips it).
Per-launch times of the roofline blocks are HIP-event brackets less what an empty event pair reads in this process (raw: *_uncorrected).

--gpus N > 1: bench.py starts N ranks ITSELF (a `python -m torch.distributed.run` child, spawned before this process touches the GPU)
unless it already runs under a launcher (WORLD_SIZE set; it then insists on WORLD_SIZE == N).  One rank per GPU, backend nccl (= RCCL).
Frames are independent, so ranks are replicas on different frames (weak scaling, no data-path collective); value = all ranks' frames /
max-over-ranks time.  --shard agent / --train add the collectives those modes need.

Rank 0 prints ONE JSON line with the contract keys plus
  roofline     : the kernel with the LARGEST measured share of the step (HIP events around every C-ABI launch of an instrumented pass),
                 `achieved` = the flops that kernel EXECUTES on the matrix pipe / its measured time (never above the peak); the
                 direct-convolution-equivalent ("algorithmic") rate is a separate key; `traffic` from the committed rocprofv3 --pmc pass
  kernel_ms_per_step : the measured time of every kernel family of the step (the table `roofline.kernel` is picked from)
  cpu_baseline : the oracle (CPU restatement of the reference modules) timed on this box's host cores, bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, 'practical-collab-perception_amd')
for _p in (REPO, PKG, os.path.join(REPO, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

CONFIGS = {
    'car': dict(yaml='v2x_pointpillar_basic_car.yaml', layout='car', agents_in_cloud=1,
                name='v2x_pointpillar_basic_car single-agent inference (VFE+scatter+backbone+HunterJr+CenterHead+NMS), 1 x 60k points per frame'),
    'ego': dict(yaml='v2x_pointpillar_basic_ego.yaml', layout='lately', agents_in_cloud=1,
                name='v2x_pointpillar_basic_ego lately-fusion ego pass only (60k points incl. 300 synthetic MoDAR rows)'),
    'early': dict(yaml='v2x_pointpillar_basic_ego_early.yaml', layout='early', agents_in_cloud=6,
                  name='v2x_pointpillar_basic_ego_early early fusion (6 agents x 60k points merged per frame)'),
    'disco': dict(yaml='v2x_pointpillar_disco.yaml', layout='disco', agents_in_cloud=6,
                  name='v2x_pointpillar_disco mid fusion, 6 agents x 60k points per frame (3 BEV makers + ego branch + compress/warp/fuse + '
                       'CenterHead + NMS)'),
    'lately6': dict(yaml='v2x_pointpillar_basic_ego.yaml', layout='lately6', agents_in_cloud=6,
                    name='lately fusion end to end, 6 agents x 60k points per frame on one GPU: 5 remote basic_car passes -> MoDAR + foreground '
                         'rows -> device-side ingestion -> basic_ego pass'),
}
MFMA_F32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32 / 16x16x4_f32)
MFMA_BF16_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='disco', choices=sorted(CONFIGS))
    ap.add_argument('--batch', type=int, default=0, help='frames per GPU per step (0 = BATCH_SIZE_PER_GPU of the YAML)')
    ap.add_argument('--dist', default='uniform', choices=['uniform', 'ring'], help='synthetic cloud distribution (SURVEY 8(d)): uniform in x, y '
                    '(about 53.6k pillars per 60k points) or the LiDAR-like ring (r = 70 u^2; 20-30k pillars)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the informational second measurement on the LiDAR-like cloud (--dist ring)')
    ap.add_argument('--no-configs', action='store_true', help='skip the `configs` block (the default N = 1 disco line also times BASELINE configs 1 - 4 and '
                    "config 5's bf16 training loop, each in a bounded child process)")
    ap.add_argument('--cpu-baseline-budget', type=float, default=25.0, help='seconds of host-core time the oracle may spend on the cpu_baseline sample')
    ap.add_argument('--dense-first-layer', action='store_true', help='A/B switch: always write the dense canvas and run the first backbone '
                    'layer as the dense stride-2 conv (default in pipeline mode: from the pillar list when the cloud is sparse)')
    ap.add_argument('--plugin-default', action='store_true', help='measure the mode tools/test.py gets WITHOUT --fast: per-pillar API tensors '
                    'materialised (one host sync per VFE), dense canvas, no buffer reuse')
    ap.add_argument('--no-overlap', action='store_true', help='run the BEV-maker passes of a DiscoNet forward one after the other on the '
                                                                'main stream (the pipeline mode overlaps them on side streams)')
    ap.add_argument('--no-pipeline', action='store_true', help='inference (car / ego / early / disco): run batch by batch, every '
                    "step ending in its own host read, instead of pcdet/models/pipelined.py (the next step's kernels are queued before the "
                    "previous step's box counts are read; all K steps and their K reads still lie inside the timed region)")
    ap.add_argument('--pipeline-replicas', type=int, default=2, help='pipelined inference: 2 = consecutive steps alternate between the model and a '
                    'deep copy of it on two HIP streams (step i+1 may run beside step i); 1 = one model, steps in order on one stream')
    ap.add_argument('--pipeline-graph', action='store_true', help='pipelined inference: each replica replays its whole forward as ONE hipGraph per step '
                    '(captured once per buffer / batch size / pose set; the host then spends ~0.3 ms per step instead of enqueueing ~210 launches); '
                    'bitwise the eager pipelined detections')
    ap.add_argument('--elide-dead-makers', action='store_true', help='DiscoNet inference: skip the BEV-maker passes whose output nothing reads '
                    '(reference quirk F3: the rsu map is overwritten by the car maker, bev_img_early feeds only the training loss); pred_dicts are '
                    'bit-identical; reported under its own metric name, never the headline')
    ap.add_argument('--layer-table', default=None, help='write the per-shape table of the instrumented pass (kernel, shape, launches per step, '
                    'us per launch, executed TFLOP/s) to this file')
    ap.add_argument('--host-input', action='store_true', help='informational: the batch lives in PINNED HOST memory and is uploaded over PCIe every '
                    'step (asynchronously, on the pipelined runner\'s side stream) instead of being resident in HBM -- the PCIe-inclusive rate '
                    'of DESIGN 4; never the headline')
    ap.add_argument('--optin', action='store_true', help='also time the same workload with the OPT-IN split-bf16 conv arithmetic (informational)')
    ap.add_argument('--graph', action='store_true', help='replay the whole forward as one hipGraph (launch-bound small batches)')
    ap.add_argument('--conv-algo', default=None, choices=['auto', 'direct', 'winograd', 'winograd4', 'winograd4f', 'winograd4h', 'winograd4c', 'bf16x3', 'bf16'],
                    help='3x3 convolution arithmetic (default auto = fp32 MFMA: direct / Winograd).  bf16x3 is the OPT-IN split-bf16 mode '
                         '(three bf16 MFMAs per product, fp32 accumulate, ~1e-5 relative error); the JSON line then says so in `dtype`')
    ap.add_argument('--shard', default='frame', choices=['frame', 'agent'], help="frame (default): every rank is a replica on its own frames; "
                    "agent: configs early / disco only -- each rank holds the points of ITS agents, one all-gather of raw points (early) or "
                    "of compressed BEV maps (disco) per step, then the frames of the batch are dealt to the ranks (pcdet/models/sharded.py; "
                    "strong scaling: the group processes ONE batch per step)")
    ap.add_argument('--latency', type=int, default=0, help='B = 1 latency mode (SURVEY 8(d)): time N single-frame forwards one by one, points '
                    'resident in HBM -> final boxes readable on the host (every forward ends in its host read; no cross-frame pipelining, no '
                    'second replica), and print p50 / p99 / mean ms per frame as the JSON line; --graph replays the forward as one hipGraph '
                    'where the config supports it')
    ap.add_argument('--train', action='store_true', help='configs ego / early / disco: time full training iterations (forward + backward + '
                    'clip + fused Adam one-cycle step; data parallel over ranks with one RCCL all-reduce of the flat gradient)')
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """--gpus N outside a launcher: start N ranks as a torch.distributed.run CHILD process and return its exit code.  This parent has not
    touched the GPU (no torch.cuda call yet), and it does not exec: it waits for the child and hands its output through."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.call(cmd, env=env)


def load_cfg(yaml_name):
    from pcdet.config import EasyDict, cfg_from_yaml_file
    cfg = cfg_from_yaml_file(os.path.join(PKG, 'tools', 'cfgs', 'v2x_sim_models', yaml_name), EasyDict())
    for key in ('BEV_MAKER_RSU', 'BEV_MAKER_CAR', 'BEV_MAKER_EARLY'):
        if cfg.MODEL.get(key, None) is not None:
            cfg.MODEL[key].CKPT = None          # random-init weights: no checkpoints offline
    return cfg


def build_model(cfg):
    import torch
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


def set_pipeline_mode(model, overlap=True, dense_first_layer=False):
    """the mode every bench.py line without --plugin-default measures (tests/test_gpu_bench_mode.py pins it on reference fixtures through this
    very function): no per-pillar API tensors, buffers kept across frames, the first backbone layer from the pillar list where the cloud is
    sparse, the frozen BEV-maker passes of a DiscoNet forward on their own HIP streams.  Returns whether the makers are overlapped."""
    overlapped = False
    if overlap and hasattr(model, 'overlap_makers') and any(type(m).__name__ == 'BEVMaker' for m in model.module_list):
        overlapped = model.overlap_makers = True                 # frozen BEV-maker passes on their own HIP streams, joined in front of the fusion module
    for m in model.modules():
        if hasattr(m, 'materialize_pillars'):
            m.materialize_pillars = False       # per-pillar API tensors are not consumed downstream
            m.reuse_buffers = True
            m.sparse_first_layer = not dense_first_layer   # sparse clouds: first backbone layer from the pillar list, no dense canvas
    return overlapped


def make_points(conf, batch, rank, dist='uniform'):
    """B frames per rank; frame f of rank r uses agent streams 1000*r + 10*f + a (distinct data on every rank)."""
    import numpy as np
    from pcp_amd import synth
    clouds, metas = [], []
    layout = 'car' if conf['layout'] == 'lately6' else conf['layout']
    for f in range(batch):
        parts = []
        for a in range(conf['agents_in_cloud']):
            c = synth.agent_cloud(agent=1000 * rank + 10 * f + a, n_points=60000, layout=layout, dist=dist)
            if layout == 'disco':
                c[:, -1] = float(a)
            parts.append(c)
        clouds.append(np.concatenate(parts, 0))
        metas.append({'se3_from_ego': {a: synth.agent_pose(a) for a in range(conf['agents_in_cloud']) if a != 1}})
    return synth.collate(clouds), metas


def make_gt_boxes(batch, rank):
    """the ground-truth boxes of the training steps bench.py times: (batch, 40, 8) float32, frame f holds 40 - 3 f boxes [x, y, z, dx, dy, dz,
    heading, class 1], the rest are zero rows (tests/golden/make_golden.py g7fb4 feeds the same boxes to the reference's train step)"""
    import numpy as np
    from pcp_amd import synth
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
    return gt


def cpu_baseline(conf, cfg, state, batch_points, metas, budget_s=25.0):
    """oracle forward on the host cores: 1 frame per run (B=1), as many runs as fit the budget (at least 1)."""
    import torch
    from oracle import model as omodel
    from pcdet.config import EasyDict  # noqa: F401

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
                sample='%d x 1 frame (%d points), oracle/model.py forward incl. decode+NMS, torch CPU threads=%d' % (runs, pts.shape[0], cores),
                note='the oracle is the parity checker (numpy np.add.at / np.maximum.at scatter paths, torch-CPU convolutions), not a tuned CPU '
                     'implementation: per core it is slower than the reference\'s own torch modules, which BASELINE.md section 2 timed in the build '
                     'container on 8 cores (basic_car 0.76, basic_ego 2.9, early 0.70, disco ~0.2 frames/s)')


def cpu_baseline_lately(car_cfg, car_state, ego_cfg, ego_state, frame, budget_s=30.0):
    """config 3 on the host cores: the oracle's basic_car forward for each of the 5 remote agents, its ingestion, its basic_ego forward
    -- one frame per run, as many runs as fit the budget (at least 1)."""
    import numpy as np
    import torch
    from oracle import exchange as oex
    from oracle import model as omodel
    from pcp_amd import synth

    def plain(d):
        if isinstance(d, dict):
            return {k: plain(v) for k, v in d.items()}
        if isinstance(d, (list, tuple)):
            return [plain(v) for v in d]
        return d

    def arch_of(cfg):
        vs = [p.VOXEL_SIZE for p in cfg.DATA_CONFIG.DATA_PROCESSOR if 'VOXEL_SIZE' in p][0]
        return omodel.arch_from_cfg(plain(cfg.MODEL), list(cfg.DATA_CONFIG.POINT_CLOUD_RANGE), list(vs))
    car_arch, ego_arch = arch_of(car_cfg), arch_of(ego_cfg)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))
    torch.set_num_threads(cores)

    def one():
        e = frame['ego']
        ego_rows = np.zeros((e.shape[0], 13), np.float32)
        ego_rows[:, :5], ego_rows[:, 11:13] = e[:, :5], e[:, 5:7]
        parts = [ego_rows]
        for cloud, T in zip(frame['remote'], frame['target_se3_lidar']):
            out = omodel.forward(synth.collate([cloud]), car_state, car_arch, metadata=[{}])
            fb = out['final_box_dicts'][0]
            modar = np.concatenate([np.asarray(fb['pred_boxes']), np.asarray(fb['pred_scores'])[:, None],
                                    np.asarray(fb['pred_labels'], dtype=np.float32)[:, None]], 1).astype(np.float32)
            hj = out['hunter']
            fg, _ = oex.foreground_rows(np.asarray(hj['points']), np.asarray(hj['cls_logit']), np.asarray(hj['flow']))
            if modar.shape[0]:
                parts.append(oex.modar_ingest(modar, fg, T, frame['max_sweep_idx']))
        omodel.forward(synth.collate([np.concatenate(parts, 0)]), ego_state, ego_arch, metadata=[{}])
    t0 = time.time()
    one()
    warm = time.time() - t0
    runs, spent = 0, 0.0
    while runs < 1 or (spent + spent / max(runs, 1) < budget_s - warm and runs < 10):
        t1 = time.time()
        one()
        spent += time.time() - t1
        runs += 1
    return dict(value=round(runs / spent, 4), unit='frames/s', cores=cores, kind='port',
                sample='%d x 1 frame (6 agents x 60000 points: 5 oracle basic_car forwards + ingestion + 1 basic_ego forward), torch CPU threads=%d'
                       % (runs, cores))


# ---------------------------------------------------------------------------------------------------------------------------------------
# instrumented pass: HIP events around EVERY C-ABI launch on the launch stream (torch's current stream is the stream pcp_amd.ops passes)
# ---------------------------------------------------------------------------------------------------------------------------------------

class AbiTimer:
    """Wraps the ctypes library object so that each pcp_* launch is bracketed by two HIP events on the stream it is enqueued on.  A
    C-ABI entry is one kernel for all the heavy families (the multi-launch entries -- voxelize, NMS -- are latency work reported as a
    group); the three launches of the F(4x4) path are split by the library's own measurement entry point."""

    SKIP = ('_bytes', '_plan', 'pcp_abi_version', 'pcp_status_string', '_timed')

    def __init__(self):
        self.records = []           # (label, e0, e1, exec_flops, alg_flops, bound, peak)
        self.w4 = []                # (in_ms, gemm_ms, out_ms, gemm_flops, alg_flops)
        self.vfe_P, self.vfe_bytes = {}, 0.0
        import ctypes
        import torch
        self.rt = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
        self.rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]

    def install(self):
        import ctypes
        import torch
        from pcp_amd import lib
        self._lib = lib
        real = lib.load()
        self._real = real
        timer = self

        def describe(name, a):
            if name in ('pcp_conv3x3', 'pcp_conv3x3_winograd', 'pcp_conv3x3_bf16x3', 'pcp_conv3x3_bf16'):
                d = a[0]._obj
                s = d.stride
                ho, wo = (d.in_h - 1) // s + 1, (d.in_w - 1) // s + 1
                alg = 2.0 * d.batch * ho * wo * d.cout * 9 * d.cin
                if name == 'pcp_conv3x3_winograd':
                    v, fl = ctypes.c_int32(0), ctypes.c_double(0.0)
                    real.pcp_conv3x3_winograd_plan(ctypes.byref(d), ctypes.byref(v), ctypes.byref(fl))
                    names = {1: 'k_conv3x3_wino<1> (3x3 s1 fused Winograd F(2x2,3x3), 32-tile workgroups, v_mfma_f32_32x32x2_f32)',
                             2: 'k_conv3x3_wino<2> (3x3 s1 fused Winograd F(2x2,3x3), 64-tile workgroups, v_mfma_f32_32x32x2_f32)'}
                    return names.get(v.value, 'k_conv3x3_wino<%d> (3x3 s1 fused Winograd, fp32 MFMA)' % v.value), fl.value, alg, 'mfma', MFMA_F32_PEAK_TFLOPS
                if name == 'pcp_conv3x3':
                    return ('k_conv3x3_direct<s%d> (3x3 direct implicit GEMM, v_mfma_f32_32x32x2_f32)' % s,
                            2.0 * d.batch * ho * wo * d.cout_pad * 9 * d.cin, alg, 'mfma', MFMA_F32_PEAK_TFLOPS)
                mult = 3.0 if name == 'pcp_conv3x3_bf16x3' else 1.0
                return ('k_%s<s%d> [opt-in arithmetic]' % (name[4:], s), mult * 2.0 * d.batch * ho * wo * d.cout_pad * 9 * d.cin, alg,
                        'mfma', MFMA_BF16_PEAK_TFLOPS)
            if name == 'pcp_conv3x3_winograd4f':
                d = a[0]._obj
                fl = ctypes.c_double(0.0)
                real.pcp_conv3x3_winograd4f_plan(ctypes.byref(d), ctypes.byref(fl))
                return ('k_wino4f (3x3 s1 fused Winograd F(4x4,3x3), one 8-wave workgroup per CU, v_mfma_f32_32x32x2_f32)', fl.value,
                        2.0 * d.batch * d.in_h * d.in_w * d.cout * 9 * d.cin, 'mfma', MFMA_F32_PEAK_TFLOPS)
            if name == 'pcp_conv3x3_winograd4h':
                d = a[0]._obj
                fl = ctypes.c_double(0.0)
                real.pcp_conv3x3_winograd4h_plan(ctypes.byref(d), ctypes.byref(fl))
                return ('k_wino4h (3x3 s1 fused Winograd F(4x4,3x3), two 4-wave workgroups per CU, v_mfma_f32_16x16x4_f32)', fl.value,
                        2.0 * d.batch * d.in_h * d.in_w * d.cout * 9 * d.cin, 'mfma', MFMA_F32_PEAK_TFLOPS)
            if name == 'pcp_conv3x3_winograd4c':
                d = a[0]._obj
                fl = ctypes.c_double(0.0)
                real.pcp_conv3x3_winograd4c_plan(ctypes.byref(d), ctypes.byref(fl))
                return ('k_wino4c (3x3 s1 fused Winograd F(4x4,3x3), two 4-wave workgroups per CU, waves split over output channels, output '
                        'transform in registers, v_mfma_f32_16x16x4_f32)', fl.value, 2.0 * d.batch * d.in_h * d.in_w * d.cout * 9 * d.cin, 'mfma',
                        MFMA_F32_PEAK_TFLOPS)
            if name == 'pcp_mp_conv3x3':
                # the bf16 training loop's convolution (forward, data gradient, frozen teachers): v_mfma_f32_32x32x16_bf16, bf16 activations
                d = a[0]._obj
                s = d.stride
                ho, wo = (d.in_h - 1) // s + 1, (d.in_w - 1) // s + 1
                fast, fl = ctypes.c_int32(0), ctypes.c_double(0.0)
                real.pcp_mp_conv3x3_plan(ctypes.byref(d), ctypes.byref(fast), ctypes.byref(fl))
                lab = ('k_mp_conv3x3_s1 (3x3 s1, bf16 activations, persistent direct-to-LDS implicit GEMM, v_mfma_f32_32x32x16_bf16)' if fast.value
                       else 'k_mp_conv3x3_gen<s%d> (3x3, register-staged bf16 implicit GEMM, v_mfma_f32_32x32x16_bf16)' % s)
                return lab, fl.value, 2.0 * d.batch * ho * wo * d.cout * 9 * d.cin, 'mfma', MFMA_BF16_PEAK_TFLOPS
            if name == 'pcp_mp_conv3x3_wgrad':
                d = a[0]._obj
                s = d.stride
                ho, wo = d.in_h // s, d.in_w // s
                r64 = lambda v: (v + 63) // 64 * 64
                return ('k_mp_wgrad3x3 (3x3 weight gradient, bf16 pixel-contraction GEMM on transposed LDS reads, split over pixels, v_mfma_f32_32x32x16_bf16)',
                        2.0 * d.batch * ho * wo * r64(d.cout) * 9 * r64(d.cin), 2.0 * d.batch * ho * wo * d.cout * 9 * d.cin, 'mfma', MFMA_BF16_PEAK_TFLOPS)
            if name == 'pcp_conv3x3_wgrad':
                # pixel-contraction GEMM of the weight gradient: 64(co) x 64(ci) x 9-tap tiles, padding channels included in `executed`
                d = a[0]._obj
                s = d.stride
                ho, wo = d.in_h // s, d.in_w // s
                r64 = lambda v: (v + 63) // 64 * 64
                return ('pcp_conv3x3_wgrad (3x3 weight gradient, pixel-contraction GEMM, split-K, v_mfma_f32_32x32x2_f32)',
                        2.0 * d.batch * ho * wo * r64(d.cout) * 9 * r64(d.cin), 2.0 * d.batch * ho * wo * d.cout * 9 * d.cin, 'mfma', MFMA_F32_PEAK_TFLOPS)
            if name == 'pcp_pointwise_wgrad':
                ra, rb, rows = a[0]._obj, a[1]._obj, int(a[2])
                r64 = lambda v: (v + 63) // 64 * 64
                return ('pcp_pointwise_wgrad (1x1 / k2s2 weight gradient, pixel-contraction GEMM, fp32 MFMA)',
                        2.0 * rows * r64(ra.channels) * r64(rb.channels), 2.0 * rows * ra.channels * rb.channels, 'mfma', MFMA_F32_PEAK_TFLOPS)
            if name == 'pcp_pointwise':
                d = a[0]._obj
                if d.mode == 0:
                    fl, tag = 2.0 * d.rows * d.cin * d.cout_pad, 'plain'
                elif d.mode == 1:
                    fl, tag = 2.0 * d.batch * (d.in_h // 2) * (d.in_w // 2) * 4 * d.cin * d.cout_pad, 'conv_k2s2'
                else:
                    fl, tag = 2.0 * d.batch * d.in_h * d.in_w * d.cin * 4 * d.cout_pad, 'convT_k2s2'
                return 'k_pointwise<%s> (fp32 MFMA GEMM)' % tag, fl, fl * d.cout / max(d.cout_pad, 1), 'mfma', MFMA_F32_PEAK_TFLOPS
            label = {'pcp_pfn_scatter': 'k_pfn (fused PFN + scatter)', 'pcp_sparse_conv3x3_s2': 'k_sparse_conv_s2 (first backbone layer from the pillar list)',
                     'pcp_voxelize': 'pcp_voxelize (all its launches)', 'pcp_voxelize_cells_ready': 'pcp_voxelize (all its launches)',
                     'pcp_pillarise_rows': 'pcp_pillarise_rows (histogram + single-pass scan + rows in pillar order, all its launches)',
                     'pcp_pfn_rows': 'k_pfn_rows (fused PFN + scatter, one wave per ~30-point run of pillars)',
                     'pcp_select_transform_compact': 'pcp_select_transform_compact (agent selection + pose + compaction + cell ids, all its launches)',
                     'pcp_disco_weight_fuse': 'k_weight_fuse (DiscoNet pixel weightor + softmax + weighted sum, one launch)', 'pcp_nms_rotated': 'pcp_nms_rotated (all its launches)',
                     'pcp_hunter_point_head_ex': 'k_point_head', 'pcp_hunter_point_head': 'k_point_head',
                     'pcp_conv3x3_grouped_small': 'k_head_grouped'}.get(name, name)
            return label, None, None, 'latency', None

        def shape_of(name, a):
            d = getattr(a[0], '_obj', None) if a else None
            if d is None:
                return ''
            if name.startswith('pcp_conv3x3') or name.startswith('pcp_mp_conv3x3'):
                return 'B%d %dx%d %d->%d s%d' % (d.batch, d.in_h, d.in_w, d.cin, d.cout, getattr(d, 'stride', 1) or 1)
            if name == 'pcp_pointwise':
                return 'mode%d rows %d B%d %dx%d %d->%d' % (d.mode, d.rows, d.batch, d.in_h, d.in_w, d.cin, d.cout)
            return ''

        class Proxy:
            def __getattr__(self, name):
                fn = getattr(real, name)
                if not name.startswith('pcp_') or any(t in name for t in AbiTimer.SKIP):
                    return fn
                if name == 'pcp_conv3x3_winograd4':
                    def w4(d, x, u, bias, out, ws, stream):
                        ms = (ctypes.c_float * 3)()
                        fl = ctypes.c_double(0.0)
                        r = real.pcp_conv3x3_winograd4_timed(d, x, u, bias, out, ws, stream, ms, ctypes.byref(fl))
                        dd = d._obj
                        timer.w4.append((ms[0], ms[1], ms[2], fl.value, 2.0 * dd.batch * dd.in_h * dd.in_w * dd.cout * 9 * dd.cin))
                        return r
                    return w4

                def wrapped(*a):
                    s = torch.cuda.current_stream()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                    r = fn(*a)
                    e1.record(s)
                    timer.records.append((e0, e1) + describe(name, a) + (shape_of(name, a),))
                    # VFE stage: algorithmic bytes = points once + P coords + the canvas (dense) or the P pillar rows
                    if name in ('pcp_voxelize', 'pcp_voxelize_cells_ready', 'pcp_pillarise_rows'):
                        cptr = a[{'pcp_voxelize': 9, 'pcp_voxelize_cells_ready': 8, 'pcp_pillarise_rows': 10}[name]]
                        s.synchronize()                 # between two event pairs: no launch's measured duration contains this wait
                        host = (ctypes.c_int32 * 4)()
                        timer.rt.hipMemcpy(host, cptr, 16, 2)
                        timer.vfe_P[s.cuda_stream] = int(host[0])
                        timer.vfe_P[(s.cuda_stream, 'stride')] = int(a[2])
                    elif name == 'pcp_pfn_rows':
                        n, g = int(a[2]), a[0]._obj
                        P, stride = timer.vfe_P.get(s.cuda_stream, 0), timer.vfe_P.get((s.cuda_stream, 'stride'), 0)
                        canvas = a[9]
                        dense = bool(getattr(canvas, 'value', canvas))
                        timer.vfe_bytes += 4.0 * n * stride + 16.0 * P + (4.0 * g.batch_size * g.ny * g.nx * 64 if dense else 4.0 * 64 * P)
                    elif name == 'pcp_pfn_scatter':
                        n, stride, g = int(a[1]), int(a[2]), a[4]._obj
                        P = timer.vfe_P.get(s.cuda_stream, 0)
                        canvas = a[11]
                        dense = bool(getattr(canvas, 'value', canvas))
                        timer.vfe_bytes += 4.0 * n * stride + 16.0 * P + (4.0 * g.batch_size * g.ny * g.nx * 64 if dense else 4.0 * 64 * P)
                    return r
                return wrapped
        lib._LIB = Proxy()

    def remove(self):
        self._lib._LIB = self._real

    def families(self, steps):
        """-> list of dicts sorted by time: label, ms_per_step, launches_per_step, executed/algorithmic flops per step, bound, peak"""
        import torch
        torch.cuda.synchronize()
        # What an event pair itself reads with NOTHING between the two records (tools/event_overhead.py: 4.5 us on MI355X, p10 - p90
        # 4.44 - 4.60, the same on an idle and on a busy queue; a one-element fill between them reads 6.2 us).  It is in every bracketed
        # launch's elapsed time and not in the kernel's duration as rocprofv3 reports it (k_wino4c: 103.7 us bracketed, 99.7 us in the
        # rocprofv3 summary of the same run), so it is measured here, in this process, and taken off every launch; the raw sums are kept.
        s = torch.cuda.current_stream()
        trials = []
        for _ in range(64):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            e1.record(s)
            trials.append((e0, e1))
        torch.cuda.synchronize()
        self.bracket_ms = float(sorted(a.elapsed_time(b) for a, b in trials)[len(trials) // 2])
        fam = {}

        def add(label, ms_raw, ex, alg, bound, peak):
            f = fam.setdefault(label, dict(kernel=label, ms=0.0, ms_raw=0.0, launches=0, exec_flops=0.0, alg_flops=0.0, bound=bound, peak=peak))
            f['ms'] += max(ms_raw - self.bracket_ms, 0.0)
            f['ms_raw'] += ms_raw
            f['launches'] += 1
            f['exec_flops'] += ex or 0.0
            f['alg_flops'] += alg or 0.0
        self.layers = {}
        for (e0, e1, label, ex, alg, bound, peak, shape) in self.records:
            ms = e0.elapsed_time(e1)
            add(label, ms, ex, alg, bound, peak)
            if shape:
                L = self.layers.setdefault((label.split(' ')[0], shape), [0.0, 0, 0.0, 0.0])
                L[0] += max(ms - self.bracket_ms, 0.0)
                L[1] += 1
                L[2] += ex or 0.0
                L[3] += alg or 0.0
        for (a, b, c, gf, alg) in self.w4:
            add('k_w4_input (F(4x4,3x3) input transform)', a, None, None, 'hbm', None)
            add('k_w4_gemm (36 batched GEMMs of the Winograd F(4x4,3x3) wide layers, 128x128x32 LDS tiles, v_mfma_f32_32x32x2_f32)', b, gf, alg,
                'mfma', MFMA_F32_PEAK_TFLOPS)
            add('k_w4_output (F(4x4,3x3) output transform + bias + ReLU)', c, None, None, 'hbm', None)
        out = sorted(fam.values(), key=lambda f: -f['ms'])
        for f in out:
            f['ms_per_step'] = f['ms'] / steps
            f['ms_raw_per_step'] = f['ms_raw'] / steps
            f['launches_per_step'] = f['launches'] / steps
        return out


PROFILE_ROUND = 'r06'
PMC_TRAFFIC_JSON = os.path.join(REPO, 'profiles', PROFILE_ROUND + '_pmc_traffic.json')


def pmc_traffic(config, kernel_label):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of THIS command (separate FETCH_SIZE and
    WRITE_SIZE runs, --kernel-trace only; FETCH_SIZE doubled per MI355X_MICROARCH.md for 16-B/lane streaming reads), written by
    tools/pmc_summary.py together with the SHA-256 of the kernel's source file.  A kernel edited after its PMC pass gets `traffic` null and a
    note instead of stale bytes.  Returns (bytes_per_launch | None, note | None)."""
    import hashlib
    if not os.path.isfile(PMC_TRAFFIC_JSON):
        return None, 'no PMC pass committed for this round'
    with open(PMC_TRAFFIC_JSON) as f:
        d = json.load(f).get(config, {})
    short = kernel_label.split(' ')[0]
    e = d.get(short)
    if e is None:
        return None, 'no PMC entry for %s / %s' % (config, short)
    src = e.get('source')
    path = os.path.join(PKG, 'csrc', src) if src else None
    if not path or not os.path.isfile(path) or not e.get('source_sha256'):
        return None, 'PMC entry carries no source hash'
    with open(path, 'rb') as f:
        if hashlib.sha256(f.read()).hexdigest() != e['source_sha256']:
            return None, 'stale: csrc/%s changed after the PMC pass in %s' % (src, os.path.basename(PMC_TRAFFIC_JSON))
    return e['bytes_per_launch'], None


def hip_current_device():
    """the device the HIP runtime of THIS process launches on (what every C-ABI call of libpcp_hip.so uses), asked of the runtime itself"""
    import ctypes
    import torch
    rt = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
    d = ctypes.c_int(-1)
    if rt.hipGetDevice(ctypes.byref(d)) != 0:
        return -1
    return int(d.value)


def hip_device_identity():
    """the PHYSICAL device this process launches on, as an integer made of its PCI domain:bus:device.function -- ordinals cannot tell two
    ranks apart when the launcher gives every rank its own one-device view (ROCR_ / HIP_ / CUDA_VISIBLE_DEVICES per rank: all see ordinal 0)"""
    import ctypes
    import torch
    rt = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
    d = ctypes.c_int(-1)
    buf = ctypes.create_string_buffer(64)
    if rt.hipGetDevice(ctypes.byref(d)) != 0 or rt.hipDeviceGetPCIBusId(buf, 64, d) != 0:
        return -1
    try:
        dom, bus, rest = buf.value.decode().split(':')
        devn, fn = rest.split('.')
        return (int(dom, 16) << 16) | (int(bus, 16) << 8) | (int(devn, 16) << 3) | int(fn, 16)
    except ValueError:
        return -1


from pcp_amd.hostcpu import cpu_list as _cpu_list, pin_rank_to_cpus  # noqa: E402  (no torch, no GPU: plain os / sysfs)


# BASELINE.json's other workloads, each timed by a bounded child run of this script (a fresh process started with subprocess -- never an exec)
CONFIG_CHILDREN = [
    ('car', ['--config', 'car'], 10.0, 'config 2 (and config 1 = its cpu_baseline: the CPU path of the same workload on this box)'),
    ('ego', ['--config', 'ego'], 6.0, 'config 3, ego pass only'),
    ('lately6', ['--config', 'lately6'], 12.0, 'config 3 end to end on one GPU (5 remote basic_car passes -> MoDAR ingestion -> basic_ego pass)'),
    ('early', ['--config', 'early'], 6.0, 'config 4 on one GPU (the union of 6 clouds per frame)'),
    ('disco_train_bf16', ['--config', 'disco', '--train', '--conv-algo', 'bf16', '--steps', '10', '--warmup', '3'], 0.0,
     "config 5's training loop in bf16 (forward + backward + clip + Adam one-cycle step per iteration)"),
]


def summarize_line(r, note=None):
    """the entry of `configs` for one measured JSON line of this script"""
    roof = r.get('roofline') or {}
    hbm = r.get('roofline_hbm') or {}
    out = {'value': r['value'], 'unit': r['unit'], 'ms_per_step': r['ms_per_step'], 'steps': r['steps'], 'warmup': r['warmup'],
           'frames_per_step': (r.get('config') or {}).get('frames_per_gpu_per_step'), 'dtype': r['dtype'].split(' ')[0],
           'workload': (r.get('config') or {}).get('workload'),
           'dominant_kernel': (roof.get('kernel') or '').split(' ')[0] or None, 'roofline_bound': roof.get('bound'), 'roofline_frac': roof.get('frac'),
           'roofline_achieved': roof.get('achieved'), 'roofline_unit': roof.get('unit'),
           'vfe_stage_hbm_frac': hbm.get('frac'), 'vfe_stage_ms_per_step': hbm.get('ms_per_step'),
           'host_cpu_ms_per_step': r.get('host_cpu_ms_per_step'), 'final_boxes_last_step': (r.get('config') or {}).get('final_boxes_last_step'),
           'cpu_baseline': r.get('cpu_baseline')}
    if 'loss_last_step' in (r.get('config') or {}):
        out['loss_last_step'] = r['config']['loss_last_step']
    if note:
        out['note'] = note
    return out


def run_config_children(steps, warmup, timeout_s=120):
    out = {}
    for name, extra, cpu_budget, note in CONFIG_CHILDREN:
        cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(min(steps, 20)), '--warmup', str(min(warmup, 5))] + extra + \
              ['--no-secondary', '--no-configs'] + (['--cpu-baseline-budget', str(cpu_budget)] if cpu_budget > 0 else ['--no-cpu-baseline'])
        t0 = time.time()
        try:
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, check=False)
            rl = [ln for ln in res.stdout.decode().splitlines() if ln.startswith('{')]
            if res.returncode != 0 or not rl:
                out[name] = {'error': 'child exited with %d: %s' % (res.returncode, res.stderr.decode()[-300:])}
                continue
            out[name] = summarize_line(json.loads(rl[-1]), note)
            out[name]['child_wall_s'] = round(time.time() - t0, 1)
        except Exception as e:                                         # the headline line must not depend on these extras
            out[name] = {'error': repr(e)[:300]}
    return out


def dry_run(args, world, rank):
    """PCP_BENCH_DRY_RUN=1 (tests/test_dist_cpu.py only): the launcher, rendezvous, barrier and max-over-ranks aggregation of this script
    with a sleep in place of the kernels -- no GPU, no library, backend gloo.  The line says so in `data`; it is never a measurement."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='gloo')
    batch = args.batch or 4
    for _ in range(args.warmup):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (1 + rank))
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    seen = 1
    layout = {'workload': 'dry run'}
    mine = pin_rank_to_cpus(int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('LOCAL_WORLD_SIZE', str(world))))
    if world > 1:
        masks = [None] * world
        dist.all_gather_object(masks, _cpu_list(mine or []))
        layout['rank_cpu_affinity'] = masks
    else:
        layout['rank_cpu_affinity'] = [_cpu_list(mine or [])]
    if world > 1 and args.shard == 'agent':
        # the row split of --shard agent (agent % world) and the exchanges of the sharded runners on CPU tensors: ranks without an
        # agent contribute empty row blocks, ranks without a frame still take part in every collective
        from pcdet.utils import v2x_exchange as ex
        agents = [a for a in range(6) if a % world == rank]
        rows = torch.full((4 * len(agents), 3), float(rank))
        _union, counts = ex.all_gather_v_rows(rows)
        frames = torch.tensor([len(ex.shard_frames(batch, world, rank))])
        per = [torch.zeros_like(frames) for _ in range(world)]
        dist.all_gather(per, frames)
        layout.update(rows_per_rank=[int(c) for c in counts], frames_per_rank=[int(p.item()) for p in per])
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(c)
        seen = int(c.item())
    if rank == 0:
        print(json.dumps({'metric': 'frames/sec', 'value': round(world * batch * args.steps / elapsed, 3), 'unit': 'frames/s', 'n_gpus': world,
                          'ranks_seen_by_collective': seen, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': round(1e3 * elapsed / args.steps, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                          'dtype': 'f32', 'data': 'DRY RUN: no kernels executed (launcher / aggregation self-test)',
                          'config': layout, 'roofline': None, 'cpu_baseline': None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))            # nothing in this process has initialised the GPU
    world = int(env_world or '1')
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node equal to --gpus, or let bench.py start the ranks)'
                         % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('PCP_BENCH_DRY_RUN') == '1':
        return dry_run(args, world, rank)
    my_cpus = pin_rank_to_cpus(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', str(world))))      # before anything touches the GPU

    import numpy as np
    import torch
    import torch.distributed as dist
    if world > 1 and my_cpus:
        torch.set_num_threads(max(1, min(torch.get_num_threads(), len(my_cpus))))

    if args.conv_algo is not None:
        if args.conv_algo == 'bf16' and not args.train:
            raise SystemExit('--conv-algo bf16 (plain bf16 products) is the mixed-precision TRAINING mode: add --train')
        os.environ['PCP_CONV_ALGO'] = args.conv_algo
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (the hot path has no CPU fallback)'
    # one rank per GPU; PCP_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than ranks
    # (ranks then share devices round-robin -- a functional check, not a measurement, and the line says so)
    n_dev = torch.cuda.device_count()
    backend = os.environ.get('PCP_BENCH_BACKEND', 'nccl')
    # a launcher may hand every rank a one-device view (per-rank ROCR_ / HIP_ / CUDA_VISIBLE_DEVICES): then each rank uses ITS ordinal 0 and
    # the distinct-device check below goes by PCI address
    isolated = (world > 1 and n_dev == 1 and any(os.environ.get(k) not in (None, '') for k in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES',
                                                                                                   'CUDA_VISIBLE_DEVICES')))
    if world > n_dev and backend == 'nccl' and not isolated:
        raise SystemExit('bench.py: %d ranks but %d GPUs visible (one rank per GPU; PCP_BENCH_BACKEND=gloo shares devices for a functional check)'
                         % (world, n_dev))
    dev_index = local_rank % n_dev if world > 1 else 0
    ranks_seen = 1
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(dev_index)
        dist.init_process_group(backend=backend)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        c = torch.ones(1, dtype=torch.int64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(c)                         # the world size as the collective library itself observes it
        ranks_seen = int(c.item())

    conf = CONFIGS[args.config]
    cfg = load_cfg(conf['yaml'])
    batch = args.batch or int(cfg.OPTIMIZATION.BATCH_SIZE_PER_GPU)
    model, state, ds = build_model(cfg)
    if args.train and args.config == 'lately6':
        raise SystemExit('--train: use --config car | ego | early | disco')
    model = model.to(dev).eval()
    overlapped = False
    if args.elide_dead_makers:
        if args.train or not hasattr(model, 'elide_dead_makers'):
            raise SystemExit('--elide-dead-makers: DiscoNet inference only')
        model.elide_dead_makers = True
    if not args.plugin_default:
        overlapped = set_pipeline_mode(model, overlap=not args.no_overlap, dense_first_layer=args.dense_first_layer)
    pts_np, metas = make_points(conf, batch, rank, args.dist)
    pristine = torch.from_numpy(pts_np).to(dev)
    work = torch.empty_like(pristine)

    lately = None
    if args.config == 'lately6':
        # config 3 end to end: 5 remote agents run the basic_car detector on their own clouds (one stacked pass), their MoDAR + foreground
        # rows are ingested on the device into the ego cloud, the basic_ego detector runs on the result (pcdet/models/lately_chain.py)
        from pcdet.models.lately_chain import LatelyFusionChain
        from pcp_amd import synth
        car_cfg = load_cfg(CONFIGS['car']['yaml'])
        car_model, car_state, _ = build_model(car_cfg)
        car_model = car_model.to(dev).eval()
        lately = LatelyFusionChain(car_model, model, pipeline=not args.plugin_default)
        lately_frames = []
        for f in range(batch):
            cl = lambda a: synth.agent_cloud(agent=1000 * rank + 10 * f + a, n_points=60000, layout='car', dist=args.dist)
            lately_frames.append(dict(ego=cl(1), remote=[cl(a) for a in (0, 2, 3, 4, 5)],
                               target_se3_lidar=[np.linalg.inv(synth.agent_pose(a)) for a in (0, 2, 3, 4, 5)], max_sweep_idx=10.0))
        lately_inputs = LatelyFusionChain.build_inputs(lately_frames, dev)
        lately_pristine = lately_inputs['remote_points'].clone()

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
        gt = make_gt_boxes(batch, rank)
        train_state = dict(opt=opt, sched=sched, gt=torch.from_numpy(gt).to(dev), it=0)
        if args.config == 'car':
            # configs 1 / 2 train HunterJr: foreground points with (sweep, instance) columns and the per-sweep motion of every instance
            tf = np.zeros((batch, 40, 11, 3, 4), dtype=np.float32)
            tf[..., :3, :3] = np.eye(3, dtype=np.float32)
            extra = []
            for f in range(batch):
                fg, tf_f = synth.instance_foreground(1000 * rank + f, gt[f, :40 - 3 * f])
                tf[f, :tf_f.shape[0]] = tf_f
                extra.append(np.concatenate([np.full((fg.shape[0], 1), float(f), np.float32), fg], 1))
            pristine = torch.cat([pristine, torch.from_numpy(np.concatenate(extra, 0)).to(dev)], 0).contiguous()
            work = torch.empty_like(pristine)
            train_state['instances_tf'] = torch.from_numpy(tf).to(dev)

    def train_step():
        ts = train_state
        ts['sched'].step(ts['it'])
        if not model.training:
            model.train()
        ts['opt'].zero_grad()
        bd = {'points': pristine, 'batch_size': batch, 'metadata': metas, 'gt_boxes': ts['gt']}
        if 'instances_tf' in ts:
            work.copy_(pristine)                    # HunterJr corrects xyz in place
            bd.update(points=work, instances_tf=ts['instances_tf'])
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
        pts_np, metas = make_points(conf, batch, 0, args.dist)
        agent_of_row = pts_np[:, -1] if conf['layout'] == 'disco' else np.repeat(np.arange(conf['agents_in_cloud']), 60000)[None].repeat(batch, 0).reshape(-1)
        mine = pts_np[(agent_of_row.astype(np.int64) % world) == rank]
        pristine = torch.from_numpy(np.ascontiguousarray(mine)).to(dev)

    pipelined = None
    from pcdet.models.pipelined import PipelinedDetector
    if (not args.no_pipeline and not args.latency and not args.plugin_default and not args.train and not args.graph and args.shard == 'frame'
            and lately is None and PipelinedDetector.supports(model)):          # anything else (a head without a deferred finalize, ...) runs batch by batch
        shared_device = world > 1 and backend == 'gloo'            # functional check: several ranks on one GPU -- no second replica each
        if shared_device:
            args.pipeline_replicas = 1
        pipelined = PipelinedDetector(model, replicas=max(1, args.pipeline_replicas), graph=args.pipeline_graph)
        work_bufs = [work, torch.empty_like(pristine)]
        work.copy_(pristine)
        pipelined.prepare(work, batch, metas)       # setup: every replica builds its packed weights / buffers once, before the W warm-up steps
    host_pristine = None
    if args.host_input:
        if pipelined is None:
            raise SystemExit('--host-input measures the pipelined inference runner (car / ego / early / disco without --no-pipeline / --graph / --train)')
        host_pristine = pristine.cpu().pin_memory()
    pipe_state = {'n': 0}
    lately_pipe = None
    if (lately is not None and not args.no_pipeline and not args.latency and not args.plugin_default and not args.graph
            and not (world > 1 and backend == 'gloo')):
        from pcdet.models.lately_chain import PipelinedChain
        lately_pipe = PipelinedChain(lately, replicas=max(1, args.pipeline_replicas))
        lately_sets = [lately_inputs, dict(lately_inputs, remote_points=lately_pristine.clone())]      # two input sets, used alternately
        for st_ in lately_sets:
            st_['remote_points'].copy_(lately_pristine)
        lately_pipe.prepare(lately_sets)

    def step():
        if pipelined is not None:
            i = pipe_state['n']
            pipe_state['n'] = i + 1
            return pipelined.submit(work_bufs[i & 1], batch, metas, copy_from=host_pristine if host_pristine is not None else pristine)      # pred_dicts of the PREVIOUS step
        if sharded_runner is not None:
            _frames, preds_local = sharded_runner(pristine, batch, metas)
            return preds_local
        if train_state is not None:
            return train_step()
        if graphed is not None:
            return graphed(pristine)            # copy-in + every kernel of the path = one graph replay
        if lately_pipe is not None:
            i = pipe_state['n']
            pipe_state['n'] = i + 1
            cur_set = lately_sets[i & 1]
            cur_set['remote_points'].copy_(lately_pristine)           # HunterJr corrects xyz in place: every step starts from the same bits
            return lately_pipe.submit(cur_set)                        # pred_dicts of the PREVIOUS step
        if lately is not None:
            lately_inputs['remote_points'].copy_(lately_pristine)     # HunterJr corrects xyz in place: every step starts from the same bits
            return lately(lately_inputs)
        work.copy_(pristine)                    # HunterJr corrects xyz in place: every step starts from the same bits
        bd = {'points': work, 'batch_size': batch, 'metadata': metas}
        with torch.no_grad():
            pred_dicts, _ = model(bd)
        return pred_dicts

    if args.latency > 0:
        # ---- B = 1 latency: one frame at a time, every forward ends in its own host read -------------------------------------------------
        assert pipelined is None and lately_pipe is None and train_state is None and sharded_runner is None
        for _ in range(max(args.warmup, 5)):
            step()
        torch.cuda.synchronize()
        lat = []
        for _ in range(args.latency):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            preds = step()
            n_last = int(sum(p['pred_boxes'].shape[0] for p in preds))          # exact-shape tensors: the count was read on the host
            torch.cuda.synchronize()
            lat.append(1e3 * (time.perf_counter() - t1))
        lat = np.sort(np.array(lat))
        if rank == 0:
            print(json.dumps({'metric': 'latency per frame (60k-pt cloud, %d agent%s), B = %d' % (conf['agents_in_cloud'], 's' if conf['agents_in_cloud'] > 1 else '', batch),
                              'value': round(float(np.percentile(lat, 50)), 4), 'unit': 'ms', 'p50_ms': round(float(np.percentile(lat, 50)), 4),
                              'p99_ms': round(float(np.percentile(lat, 99)), 4), 'mean_ms': round(float(lat.mean()), 4), 'min_ms': round(float(lat[0]), 4),
                              'frames_timed': int(args.latency), 'n_gpus': 1, 'higher_is_better': False, 'dtype': 'f32', 'data': 'synthetic',
                              'config': {'workload': conf['name'], 'yaml': conf['yaml'], 'frames_per_step': batch, 'hipgraph': bool(args.graph),
                                         'mode': 'one forward at a time: points resident in HBM -> boxes on the host; no cross-frame pipelining, one model replica'
                                                 + ('; BEV-maker passes on their own HIP streams' if overlapped else ''),
                                         'final_boxes_last_frame': n_last}}))
        return
    for _ in range(args.warmup):
        step()
    if pipelined is not None:
        pipelined.flush()
    if lately_pipe is not None:
        lately_pipe.flush()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cpu_thread0, cpu_proc0 = time.thread_time(), time.process_time()
    host_s = 0.0                                # wall time the host spends inside step() (enqueueing AND, pipelined, blocking on the previous step's read)
    for _ in range(args.steps):
        h0 = time.perf_counter()
        preds = step()
        host_s += time.perf_counter() - h0
    if lately_pipe is not None:
        preds = lately_pipe.flush()
    if pipelined is not None:
        preds = pipelined.flush()               # the last step's host read: all K reads lie inside the timed region
    host_cpu_thread_s, host_cpu_proc_s = time.thread_time() - cpu_thread0, time.process_time() - cpu_proc0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank_ms, rank_devices, rank_pci = [round(1e3 * elapsed / args.steps, 4)], [hip_current_device()], [hip_device_identity()]
    rank_cpus = [_cpu_list(my_cpus or [])]
    if world > 1:
        rank_cpus = [None] * world
        dist.all_gather_object(rank_cpus, _cpu_list(my_cpus or []))
        # every rank's own time and the device its HIP runtime launches on (a straggler or two ranks on one device must be visible in the
        # line); `value` uses the MAX over ranks, as the contract says
        info = torch.tensor([elapsed, float(hip_current_device()), float(hip_device_identity())], dtype=torch.float64,
                            device=dev if backend == 'nccl' else 'cpu')
        gathered = [torch.zeros_like(info) for _ in range(world)]
        dist.all_gather(gathered, info)
        per_rank_ms = [round(1e3 * float(g[0]) / args.steps, 4) for g in gathered]
        rank_devices = [int(g[1]) for g in gathered]
        rank_pci = [int(g[2]) for g in gathered]
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if backend == 'nccl' and (len(set(rank_pci)) != world if min(rank_pci) >= 0 else sorted(rank_devices) != list(range(world))):
            raise SystemExit('bench.py: ranks do not sit on distinct devices: ordinals %s, PCI %s' % (rank_devices, rank_pci))
    n_boxes = int(sum(p['pred_boxes'].shape[0] for p in preds))

    # instrumented pass (HIP events around every C-ABI launch).  Training and agent-sharded steps contain collectives, so in those modes
    # every rank has to take part in the extra steps; only rank 0 records.
    if os.environ.get('PCP_OP_CENSUS') and rank == 0 and world == 1:
        # diagnostic: every ATen op two steps dispatch (the path's own kernels go through the C ABI and do not show here), by call site
        import collections
        import traceback
        from torch.utils._python_dispatch import TorchDispatchMode
        census = collections.Counter()

        class Census(TorchDispatchMode):
            def __torch_dispatch__(self, func, types, a=(), kw=None):
                site = '?'
                for fr in reversed(traceback.extract_stack(limit=14)[:-1]):
                    if '/torch/' not in fr.filename and fr.name != '__torch_dispatch__':
                        site = '%s:%d %s' % (os.path.relpath(fr.filename, REPO), fr.lineno, fr.name)
                        break
                census[(str(func), site)] += 1
                return func(*a, **(kw or {}))
        with Census():
            for _ in range(2):
                step()
        torch.cuda.synchronize()
        with open(os.environ['PCP_OP_CENSUS'], 'w') as f:
            for (op, site), n in sorted(census.items(), key=lambda kv: -kv[1]):
                f.write('%6.1f  %-38s %s\n' % (n / 2.0, op, site))

    INSTR_STEPS = 3
    timer = None
    was_pipelined = pipelined is not None or lately_pipe is not None
    pipelined = None                            # the instrumented pass runs batch by batch
    lately_pipe = None
    if rank == 0:
        graphed = None                          # the instrumented pass runs eagerly (events around individual launches)
        if getattr(model, 'overlap_makers', False):
            model.overlap_makers = False        # ... and on ONE stream: an event pair must not time other streams' kernels
    if rank == 0 or ((args.train or args.shard == 'agent') and world > 1):
        # one untimed step in the instrumented pass's own form first: the launch stream changes (the caller's stream instead of the
        # replicas'), so per-stream scratch buffers are allocated on first use -- a first step read 176 us for the 41-us agent compaction
        step()
        torch.cuda.synchronize()
    if rank == 0:
        timer = AbiTimer()
        timer.install()
    if rank == 0 or ((args.train or args.shard == 'agent') and world > 1):
        for _ in range(INSTR_STEPS):
            step()
    if rank == 0:
        fams = timer.families(INSTR_STEPS)
        timer.remove()
        if args.layer_table and rank == 0:
            with open(args.layer_table, 'w') as f:
                f.write('%-28s %-44s %9s %10s %9s %9s\n' % ('kernel', 'shape', 'per step', 'us/launch', 'exec TF', 'ms/step'))
                for (lab, shape), (ms, n, ex, alg) in sorted(timer.layers.items(), key=lambda kv: -kv[1][0]):
                    f.write('%-28s %-44s %9.2f %10.1f %9.1f %9.3f\n' % (lab, shape, n / INSTR_STEPS, ms / n * 1e3, ex / ms / 1e9 if ms else 0.0,
                                                                     ms / INSTR_STEPS))
        dom = fams[0]
        frames = (world if args.shard == 'frame' else 1) * batch * args.steps
        algo = os.environ.get('PCP_CONV_ALGO', 'auto')
        if dom['bound'] == 'mfma' and dom['ms'] > 0:
            ach = dom['exec_flops'] / (dom['ms'] * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'kernel': dom['kernel'], 'achieved': round(ach, 3), 'peak': dom['peak'], 'unit': 'TFLOP/s',
                    'frac': round(ach / dom['peak'], 4),
                    'frac_uncorrected': round(dom['exec_flops'] / (dom['ms_raw'] * 1e-3) / 1e12 / dom['peak'], 4),
                    'note': 'achieved = flops this kernel EXECUTES on the matrix pipe (Winograd F(2x2): 16 products per 2x2 tile, F(4x4): 36 per 4x4 tile, padding included) / '
                            'its HIP-event time, each launch less the time an EMPTY event pair reads in this process (event_bracket_overhead_us; '
                            'frac_uncorrected and avg_launch_us_uncorrected keep the raw bracket); algorithmic_tflops = flops of the equivalent direct '
                            'convolution / the same time',
                    'algorithmic_tflops': round(dom['alg_flops'] / (dom['ms'] * 1e-3) / 1e12, 3)}
        else:
            roof = {'bound': 'hbm' if dom['bound'] == 'hbm' else 'latency', 'kernel': dom['kernel'], 'achieved': None, 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': None}
        traffic, traffic_note = pmc_traffic(args.config, dom['kernel']) if (algo == 'auto' and not args.train) else (None, 'not collected for this mode')
        roof.update({'traffic': traffic, **({'traffic_note': traffic_note} if traffic_note else {}),
                     'avg_launch_us': round(1e3 * dom['ms'] / max(dom['launches'], 1), 2),
                     'avg_launch_us_uncorrected': round(1e3 * dom['ms_raw'] / max(dom['launches'], 1), 2),
                     'event_bracket_overhead_us': round(1e3 * timer.bracket_ms, 2),
                     'measured_in': 'the instrumented pass of this run: every launch between two HIP events on ONE stream, batch by batch (no '
                                    'maker overlap, no batch pipelining) -- compare with the rocprofv3 summary of `bench.py --no-overlap '
                                    '--no-pipeline` (profiles/' + PROFILE_ROUND + '_bench_disco_b4_single_stream_kernel_stats.csv), not with the overlapped run '
                                    '(profiles/' + PROFILE_ROUND + '_bench_disco_b4_overlapped_kernel_stats.csv), whose kernels share the chip',
                     'launches_per_step': round(dom['launches_per_step'], 2),
                     'share_of_kernel_time': round(dom['ms'] / max(sum(f['ms'] for f in fams), 1e-9), 4)})
        mf = [f for f in fams if f['bound'] == 'mfma']
        # the next matrix-pipe kernels of the step beside the dominant one (the two fused F(4x4) kernels share the 3x3 stride-1 layers)
        roof['next_mfma_kernels'] = [{'kernel': f['kernel'].split(' ')[0], 'ms_per_step': round(f['ms_per_step'], 4),
                                      'achieved': round(f['exec_flops'] / max(f['ms'], 1e-9) / 1e9, 3),
                                      'frac': round(f['exec_flops'] / max(f['ms'], 1e-9) / 1e9 / f['peak'], 4)} for f in mf if f is not dom][:3]
        roof['all_mfma_kernels'] = {'executed_tflops': round(sum(f['exec_flops'] for f in mf) / max(sum(f['ms'] for f in mf), 1e-9) / 1e9, 3),
                                    'algorithmic_tflops': round(sum(f['alg_flops'] for f in mf) / max(sum(f['ms'] for f in mf), 1e-9) / 1e9, 3),
                                    'ms_per_step': round(sum(f['ms_per_step'] for f in mf), 3)}
        # the HBM-bound VFE stage -- pillariser + fused PFN / scatter (+ canvas clear) -- against the HBM peak; (iii): the
        # latency-bound tail (decode + rotated NMS + gather) in microseconds per step
        vfe_ms = sum(f['ms_per_step'] for f in fams if f['kernel'].split(' ')[0] in ('pcp_voxelize', 'k_pfn', 'pcp_canvas_clear', 'pcp_select_transform_compact', 'pcp_pillarise_rows', 'k_pfn_rows'))
        vfe_ms_raw = sum(f['ms_raw_per_step'] for f in fams if f['kernel'].split(' ')[0] in ('pcp_voxelize', 'k_pfn', 'pcp_canvas_clear', 'pcp_select_transform_compact', 'pcp_pillarise_rows', 'k_pfn_rows'))
        vfe_bytes = timer.vfe_bytes / INSTR_STEPS
        vfe_roof = None
        if vfe_ms > 0 and vfe_bytes > 0:
            gbs = vfe_bytes / (vfe_ms * 1e-3) / 1e9
            vfe_roof = {'bound': 'hbm', 'stage': 'pillarise (+ agent selection) + fused PFN + scatter / pillar rows, all VFE passes of the step',
                        'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4),
                        'algorithmic_mb_per_step': round(vfe_bytes / 1e6, 2), 'ms_per_step': round(vfe_ms, 4),
                        'ms_per_step_uncorrected': round(vfe_ms_raw, 4), 'frac_uncorrected': round(vfe_bytes / (vfe_ms_raw * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        'note': 'algorithmic bytes (SURVEY 8(d) row 1): points read once + 16 B per pillar + the dense canvas or the 256-B pillar rows; '
                                'time = HIP events around the C-ABI calls of the instrumented single-stream pass, each less the empty event pair '
                                '(roofline.event_bracket_overhead_us); *_uncorrected keep the raw brackets'}
        decode_nms_us = round(1e3 * sum(f['ms_per_step'] for f in fams if f['kernel'].split(' ')[0] in
                                        ('pcp_centerhead_decode', 'pcp_nms_rotated', 'pcp_nms_normal', 'pcp_gather_detections', 'pcp_anchor_decode', 'pcp_topk_boxes')), 1)
        line = {
            'metric': ('frames/sec (60k-pt cloud, 6 agents)' if conf['agents_in_cloud'] == 6 else 'frames/sec (60k-pt cloud, 1 agent)') +
                      (' [dead BEV-maker passes elided: NOT the headline]' if args.elide_dead_makers else ''),
            'value': round(frames / elapsed, 3),
            'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 4), 'host_enqueue_ms_per_step': round(1e3 * host_s / args.steps, 4),
            # CPU time, not wall time: what the enqueueing Python thread burns per step (time.thread_time) and the whole process incl. the
            # HIP runtime's helper threads (time.process_time) -- the headroom a host core has beside the GPU's ms_per_step
            'host_cpu_ms_per_step': round(1e3 * host_cpu_thread_s / args.steps, 4),
            'host_cpu_all_threads_ms_per_step': round(1e3 * host_cpu_proc_s / args.steps, 4),
            'higher_is_better': True, 'scaling': 'weak' if args.shard == 'frame' else 'strong', 'vs_baseline': None,
            'dtype': {'bf16x3': 'f32 tensors; 3x3 conv products as split bf16 (3 MFMAs, 16 mantissa bits), f32 accumulate [opt-in]',
                      'bf16': 'bf16 training loop [--train only]: bf16 activation / gradient storage between the 3x3 layers, forward / data-gradient / '
                              'weight-gradient 3x3 convs (teachers included) on v_mfma_f32_32x32x16_bf16 with f32 accumulate; f32 master weights, BatchNorm '
                              'statistics, 1x1 / k2s2 layers, PFN, losses and optimizer'}.get(algo, 'f32'),
            'data': 'synthetic' + (' (ring distribution)' if args.dist == 'ring' else '')
                    + (' -- INFORMATIONAL: batch uploaded from pinned host memory every step (--host-input)' if args.host_input else ''),
            'config': {'workload': conf['name'] if not args.train else ('v2x_pointpillar_disco TRAINING iteration (3 frozen BEV makers + '
                       'trainable VFE/backbone/fusion/head forward+backward, CenterNet + distillation losses, clip, Adam one-cycle)'
                       if args.config == 'disco' else conf['name'] + ' -- TRAINING iteration (VFE/backbone/%shead forward+backward, CenterNet '
                       'losses, clip, Adam one-cycle)' % ('HunterJr incl. object head and its seven loss terms/' if args.config == 'car' else '')),
                       'yaml': conf['yaml'], 'frames_per_gpu_per_step': batch, 'agents_per_frame': conf['agents_in_cloud'],
                       'points_per_frame': int(pts_np.shape[0] // batch), 'point_distribution': args.dist,
                       'parallelism': ('agent-sharded x%d: ragged all-gather of points%s, frames dealt to ranks' % (world, ' + all-gather of compressed BEV maps' if args.config == 'disco' else ''))
                       if args.shard == 'agent' else ('replicas x%d (frame-sharded)' % world) if not args.train else
                       ('data parallel x%d, one RCCL all-reduce of the flat fp32 gradient per step' % world), 'hipgraph': bool(args.graph or (was_pipelined and args.pipeline_graph)),
                       'ranks_seen_by_collective': ranks_seen, 'backend': backend if world > 1 else None,
                       'per_rank_ms_per_step': per_rank_ms, 'rank_devices': rank_devices, 'rank_cpu_affinity': rank_cpus, 'rank_pci_addresses': ['%04x:%02x:%02x.%x' % (v >> 16, (v >> 8) & 255, (v >> 3) & 31, v & 7) if v >= 0 else None for v in rank_pci],
                       'mode': ('plugin default (per-pillar API tensors materialised: one host sync per VFE; dense canvas)' if args.plugin_default else
                                'pipeline: no per-pillar API tensors (their host sync), buffers kept across frames, first backbone layer ' +
                                ('as the dense stride-2 conv on the canvas' if args.dense_first_layer else
                                 'from the pillar list when points <= 0.35 x cells (no dense canvas), dense otherwise') +
                                ('; the frozen BEV-maker passes run on their own HIP streams and join in front of the fusion module '
                                 '(kernel_ms_per_step and roofline come from an extra single-stream pass)' if overlapped else '') +
                                ('; consecutive steps software-pipelined (pcdet/models/pipelined.py): the agent histogram is read on a side '
                                 'stream and the box counts of step i after step i+1 is queued, all reads inside the timed region '
                                 '(--no-pipeline: batch by batch)' + ('; steps alternate between %d replicas of the model on their own streams' % args.pipeline_replicas if args.pipeline_replicas > 1 else '') if was_pipelined else '') +
                                '; outputs equal to the plugin-default path (tests/test_gpu_e2e.py::test_pipeline_mode_*, test_overlapped_makers_*)'),
                       'peak_device_memory_mb': round(torch.cuda.max_memory_allocated(dev) / 2 ** 20, 1),
                       **({'elided': 'rsu BEV maker (overwritten by the car maker) and early BEV maker (training-only output): reference quirk F3'} if args.elide_dead_makers else {}),
                       'final_boxes_last_step': n_boxes, **({'loss_last_step': train_state['last_loss']} if args.train else {})},
            'roofline': roof,
            'roofline_hbm': vfe_roof,
            'decode_nms_us': decode_nms_us,
            'pipelined_replicas': (max(1, args.pipeline_replicas) if was_pipelined else 0),
            'kernel_ms_per_step': {f['kernel'].split(' ')[0]: round(f['ms_per_step'], 4) for f in fams[:14]},
            'kernel_ms_per_step_total': round(sum(f['ms_per_step'] for f in fams), 3),
        }
        if world > 1 and backend != 'nccl':
            line['data'] += ' -- FUNCTIONAL CHECK ONLY: backend %s, %d ranks on %d device(s)' % (backend, world, n_dev)
        if world == 1 and not args.train and not args.graph and args.shard == 'frame' and algo == 'auto' and args.optin and lately is None:
            # informational: the same workload with the OPT-IN arithmetic modes (never part of `value` / `dtype`): split bf16 (three MFMAs per
            # product, ~1e-5) and plain bf16 activations + products (the training loop's kernels, include/pcp_hip_mp.h; NOT inside the 1e-3
            # parity bar -- the error of every head map against this run's own fp32 maps is printed beside the rate)
            def plain_forward():
                work.copy_(pristine)
                bd_ = {'points': work, 'batch_size': batch, 'metadata': metas}
                with torch.no_grad():
                    pd_, _ = model(bd_)
                torch.cuda.synchronize()
                maps = {k: v.detach().float().clone() for k, v in model.dense_head.forward_ret_dict['pred_dicts'][0].items()}
                maps['spatial_features_2d'] = bd_['spatial_features_2d'].detach().float().clone()
                return pd_, maps

            def set_algo(a_):
                os.environ['PCP_CONV_ALGO'] = a_
                for m in model.modules():
                    if hasattr(m, 'invalidate_packed'):
                        m.invalidate_packed()
            ref_pred, ref_maps = plain_forward()
            for mode, note in (('bf16x3', '3x3 conv products as split bf16 (hi + lo, 16 mantissa bits; 3 bf16 MFMAs per product, f32 accumulate), f32 tensors'),
                               ('bf16', 'bf16 activation storage between the 3x3 layers + bf16 products on the training loop\'s kernels (k_mp_conv3x3_*), f32 accumulate, '
                                        'f32 1x1 / k2s2 layers, PFN, fusion, decode, NMS')):
                set_algo(mode)
                got_pred, got_maps = plain_forward()
                errs = {}
                for k, v in ref_maps.items():
                    scale = float(v.abs().max())
                    errs[k] = {'max_abs_err': round(float((got_maps[k] - v).abs().max()), 6), 'map_abs_max': round(scale, 4)}
                from helpers import match_boxes
                matched = total = 0
                for pa, pb in zip(ref_pred, got_pred):
                    n_, _w = match_boxes(pa['pred_boxes'].cpu().numpy(), pa['pred_scores'].cpu().numpy(), pb['pred_boxes'].cpu().numpy(),
                                         pb['pred_scores'].cpu().numpy(), tol=1e-3)
                    matched += n_
                    total += pa['pred_boxes'].shape[0]
                for _ in range(args.warmup):
                    step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                alt = time.perf_counter() - t1
                line['optin_' + mode] = {'value': round(batch * args.steps / alt, 3), 'unit': 'frames/s', 'ms_per_step': round(1e3 * alt / args.steps, 4),
                                         'head_map_errors_vs_this_runs_fp32_maps': errs,
                                         'final_boxes_matching_fp32_within_1e-3': '%d of %d' % (matched, total),
                                         'note': 'NOT the headline, batch by batch on one stream: ' + note + '; enable with PCP_CONV_ALGO=' + mode}
            set_algo('auto')
        if not args.no_cpu_baseline and world == 1:                    # rank 0 at N = 1 only (the contract); N > 1 lines carry null
            if args.config == 'lately6':
                line['cpu_baseline'] = cpu_baseline_lately(car_cfg, car_state, cfg, state, lately_frames[0], budget_s=args.cpu_baseline_budget)
            else:
                line['cpu_baseline'] = cpu_baseline(conf, cfg, state, pts_np, metas, budget_s=args.cpu_baseline_budget)
        else:
            line['cpu_baseline'] = None
        if (world == 1 and args.dist == 'uniform' and not args.no_secondary and not args.train and not args.graph and not args.latency
                and args.shard == 'frame' and algo == 'auto' and not args.plugin_default):
            # the LiDAR-like cloud ("ring") is reported alongside -- the same command on it, in a child process (this one keeps its
            # pipeline state); informational, never part of `value`
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), '--config', args.config, '--dist', 'ring', '--steps', str(min(args.steps, 20)),
                   '--warmup', str(min(args.warmup, 5)), '--no-cpu-baseline', '--no-secondary']
            try:
                res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, check=False)
                rl = [ln for ln in res.stdout.decode().splitlines() if ln.startswith('{')]
                r = json.loads(rl[-1]) if rl else None
            except Exception:                                              # the headline line must not depend on this extra
                r = None
            # the same command with every replica's forward replayed as ONE hipGraph per step (PipelinedDetector(graph=True)): what the host
            # costs when it does not enqueue ~210 launches per step itself -- the configuration for hosts with few cores per GPU
            try:
                res_g = subprocess.run([sys.executable, os.path.abspath(__file__), '--config', args.config, '--steps', str(min(args.steps, 20)), '--warmup',
                                        str(min(args.warmup, 5)), '--no-cpu-baseline', '--no-secondary', '--no-configs', '--pipeline-graph'],
                                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120, check=False)
                gl = [ln for ln in res_g.stdout.decode().splitlines() if ln.startswith('{')]
                rg = json.loads(gl[-1]) if gl else None
            except Exception:
                rg = None
            line['secondary_pipeline_graph'] = None if (rg is None or args.pipeline_graph) else {
                'value': rg['value'], 'unit': rg['unit'], 'ms_per_step': rg['ms_per_step'], 'steps': rg['steps'],
                'host_cpu_ms_per_step': rg.get('host_cpu_ms_per_step'), 'host_cpu_all_threads_ms_per_step': rg.get('host_cpu_all_threads_ms_per_step'),
                'note': 'the same command with --pipeline-graph (one hipGraph replay per step and replica; bitwise the eager detections); NOT the headline'}
            line['secondary_ring'] = None if r is None else {
                'value': r['value'], 'unit': r['unit'], 'ms_per_step': r['ms_per_step'], 'steps': r['steps'],
                'vfe_stage_ms_per_step': (r.get('roofline_hbm') or {}).get('ms_per_step'),
                'note': 'the same command with --dist ring (r = 70 u^2: 48 % of the points in multi-point pillars, ~850 in the cell under the '
                        'sensor); NOT the headline'}
        if (world == 1 and args.config == 'disco' and args.dist == 'uniform' and not args.no_configs and not args.train and not args.graph
                and not args.latency and args.shard == 'frame' and algo == 'auto' and not args.plugin_default and not args.elide_dead_makers
                and not args.host_input and not args.no_pipeline):
            # every BASELINE.json workload in the one line the driver records: this run's headline (config 5 inference) + bounded child runs
            # of configs 1 - 4 and of config 5's bf16 training loop
            line['configs'] = {'disco': summarize_line(line, 'config 5 inference: the headline of this line')}
            line['configs'].update(run_config_children(args.steps, args.warmup))
            car = line['configs'].get('car') or {}
            if car.get('cpu_baseline'):
                line['configs']['car_cpu_reference_path'] = dict(car['cpu_baseline'], note='config 1: the CPU path (oracle restatement of the reference modules, '
                                                                 'kind "port") of v2x_pointpillar_basic_car on 1 x 60k points, timed on this box\'s host cores')
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
