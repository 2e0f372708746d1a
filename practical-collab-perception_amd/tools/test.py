"""tools/test.py with the reference's command line (tools/test.py:21-55 of the reference):

    python test.py --cfg_file cfgs/v2x_sim_models/v2x_pointpillar_basic_car.yaml --batch_size 4 [--ckpt X.pth]
                   [--launcher none|pytorch] [--infer_time] [--set KEY VALUE ...]

Without V2X-Sim on disk the dataloader is the synthetic one (same batch_dict layout); without --ckpt the weights are the
deterministic synthetic fill.  Multi-GPU: `torchrun --nproc-per-node N test.py --launcher pytorch ...` (frames sharded
round-robin, results merged in dataset order)."""
import argparse
import os
import sys
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE))

from eval_utils import eval_utils  # noqa: E402
from pcdet.config import cfg, cfg_from_list, cfg_from_yaml_file, log_config_to_file  # noqa: E402
from pcdet.datasets import build_dataloader  # noqa: E402
from pcdet.models import build_network  # noqa: E402
from pcdet.utils import common_utils  # noqa: E402


def parse_config():
    p = argparse.ArgumentParser(description='arg parser')
    p.add_argument('--cfg_file', type=str, required=True)
    p.add_argument('--batch_size', type=int, default=None)
    p.add_argument('--workers', type=int, default=0)
    p.add_argument('--extra_tag', type=str, default='default')
    p.add_argument('--ckpt', type=str, default=None)
    p.add_argument('--launcher', choices=['none', 'pytorch', 'slurm'], default='none')
    p.add_argument('--tcp_port', type=int, default=18888)
    p.add_argument('--local_rank', type=int, default=0)
    p.add_argument('--set', dest='set_cfgs', default=None, nargs=argparse.REMAINDER)
    p.add_argument('--infer_time', action='store_true', default=False)
    p.add_argument('--fast', action='store_true', help='MI355X pipeline mode: no per-pillar API tensors, persistent buffers, first backbone layer from the pillar list, '
                                                     'BEV-maker passes of a DiscoNet forward on their own HIP streams')
    p.add_argument('--fast_capacity', type=int, default=0, help='with --fast: pad every batch to this many rows (frame index -1 behind the real '
                   'ones) and replay each model replica\'s forward as one hipGraph per batch (host cost ~0.3 ms per batch instead of ~200 kernel '
                   'launches); the agents\' poses stay per-batch data.  0 = eager launches')
    args = p.parse_args()
    cfg_from_yaml_file(args.cfg_file, cfg)
    cfg.TAG = Path(args.cfg_file).stem
    if args.set_cfgs is not None:
        cfg_from_list(args.set_cfgs, cfg)
    return args, cfg


def main():
    args, cfg = parse_config()
    dist_test = args.launcher != 'none'
    if dist_test:                                           # the rank gates logging and the merged-result report (reference :148 drops it: Q3)
        _, cfg.LOCAL_RANK = getattr(common_utils, 'init_dist_%s' % args.launcher)(
            args.tcp_port, int(os.environ.get('LOCAL_RANK', args.local_rank)), backend=os.environ.get('PCP_DIST_BACKEND', 'nccl'))
    if args.batch_size is None:
        args.batch_size = cfg.OPTIMIZATION.BATCH_SIZE_PER_GPU
    logger = common_utils.create_logger(None, rank=cfg.LOCAL_RANK)
    log_config_to_file(cfg, logger=logger)
    test_set, test_loader, _ = build_dataloader(cfg.DATA_CONFIG, cfg.CLASS_NAMES, args.batch_size, dist_test, workers=args.workers,
                                                logger=logger, training=False)
    for key in ('BEV_MAKER_RSU', 'BEV_MAKER_CAR', 'BEV_MAKER_EARLY'):
        if cfg.MODEL.get(key, None) is not None and not os.path.isfile(str(cfg.MODEL[key].CKPT)):
            logger.info('%s.CKPT %s not found: synthetic weights' % (key, cfg.MODEL[key].CKPT))
            cfg.MODEL[key].CKPT = None
    model = build_network(model_cfg=cfg.MODEL, num_class=len(cfg.CLASS_NAMES), dataset=test_set)
    if args.ckpt is not None:
        model.load_params_from_file(filename=args.ckpt, logger=logger, to_cpu=True)
    else:
        from pcp_amd import synth
        st = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
        model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model.cuda()
    if args.fast:
        for m in model.modules():
            if hasattr(m, 'materialize_pillars'):
                m.materialize_pillars, m.reuse_buffers = False, True
                m.sparse_first_layer = True          # sparse clouds: no dense canvas, first backbone layer from the pillar list
        if hasattr(model, 'overlap_makers'):
            model.overlap_makers = True              # DiscoNet: frozen BEV-maker passes on their own HIP streams (bit-identical outputs)
    eval_utils.eval_one_epoch(cfg, args, model, test_loader, 'synthetic' if args.ckpt is None else Path(args.ckpt).stem, logger,
                              dist_test=dist_test)


if __name__ == '__main__':
    main()
