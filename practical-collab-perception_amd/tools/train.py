"""tools/train.py with the reference's command line (tools/train.py:24-60 of the reference):

    python train.py --cfg_file cfgs/v2x_sim_models/v2x_pointpillar_disco.yaml [--batch_size 4] [--epochs 20] [--ckpt X.pth]
                    [--launcher none|pytorch] [--set KEY VALUE ...]

Config 5 (DiscoNet) is the configuration with training kernels.  Multi-GPU:
`torchrun --nproc-per-node N train.py --launcher pytorch ...` -- one process per GPU, frames sharded by DistributedSampler, ONE
RCCL all-reduce of the flat fp32 gradient buffer per iteration inside optimizer.step() (no DistributedDataParallel wrapper: the
backward pass is hand written, there are no autograd hooks to bucket on).  Without V2X-Sim on disk the loader is the synthetic one.

Checkpoints: output/<cfg group>/<cfg name>/<extra_tag>/ckpt/checkpoint_epoch_N.pth, written by rank 0 only (temp file + rename).
`model_state` is interchangeable with the reference (same keys / shapes).  `optimizer_state` is this build's flat-buffer Adam state; a
reference checkpoint's torch-Adam state is mapped into it by parameter order when every shape matches, otherwise skipped with a warning;
the reference cannot read this build's optimizer state (its OptimWrapper expects torch.optim.Adam's dict).
"""
import argparse
import os
import pickle
import sys
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
sys.path.insert(0, str(HERE))

from pcdet.config import cfg, cfg_from_list, cfg_from_yaml_file, log_config_to_file  # noqa: E402
from pcdet.datasets import build_dataloader  # noqa: E402
from pcdet.models import build_network, model_fn_decorator  # noqa: E402
from pcdet.utils import common_utils  # noqa: E402
from train_utils.optimization import build_optimizer, build_scheduler  # noqa: E402
from train_utils.train_utils import train_model  # noqa: E402


def parse_config():
    p = argparse.ArgumentParser(description='arg parser')
    p.add_argument('--cfg_file', type=str, required=True)
    p.add_argument('--batch_size', type=int, default=None)
    p.add_argument('--epochs', type=int, default=None)
    p.add_argument('--workers', type=int, default=0)
    p.add_argument('--extra_tag', type=str, default='default')
    p.add_argument('--ckpt', type=str, default=None)
    p.add_argument('--pretrained_model', type=str, default=None)
    p.add_argument('--launcher', choices=['none', 'pytorch', 'slurm'], default='none')
    p.add_argument('--tcp_port', type=int, default=18888)
    p.add_argument('--local_rank', type=int, default=0)
    p.add_argument('--ckpt_save_interval', type=int, default=1)
    p.add_argument('--max_ckpt_save_num', type=int, default=30)
    p.add_argument('--output_dir', type=str, default=None)
    p.add_argument('--sync_bn', action='store_true', default=False, help='whether to use sync bn')
    p.add_argument('--set', dest='set_cfgs', default=None, nargs=argparse.REMAINDER)
    args = p.parse_args()
    cfg_from_yaml_file(args.cfg_file, cfg)
    cfg.TAG = Path(args.cfg_file).stem
    cfg.EXP_GROUP_PATH = '/'.join(args.cfg_file.split('/')[1:-1])      # reference tools/train.py:61: drop 'cfgs' and the file name
    if args.set_cfgs is not None:
        cfg_from_list(args.set_cfgs, cfg)
    return args, cfg


def init_distributed(args, cfg):
    """reference tools/train.py:69-78: `total_gpus, cfg.LOCAL_RANK = init_dist_<launcher>(tcp_port, local_rank, backend)` -- the rank every
    later rank-0-only action (logging, checkpoint save / prune) is gated on.  PCP_DIST_BACKEND=gloo is for the CPU tests."""
    if args.launcher == 'none':
        return False, 1
    init = getattr(common_utils, 'init_dist_%s' % args.launcher)
    local_rank = int(os.environ.get('LOCAL_RANK', args.local_rank))
    total_gpus, cfg.LOCAL_RANK = init(args.tcp_port, local_rank, backend=os.environ.get('PCP_DIST_BACKEND', 'nccl'))
    return True, total_gpus



def _optimizer_payload_ok(ck, path, n_flat):
    """the optimizer half of a checkpoint is loadable: absent (moments restart), the fused optimizer's own dict with flat buffers of this
    model's size, a reference torch-Adam dict (mapped tensor by tensor later; a mismatch there only resets the moments), or a readable
    `<name>_optim.pth` side file of one of those kinds.  -> (ok, reason)"""
    sd = ck.get('optimizer_state', None)
    if sd is None:
        side = '%s_optim.%s' % (path[:-4], path[-3:])
        if not os.path.exists(side):
            return True, ''
        try:
            sd = torch.load(side, map_location='cpu', weights_only=False)['optimizer_state']
        except (OSError, EOFError, RuntimeError, KeyError, TypeError, AttributeError, ValueError, pickle.UnpicklingError) as e:
            return False, 'optimizer side file unreadable (%s: %s)' % (type(e).__name__, e)
    if not isinstance(sd, dict):
        return False, 'optimizer_state is a %s' % type(sd).__name__
    if 'state' in sd and 'param_groups' in sd:
        return True, ''
    for k in ('t', 'exp_avg', 'exp_avg_sq', 'lr', 'mom'):
        if k not in sd:
            return False, 'optimizer_state lacks %r' % k
    for k in ('exp_avg', 'exp_avg_sq'):
        if not torch.is_tensor(sd[k]) or (n_flat is not None and sd[k].numel() != n_flat):
            return False, 'optimizer_state[%r] holds %s values, the model needs %s' % (k, sd[k].numel() if torch.is_tensor(sd[k]) else '?', n_flat)
    return True, ''


def choose_resume_checkpoint(model, ckpt_dir, logger, optimizer=None):
    """newest checkpoint under ckpt_dir that can be read, holds EXACTLY the tensors of `model` at their shapes (the loader is strict: an
    extra key would raise after validation) and whose optimizer payload fits `optimizer`, or None.  Nothing is loaded into the model
    here (reference tools/train.py:143-156 tries the newest file and lets a torn one raise).  -> path"""
    import glob
    want = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    n_flat = optimizer.flat_p.numel() if optimizer is not None and hasattr(optimizer, 'flat_p') else None
    for f in sorted(glob.glob(str(Path(ckpt_dir) / '*.pth')), key=os.path.getmtime, reverse=True):
        if f.endswith('_optim.pth'):
            continue
        try:
            ck = torch.load(f, map_location='cpu', weights_only=False)
            have = {k: tuple(v.shape) for k, v in ck['model_state'].items()}
        except (OSError, EOFError, RuntimeError, KeyError, TypeError, AttributeError, ValueError, pickle.UnpicklingError) as e:
            logger.info('could not resume from %s (%s: %s)' % (f, type(e).__name__, e))
            continue
        missing = [k for k in want if have.get(k) != want[k]]
        if missing:
            logger.info('could not resume from %s (%d tensors missing or of another shape, e.g. %s)' % (f, len(missing), missing[0]))
            continue
        extra = [k for k in have if k not in want]
        if extra:
            logger.info('could not resume from %s (%d tensors the model does not have, e.g. %s)' % (f, len(extra), extra[0]))
            continue
        ok, why = _optimizer_payload_ok(ck, f, n_flat) if optimizer is not None else (True, '')
        if not ok:
            logger.info('could not resume from %s (%s)' % (f, why))
            continue
        return f
    return None


def main():
    args, cfg = parse_config()
    dist_train, total_gpus = init_distributed(args, cfg)
    if args.batch_size is None:
        args.batch_size = cfg.OPTIMIZATION.BATCH_SIZE_PER_GPU
    else:
        assert args.batch_size % total_gpus == 0, 'Batch size should match the number of gpus'
        args.batch_size = args.batch_size // total_gpus
    args.epochs = cfg.OPTIMIZATION.NUM_EPOCHS if args.epochs is None else args.epochs
    # reference tools/train.py:93-96: output/<cfg group>/<cfg name>/<extra_tag>/ckpt is ALWAYS written (--output_dir overrides the root)
    out_dir = Path(args.output_dir) if args.output_dir else cfg.ROOT_DIR / 'output' / cfg.EXP_GROUP_PATH / cfg.TAG / args.extra_tag
    ckpt_dir = out_dir / 'ckpt'
    if cfg.LOCAL_RANK == 0:
        ckpt_dir.mkdir(parents=True, exist_ok=True)
    logger = common_utils.create_logger(None, rank=cfg.LOCAL_RANK)
    if dist_train:
        import torch.distributed as dist
        logger.info('total_batch_size: %d (%d ranks, backend %s)' % (total_gpus * args.batch_size, total_gpus, dist.get_backend()))
    log_config_to_file(cfg, logger=logger)
    train_set, train_loader, train_sampler = build_dataloader(cfg.DATA_CONFIG, cfg.CLASS_NAMES, args.batch_size, dist_train,
                                                              workers=args.workers, logger=logger, training=True, total_epochs=args.epochs)
    for key in ('BEV_MAKER_RSU', 'BEV_MAKER_CAR', 'BEV_MAKER_EARLY'):
        if cfg.MODEL.get(key, None) is not None and not os.path.isfile(str(cfg.MODEL[key].CKPT)):
            logger.info('%s.CKPT %s not found: synthetic teacher weights' % (key, cfg.MODEL[key].CKPT))
            cfg.MODEL[key].CKPT = None
    model = build_network(model_cfg=cfg.MODEL, num_class=len(cfg.CLASS_NAMES), dataset=train_set)
    if args.sync_bn and dist_train:
        # reference tools/train.py:128-129 (nn.SyncBatchNorm.convert_sync_batchnorm): every training-mode BatchNorm of the hand-written
        # path all-reduces its per-channel sums over the ranks (pcp_amd/train_ops.py; csrc/bn_train.hip pcp_bn_*_sums)
        from pcp_amd import train_ops
        train_ops.SYNC_BN = True
        logger.info('cross-rank BatchNorm statistics: on (%d ranks)' % total_gpus)
    start_epoch = it = 0
    if args.pretrained_model is not None:
        model.load_params_from_file(filename=args.pretrained_model, logger=logger, to_cpu=True)
    elif args.ckpt is None:
        from pcp_amd import synth
        st = synth.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
        model.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    model.cuda()
    optimizer = build_optimizer(model, cfg.OPTIMIZATION)
    if args.ckpt is not None:
        it, start_epoch = model.load_params_with_optimizer(args.ckpt, to_cpu=True, optimizer=optimizer, logger=logger)
    else:                                                   # resume from the newest loadable checkpoint (reference tools/train.py:143-156)
        # A checkpoint is VALIDATED (readable, every model key present with the right shape) before anything is loaded into the model or
        # the optimizer, so a torn file from a killed run cannot leave the model half-overwritten; rank 0 chooses and broadcasts its
        # choice, so all ranks resume from the same file even when a newer one appears while they start.
        choice = [choose_resume_checkpoint(model, ckpt_dir, logger, optimizer) if (cfg.LOCAL_RANK == 0 or not dist_train) else None]
        if dist_train:
            import torch.distributed as dist
            dist.broadcast_object_list(choice, src=0)
        if choice[0] is not None:
            it, start_epoch = model.load_params_with_optimizer(choice[0], to_cpu=True, optimizer=optimizer, logger=logger)
    it, start_epoch = int(it), max(int(start_epoch), 0)
    lr_scheduler, lr_warmup = build_scheduler(optimizer, total_iters_each_epoch=len(train_loader), total_epochs=args.epochs,
                                              last_epoch=start_epoch - 1, optim_cfg=cfg.OPTIMIZATION)
    logger.info('**********************Start training %s (%d trainable tensors, %.2f M parameters)**********************'
                % (cfg.TAG, len(optimizer.params), sum(p.numel() for p in optimizer.params) / 1e6))
    train_model(model, optimizer, train_loader, model_func=model_fn_decorator(), lr_scheduler=lr_scheduler, optim_cfg=cfg.OPTIMIZATION,
                start_epoch=start_epoch, total_epochs=args.epochs, start_iter=it, rank=cfg.LOCAL_RANK, tb_log=None, ckpt_save_dir=ckpt_dir,
                train_sampler=train_sampler, lr_warmup_scheduler=lr_warmup, ckpt_save_interval=args.ckpt_save_interval,
                max_ckpt_save_num=args.max_ckpt_save_num, logger=logger)
    logger.info('**********************End training**********************')


if __name__ == '__main__':
    main()
