"""train_one_epoch / train_model / checkpointing with the reference's signatures (tools/train_utils/train_utils.py:11-190).
The iteration body is the reference's (:39-58): lr_scheduler.step(it); model.train(); optimizer.zero_grad(); forward;
loss.backward(); clip; optimizer.step() -- with clipping folded into the fused optimizer step (optimizer.clip_grad_norm records
the threshold, the scaling happens inside pcp_adam_step from a device-resident norm, no host sync)."""
import glob
import os
import time

import torch


def train_one_epoch(model, optimizer, train_loader, model_func, lr_scheduler, accumulated_iter, optim_cfg, rank, tbar=None,
                    total_it_each_epoch=None, dataloader_iter=None, tb_log=None, leave_pbar=False, use_logger_to_record=False, logger=None,
                    logger_iter_interval=None, cur_epoch=None, total_epochs=None, ckpt_save_dir=None, ckpt_save_time_interval=None,
                    show_gpu_stat=False, log_every=10):
    """the keyword list of the reference's train_one_epoch (tools/train_utils/train_utils.py:11-14).  logger_iter_interval: iterations between
    log lines (log_every when None); ckpt_save_time_interval: seconds between `latest_model.pth` snapshots inside the epoch (:120-128; None =
    never); use_logger_to_record / show_gpu_stat are accepted for the reference's caller: progress always goes to `logger` (no tqdm bar),
    and gpustat is an NVIDIA tool."""
    if total_it_each_epoch is None:
        total_it_each_epoch = len(train_loader)
    if dataloader_iter is None:
        dataloader_iter = iter(train_loader)
    if logger_iter_interval:
        log_every = int(logger_iter_interval)
    t0 = time.time()
    ckpt_save_cnt = 1
    disp = {}
    for cur_it in range(total_it_each_epoch):
        try:
            batch = next(dataloader_iter)
        except StopIteration:
            dataloader_iter = iter(train_loader)
            batch = next(dataloader_iter)
        lr_scheduler.step(accumulated_iter)
        cur_lr = float(optimizer.lr)
        if tb_log is not None:
            tb_log.add_scalar('meta_data/learning_rate', cur_lr, accumulated_iter)
        model.train()
        optimizer.zero_grad()
        loss, tb_dict, disp_dict = model_func(model, batch)
        loss.backward()
        optimizer.clip_grad_norm(optim_cfg.GRAD_NORM_CLIP)
        optimizer.step()
        accumulated_iter += 1
        disp = {'loss': tb_dict.get('loss_total', tb_dict.get('loss_rpn', float('nan'))), 'lr': cur_lr}    # PointPillar reports loss_rpn only
        if tb_log is not None:
            tb_log.add_scalar('train/loss', disp['loss'], accumulated_iter)
            for key, val in tb_dict.items():
                tb_log.add_scalar('train/' + key, val, accumulated_iter)
        if rank == 0 and logger is not None and (cur_it % log_every == 0 or cur_it == total_it_each_epoch - 1):
            logger.info('%siter %d/%d  loss %.4f  lr %.3e  %.2f it/s' % ('' if cur_epoch is None else 'epoch %s/%s  ' % (cur_epoch, total_epochs),
                                                                         cur_it + 1, total_it_each_epoch, disp['loss'], cur_lr,
                                                                         (cur_it + 1) / max(time.time() - t0, 1e-9)))
        if (ckpt_save_time_interval and ckpt_save_dir is not None and _is_main_process(rank)
                and (time.time() - t0) // ckpt_save_time_interval >= ckpt_save_cnt):
            os.makedirs(str(ckpt_save_dir), exist_ok=True)
            save_checkpoint(checkpoint_state(model, optimizer, cur_epoch, accumulated_iter), filename=os.path.join(str(ckpt_save_dir), 'latest_model'))
            if logger is not None:
                logger.info('Save latest model to %s' % os.path.join(str(ckpt_save_dir), 'latest_model'))
            ckpt_save_cnt += 1
    return accumulated_iter


def model_state_to_cpu(model_state):
    return type(model_state)((k, v.cpu()) for k, v in model_state.items())


def checkpoint_state(model=None, optimizer=None, epoch=None, it=None):
    """{'epoch', 'it', 'model_state', 'optimizer_state', 'version'} (reference :165-183)"""
    optim_state = optimizer.state_dict() if optimizer is not None else None
    if optim_state is not None:
        optim_state = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in optim_state.items()}
    model_state = None
    if model is not None:
        m = model.module if hasattr(model, 'module') else model
        model_state = model_state_to_cpu(m.state_dict())
    return {'epoch': epoch, 'it': it, 'model_state': model_state, 'optimizer_state': optim_state, 'version': 'pcdet+pcp_amd'}


def save_checkpoint(state, filename='checkpoint'):
    """write to a temporary name, then rename: a reader (resume, repeat_eval) never sees a half-written file"""
    final = '%s.pth' % filename
    tmp = '%s.tmp.%d' % (final, os.getpid())
    torch.save(state, tmp)
    os.replace(tmp, final)


def _is_main_process(rank):
    """the caller's rank argument AND the process group's own rank must both say 0 (a caller that forgot to set cfg.LOCAL_RANK would
    otherwise make every rank prune and write the same files at once)"""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return rank == 0 and torch.distributed.get_rank() == 0
    return rank == 0


def train_model(model, optimizer, train_loader, model_func, lr_scheduler, optim_cfg, start_epoch, total_epochs, start_iter, rank,
                tb_log, ckpt_save_dir, train_sampler=None, lr_warmup_scheduler=None, ckpt_save_interval=1, max_ckpt_save_num=50,
                merge_all_iters_to_one_epoch=False, use_logger_to_record=False, logger=None, logger_iter_interval=None,
                ckpt_save_time_interval=None, show_gpu_stat=False):
    """exactly the keyword list the reference's tools/train.py:174-197 passes (train_utils.py:137-141), so that script drives this loop
    unchanged"""
    accumulated_iter = start_iter
    total_it_each_epoch = len(train_loader)
    dataloader_iter = None
    if merge_all_iters_to_one_epoch:                      # reference :145-148
        assert hasattr(train_loader.dataset, 'merge_all_iters_to_one_epoch')
        train_loader.dataset.merge_all_iters_to_one_epoch(merge=True, epochs=total_epochs)
        total_it_each_epoch = len(train_loader) // max(total_epochs, 1)
        dataloader_iter = iter(train_loader)
    for cur_epoch in range(start_epoch, total_epochs):
        if train_sampler is not None:
            train_sampler.set_epoch(cur_epoch)
        accumulated_iter = train_one_epoch(model, optimizer, train_loader, model_func, lr_scheduler=lr_scheduler,
                                           accumulated_iter=accumulated_iter, optim_cfg=optim_cfg, rank=rank, tb_log=tb_log,
                                           total_it_each_epoch=total_it_each_epoch, dataloader_iter=dataloader_iter, logger=logger,
                                           use_logger_to_record=use_logger_to_record, logger_iter_interval=logger_iter_interval,
                                           cur_epoch=cur_epoch, total_epochs=total_epochs, ckpt_save_dir=ckpt_save_dir,
                                           ckpt_save_time_interval=ckpt_save_time_interval, show_gpu_stat=show_gpu_stat)
        trained_epoch = cur_epoch + 1
        if trained_epoch % ckpt_save_interval == 0 and ckpt_save_dir is not None and _is_main_process(rank):
            os.makedirs(str(ckpt_save_dir), exist_ok=True)
            ckpts = sorted(glob.glob(os.path.join(str(ckpt_save_dir), 'checkpoint_epoch_*.pth')), key=os.path.getmtime)
            for old in ckpts[:max(0, len(ckpts) - max_ckpt_save_num + 1)]:
                try:
                    os.remove(old)
                except FileNotFoundError:
                    pass
            save_checkpoint(checkpoint_state(model, optimizer, trained_epoch, accumulated_iter),
                            filename=os.path.join(str(ckpt_save_dir), 'checkpoint_epoch_%d' % trained_epoch))
    return accumulated_iter
