"""eval_one_epoch with the reference's signature (tools/eval_utils/eval_utils.py:22-140): the hot loop is
`load_data_to_gpu(batch_dict); pred_dicts, ret_dict = model(batch_dict)`."""
import time

import torch

from pcdet.models import load_data_to_gpu
from pcdet.utils import common_utils


def eval_one_epoch(cfg, args, model, dataloader, epoch_id, logger, dist_test=False, result_dir=None):
    dataset = dataloader.dataset
    class_names = dataset.class_names
    det_annos = []
    infer_meter = common_utils.AverageMeter()
    model.eval()
    t_start = time.time()
    # --fast (MI355X pipeline mode): consecutive batches software-pipelined -- the detections of batch i are read back while batch i+1 is
    # queued (pcdet/models/pipelined.py; same detections, bit for bit).  Not with --infer_time (a per-batch wall time needs the sync).
    pipe, waiting, padded = None, None, None
    capacity = int(getattr(args, 'fast_capacity', 0) or 0)
    if getattr(args, 'fast', False) and not getattr(args, 'infer_time', False):
        from pcdet.models.pipelined import PaddedPoints, PipelinedDetector
        if PipelinedDetector.supports(model):
            # batches alternate between the model and a copy on their own streams.  --fast_capacity N (round 6): every batch is padded to N rows
            # (frame index -1 behind the real ones) and each replica's forward is replayed as ONE hipGraph -- the poses of a DiscoNet batch are
            # device-side data refreshed per batch, so one capture per (batch size, set of agents) serves the whole epoch
            pipe = PipelinedDetector(model, replicas=2, graph=capacity > 0)
            padded = PaddedPoints(capacity) if capacity > 0 else None
    for batch_dict in dataloader:
        load_data_to_gpu(batch_dict)
        if pipe is not None:
            pts = batch_dict['points']
            if padded is not None:
                if pts.shape[0] > padded.capacity:               # a larger batch than promised: a new capacity (and new captures) from here on
                    logger.info('--fast_capacity %d is below a batch of %d rows: padding to %d from here on'
                                % (padded.capacity, pts.shape[0], (pts.shape[0] + 4095) // 4096 * 4096))
                    padded = PaddedPoints((pts.shape[0] + 4095) // 4096 * 4096)
                pts = padded.fill(pts)
            prev = pipe.submit(pts, batch_dict['batch_size'], batch_dict.get('metadata', None), extra=batch_dict)
            if prev is not None:
                det_annos += dataset.generate_prediction_dicts(waiting, prev, class_names)
            waiting = batch_dict
            continue
        if getattr(args, 'infer_time', False):
            torch.cuda.synchronize()
            t0 = time.time()
        with torch.no_grad():
            pred_dicts, ret_dict = model(batch_dict)
        if getattr(args, 'infer_time', False):
            torch.cuda.synchronize()
            infer_meter.update((time.time() - t0) * 1000)
        det_annos += dataset.generate_prediction_dicts(batch_dict, pred_dicts, class_names)
    if pipe is not None and waiting is not None:
        det_annos += dataset.generate_prediction_dicts(waiting, pipe.flush(), class_names)
    if dist_test:
        det_annos = common_utils.merge_results_dist(det_annos, len(dataset))
    rank, _ = common_utils.get_dist_info()
    if rank != 0:
        return {}
    sec = time.time() - t_start
    logger.info('*************** Performance of EPOCH %s *****************' % epoch_id)
    logger.info('Generate label finished (sec_per_example: %.4f second).' % (sec / max(len(dataset), 1)))
    if getattr(args, 'infer_time', False):
        logger.info('Average infer time per batch: %.3f ms' % infer_meter.avg)
    result_str, result_dict = dataset.evaluation(det_annos, class_names)
    logger.info(result_str)
    result_dict['infer_time_ms'] = infer_meter.avg
    return result_dict
