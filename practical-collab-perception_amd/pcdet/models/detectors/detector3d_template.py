"""Detector3DTemplate: the plugin surface of the reference (pcdet/models/detectors/detector3d_template.py:18-476) --
`module_topology`, one `build_<name>(model_info_dict)` per slot, modules registered under the topology name (hence the
state-dict prefixes), recall bookkeeping and checkpoint loading by name + shape.  Only the slots of the PointPillars hot
path have registries behind them; a config that asks for another slot gets a clear error instead of an import failure.
"""
import os

import torch
import torch.nn as nn

from .. import backbones_2d, backbones_3d, dense_heads
from ..backbones_2d import map_to_bev
from ..backbones_3d import vfe
from ..bev_layers.bev_maker import BEVMaker
from ..bev_layers.hunter_jr import HunterJr
from ..bev_layers.v2x_fusion_disco import V2XMidFusionDisco
from ...ops.iou3d_nms import iou3d_nms_utils


class Detector3DTemplate(nn.Module):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__()
        from ..packed import _flush_counters
        self.register_state_dict_pre_hook(_flush_counters)      # checkpoints see the deferred BatchNorm counters (pcp_amd/train_layers.py)
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.dataset = dataset
        self.class_names = dataset.class_names
        self.register_buffer('global_step', torch.LongTensor(1).zero_())
        self.module_topology = [
            'bev_maker_rsu', 'bev_maker_car', 'bev_maker_early', 'vfe', 'backbone_3d', 'map_to_bev_module', 'pfe',
            'backbone_2d', 'corrector', 'v2x_mid_fusion', 'dense_head', 'point_head', 'roi_head'
        ]

    @property
    def mode(self):
        return 'TRAIN' if self.training else 'TEST'

    def update_global_step(self):
        self.global_step += 1

    def build_networks(self):
        info = {
            'module_list': [],
            'num_rawpoint_features': self.dataset.point_feature_encoder.num_point_features,
            'num_point_features': self.dataset.point_feature_encoder.num_point_features,
            'grid_size': self.dataset.grid_size,
            'point_cloud_range': self.dataset.point_cloud_range,
            'voxel_size': self.dataset.voxel_size,
            'depth_downsample_factor': self.dataset.depth_downsample_factor,
        }
        for name in self.module_topology:
            module, info = getattr(self, 'build_%s' % name)(model_info_dict=info)
            self.add_module(name, module)
        return info['module_list']

    # ---- slots on the hot path ---------------------------------------------------------------------------------------------
    def _build_bev_maker(self, key, info):
        if self.model_cfg.get(key, None) is None:
            return None, info
        maker = BEVMaker(self.model_cfg[key], self.num_class, self.dataset)
        info['module_list'].append(maker)
        return maker, info

    def build_bev_maker_rsu(self, model_info_dict):
        return self._build_bev_maker('BEV_MAKER_RSU', model_info_dict)

    def build_bev_maker_car(self, model_info_dict):
        return self._build_bev_maker('BEV_MAKER_CAR', model_info_dict)

    def build_bev_maker_early(self, model_info_dict):
        return self._build_bev_maker('BEV_MAKER_EARLY', model_info_dict)

    def build_vfe(self, model_info_dict):
        if self.model_cfg.get('VFE', None) is None:
            return None, model_info_dict
        module = vfe.__all__[self.model_cfg.VFE.NAME](
            model_cfg=self.model_cfg.VFE, num_point_features=model_info_dict['num_rawpoint_features'],
            point_cloud_range=model_info_dict['point_cloud_range'], voxel_size=model_info_dict['voxel_size'],
            grid_size=model_info_dict['grid_size'], depth_downsample_factor=model_info_dict['depth_downsample_factor'])
        model_info_dict['num_point_features'] = module.get_output_feature_dim()
        model_info_dict['module_list'].append(module)
        return module, model_info_dict

    def build_map_to_bev_module(self, model_info_dict):
        if self.model_cfg.get('MAP_TO_BEV', None) is None:
            return None, model_info_dict
        module = map_to_bev.__all__[self.model_cfg.MAP_TO_BEV.NAME](model_cfg=self.model_cfg.MAP_TO_BEV,
                                                                     grid_size=model_info_dict['grid_size'])
        model_info_dict['module_list'].append(module)
        model_info_dict['num_bev_features'] = module.num_bev_features
        return module, model_info_dict

    def build_backbone_2d(self, model_info_dict):
        if self.model_cfg.get('BACKBONE_2D', None) is None:
            return None, model_info_dict
        module = backbones_2d.__all__[self.model_cfg.BACKBONE_2D.NAME](
            model_cfg=self.model_cfg.BACKBONE_2D, input_channels=model_info_dict.get('num_bev_features', None))
        model_info_dict['module_list'].append(module)
        model_info_dict['num_bev_features'] = module.num_bev_features
        return module, model_info_dict

    def build_corrector(self, model_info_dict):
        if self.model_cfg.get('CORRECTOR', None) is None:
            return None, model_info_dict
        assert self.model_cfg.CORRECTOR.NAME == 'HunterJr', '%s is unknown' % self.model_cfg.CORRECTOR.NAME
        corrector = HunterJr(model_cfg=self.model_cfg.CORRECTOR, num_bev_features=model_info_dict['num_bev_features'],
                             voxel_size=model_info_dict['voxel_size'], point_cloud_range=model_info_dict['point_cloud_range'])
        model_info_dict['num_point_features'] = corrector.num_points_feat
        model_info_dict['module_list'].append(corrector)
        model_info_dict['num_bev_features'] = corrector.num_points_feat
        if getattr(self, 'vfe', None) is not None and hasattr(self.vfe, 'keep_bucket_order'):
            self.vfe.keep_bucket_order = True      # HunterJr's point head visits the points in the pillariser's bucket order
        return corrector, model_info_dict

    def build_v2x_mid_fusion(self, model_info_dict):
        if self.model_cfg.get('V2X_MID_FUSION', None) is None:
            return None, model_info_dict
        fusion = V2XMidFusionDisco(self.model_cfg.V2X_MID_FUSION)
        model_info_dict['module_list'].append(fusion)
        return fusion, model_info_dict

    def build_dense_head(self, model_info_dict):
        if self.model_cfg.get('DENSE_HEAD', None) is None or not self.model_cfg.DENSE_HEAD.get('ENABLE', True):
            return None, model_info_dict
        module = dense_heads.__all__[self.model_cfg.DENSE_HEAD.NAME](
            model_cfg=self.model_cfg.DENSE_HEAD, input_channels=model_info_dict['num_bev_features'],
            num_class=self.num_class if not self.model_cfg.DENSE_HEAD.CLASS_AGNOSTIC else 1, class_names=self.class_names,
            grid_size=model_info_dict['grid_size'], point_cloud_range=model_info_dict['point_cloud_range'],
            predict_boxes_when_training=self.model_cfg.get('ROI_HEAD', False), voxel_size=model_info_dict.get('voxel_size', False))
        model_info_dict['module_list'].append(module)
        return module, model_info_dict

    # ---- slots outside the hot path -----------------------------------------------------------------------------------------
    def _off_path(self, key, model_info_dict):
        if self.model_cfg.get(key, None) is not None:
            raise NotImplementedError('MODEL.%s is outside the PointPillars hot path this build covers (SURVEY.md section 2)' % key)
        return None, model_info_dict

    def build_backbone_3d(self, model_info_dict):
        if self.model_cfg.get('BACKBONE_3D', None) is not None and self.model_cfg.BACKBONE_3D.NAME in backbones_3d.__all__:
            raise NotImplementedError
        return self._off_path('BACKBONE_3D', model_info_dict)

    def build_pfe(self, model_info_dict):
        return self._off_path('PFE', model_info_dict)

    def build_point_head(self, model_info_dict):
        return self._off_path('POINT_HEAD', model_info_dict)

    def build_roi_head(self, model_info_dict):
        return self._off_path('ROI_HEAD', model_info_dict)

    def forward(self, **kwargs):
        raise NotImplementedError

    # ---- post-processing of anchor heads (reference :239-345), class-agnostic branch on the device ----------------------------
    def post_processing(self, batch_dict):
        """batch_cls_preds (B, N, C) logits + batch_box_preds (B, N, 7) -> per frame {pred_boxes, pred_scores, pred_labels}.
        sigmoid / max / score mask ran in pcp_anchor_decode; here: top-k + gather (pcp_topk_boxes), rotated NMS (pcp_nms_rotated),
        one host sync for the keep counts."""
        from pcp_amd import ops
        cfg = self.model_cfg.POST_PROCESSING
        nms = cfg.NMS_CONFIG
        if nms.MULTI_CLASSES_NMS:
            raise NotImplementedError('MULTI_CLASSES_NMS with a single anchor head trips the reference\'s own assertion '
                                      '(detector3d_template.py:283,295); only the class-agnostic branch is defined')
        if nms.NMS_TYPE != 'nms_gpu' or cfg.get('OUTPUT_RAW_SCORE', False):
            raise NotImplementedError('post-processing kernels cover NMS_TYPE nms_gpu without OUTPUT_RAW_SCORE')
        st = batch_dict['_pcp_anchor']
        boxes = batch_dict['batch_box_preds']
        B, N, _ = boxes.shape
        k = int(min(nms.NMS_PRE_MAXSIZE, N))
        cb, cs, cl, _ci, cnt = ops.topk_boxes(st['keys'], st['labels'], boxes, k)
        keep, kcnt = ops.nms_rotated(cb, None, nms.NMS_THRESH, k, nms.NMS_POST_MAXSIZE, n_dev=cnt)
        counts = kcnt.cpu().numpy()
        pred_dicts, recall_dict = [], {}
        for b in range(B):
            sel = keep[b, :int(counts[b])].long()
            final_boxes = cb[b, sel]
            recall_dict = self.generate_recall_record(box_preds=final_boxes, recall_dict=recall_dict, batch_index=b, data_dict=batch_dict,
                                                      thresh_list=cfg.RECALL_THRESH_LIST)
            pred_dicts.append({'pred_boxes': final_boxes, 'pred_scores': cs[b, sel], 'pred_labels': cl[b, sel].long() + 1})
        return pred_dicts, recall_dict

    # ---- recall bookkeeping (reference :347-389) --------------------------------------------------------------------------
    @staticmethod
    def generate_recall_record(box_preds, recall_dict, batch_index, data_dict=None, thresh_list=None):
        if 'gt_boxes' not in data_dict:
            return recall_dict
        rois = data_dict['rois'][batch_index] if 'rois' in data_dict else None
        gt_boxes = data_dict['gt_boxes'][batch_index]
        if len(recall_dict) == 0:
            recall_dict = {'gt': 0}
            for t in thresh_list:
                recall_dict['roi_%s' % str(t)] = 0
                recall_dict['rcnn_%s' % str(t)] = 0
        k = len(gt_boxes) - 1
        while k >= 0 and gt_boxes[k].sum() == 0:
            k -= 1
        cur_gt = gt_boxes[:k + 1]
        if cur_gt.shape[0] > 0:
            iou_rcnn = iou3d_nms_utils.boxes_iou3d_gpu(box_preds[:, 0:7], cur_gt[:, 0:7]) if box_preds.shape[0] > 0 \
                else torch.zeros((0, cur_gt.shape[0]))
            iou_roi = iou3d_nms_utils.boxes_iou3d_gpu(rois[:, 0:7], cur_gt[:, 0:7]) if rois is not None else None
            for t in thresh_list:
                if iou_rcnn.shape[0] > 0:
                    recall_dict['rcnn_%s' % str(t)] += (iou_rcnn.max(dim=0)[0] > t).sum().item()
                if iou_roi is not None:
                    recall_dict['roi_%s' % str(t)] += (iou_roi.max(dim=0)[0] > t).sum().item()
            recall_dict['gt'] += cur_gt.shape[0]
        return recall_dict

    # ---- checkpoints: {'model_state', 'epoch', 'it', 'optimizer_state', 'version'} (reference :391-476) ---------------------
    def _load_state_dict(self, model_state_disk, *, strict=True):
        state_dict = self.state_dict()
        update = {k: v for k, v in model_state_disk.items() if k in state_dict and state_dict[k].shape == v.shape}
        if strict:
            self.load_state_dict(update)
        else:
            state_dict.update(update)
            self.load_state_dict(state_dict)
        return state_dict, update

    def load_params_from_file(self, filename, logger, to_cpu=False, pre_trained_path=None):
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        logger.info('==> Loading parameters from checkpoint %s to %s' % (filename, 'CPU' if to_cpu else 'GPU'))
        loc = torch.device('cpu') if to_cpu else None
        checkpoint = torch.load(filename, map_location=loc, weights_only=False)
        disk = checkpoint['model_state']
        if pre_trained_path is not None:
            disk.update(torch.load(pre_trained_path, map_location=loc, weights_only=False)['model_state'])
        version = checkpoint.get('version', None)
        if version is not None:
            logger.info('==> Checkpoint trained from version: %s' % version)
        state_dict, update = self._load_state_dict(disk, strict=False)
        for key in state_dict:
            if key not in update:
                logger.info('Not updated weight %s: %s' % (key, str(state_dict[key].shape)))
        logger.info('==> Done (loaded %d/%d)' % (len(update), len(state_dict)))

    def load_params_with_optimizer(self, filename, to_cpu=False, optimizer=None, logger=None):
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        logger.info('==> Loading parameters from checkpoint %s to %s' % (filename, 'CPU' if to_cpu else 'GPU'))
        loc = torch.device('cpu') if to_cpu else None
        checkpoint = torch.load(filename, map_location=loc, weights_only=False)
        self._load_state_dict(checkpoint['model_state'], strict=True)
        if optimizer is not None:
            if checkpoint.get('optimizer_state', None) is not None:
                optimizer.load_state_dict(checkpoint['optimizer_state'])
            else:
                side = '%s_optim.%s' % (filename[:-4], filename[-3:])
                if os.path.exists(side):
                    optimizer.load_state_dict(torch.load(side, map_location=loc, weights_only=False)['optimizer_state'])
        if 'version' in checkpoint:
            print('==> Checkpoint trained from version: %s' % checkpoint['version'])
        logger.info('==> Done')
        return checkpoint.get('it', 0.0), checkpoint.get('epoch', -1)
