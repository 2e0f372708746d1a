"""Dataset side of the plugin surface.  The reference's V2X-Sim datasets need the V2X-Sim data and nuscenes-devkit 1.1.9
(pcdet/datasets/v2x_sim/, README.md:77), neither available offline, and are numpy-only code outside the hot path
(SURVEY.md section 2).  What the hot path needs from a dataset is (a) the six attributes Detector3DTemplate reads and
(b) the collate layout: a frame-index column prepended to every point row (pcdet/datasets/dataset.py:224-229).
SyntheticV2XDataset provides both with the counter-based clouds of pcp_amd.synth; build_dataloader keeps the reference's
signature (pcdet/datasets/__init__.py:54-82) and DistributedSampler semantics."""
import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data import DistributedSampler as _DistributedSampler

from pcp_amd import synth

from ..models import DatasetInfo
from ..utils import common_utils

LAYOUT_OF = {'V2XSimDataset_CAR': 'car', 'V2XSimDataset_RSU': 'car', 'V2XSimDataset_EGO': 'lately', 'V2XSimDataset_EGO_LATE': 'lately',
             'V2XSimDataset_EGO_EARLY': 'early', 'V2XSimDataset_EGO_DISCO': 'disco'}
AGENTS_OF = {'car': 1, 'lately': 1, 'early': 6, 'disco': 6}


class SyntheticV2XDataset(DatasetInfo, Dataset):
    def __init__(self, dataset_cfg, class_names, training=False, root_path=None, logger=None):
        vs = [p.VOXEL_SIZE for p in dataset_cfg.DATA_PROCESSOR if 'VOXEL_SIZE' in p][0]
        enc = dataset_cfg.POINT_FEATURE_ENCODING
        DatasetInfo.__init__(self, class_names, dataset_cfg.POINT_CLOUD_RANGE, vs, len(enc.used_feature_list))
        self.dataset_cfg = dataset_cfg
        self.training = training
        self.logger = logger
        self.layout = LAYOUT_OF.get(dataset_cfg.DATASET, 'car')
        syn = dataset_cfg.get('SYNTHETIC', {})
        self.points_per_agent = int(syn.get('POINTS_PER_AGENT', 60000))
        self.num_frames = int(syn.get('NUM_FRAMES', 16))
        self.distribution = syn.get('DISTRIBUTION', 'uniform')

    @property
    def mode(self):
        return 'train' if self.training else 'test'

    def __len__(self):
        return self.num_frames

    def __getitem__(self, index):
        parts = []
        n_agents = AGENTS_OF[self.layout]
        for a in range(n_agents):
            c = synth.agent_cloud(agent=100 * index + a, n_points=self.points_per_agent, layout=self.layout, dist=self.distribution)
            if self.layout == 'disco':
                c[:, -1] = float(a)
            parts.append(c)
        meta = {'sample_token': 'synthetic_%06d' % index, 'lidar_id': 1,
                'se3_from_ego': {a: synth.agent_pose(a) for a in range(n_agents) if a != 1}}
        if self.dataset_cfg.DATASET == 'V2XSimDataset_EGO_LATE':
            # box-level fusion reads every agent's detections from the exchange database (v2x_sim_dataset_ego_late.py): synthetic stand-in
            # = 40 boxes per agent, neighbouring agents reporting overlapping boxes so that the NMS has something to merge
            meta['exchange_boxes'] = {a: self.synthetic_exchange_boxes(index, a) for a in range(6)}
        item = {'points': np.concatenate(parts, 0), 'frame_id': index, 'metadata': meta}
        if self.training:
            item['gt_boxes'] = self.synthetic_gt_boxes(index)
            if self.layout == 'car' and self.dataset_cfg.DATASET in ('V2XSimDataset_CAR', 'V2XSimDataset_RSU'):
                # configs 1 / 2 train HunterJr: foreground points carry (sweep, instance) and every instance its per-sweep motion
                fg, item['instances_tf'] = self.synthetic_foreground(index, item['gt_boxes'])
                item['points'] = np.concatenate([item['points'], fg], 0)
        return item

    def synthetic_foreground(self, index, gt, n_sweeps=11, per_local=12):
        """the fields V2XSimDataset_CAR adds for HunterJr (pcdet/datasets/v2x_sim/v2x_sim_dataset_car.py): see synth.instance_foreground"""
        return synth.instance_foreground(index, gt, n_sweeps, per_local)

    def synthetic_exchange_boxes(self, index, agent, n=40):
        """(n, 9) [box7, score, label]: the boxes of synthetic_gt_boxes jittered per agent (same objects seen by several agents)"""
        gt = self.synthetic_gt_boxes(index, n_max=n)
        s = synth.SEED_BASE + 7000 + 10 * index + agent
        out = np.zeros((gt.shape[0], 9), dtype=np.float32)
        out[:, :7] = gt[:, :7]
        out[:, 0:2] += synth.uniform(s, 1, gt.shape[0] * 2, -0.3, 0.3).reshape(-1, 2)
        out[:, 7] = synth.uniform(s, 2, gt.shape[0], 0.05, 0.95)
        out[:, 8] = gt[:, 7]
        return out[agent::2].copy()                       # every agent sees a different subset

    def synthetic_gt_boxes(self, index, n_max=40):
        """(n, 8) [x, y, z, dx, dy, dz, heading, class] car-sized boxes inside the range, n varies with the frame"""
        n = n_max - (index % 7)
        s = synth.SEED_BASE + 5000 + index
        r = self.point_cloud_range
        gt = np.zeros((n, 8), dtype=np.float32)
        gt[:, 0] = synth.uniform(s, 1, n, float(r[0]) + 1.0, float(r[3]) - 1.0)
        gt[:, 1] = synth.uniform(s, 2, n, float(r[1]) + 1.0, float(r[4]) - 1.0)
        gt[:, 2] = synth.uniform(s, 3, n, -3.0, -1.0)
        gt[:, 3] = synth.uniform(s, 4, n, 3.0, 5.5)
        gt[:, 4] = synth.uniform(s, 5, n, 1.5, 2.5)
        gt[:, 5] = synth.uniform(s, 6, n, 1.4, 2.0)
        gt[:, 6] = synth.uniform(s, 7, n, -3.14159, 3.14159)
        gt[:, 7] = np.floor(synth.uniform(s, 8, n, 1.0, len(self.class_names) + 0.999))
        return gt

    @staticmethod
    def collate_batch(batch_list, _unused=False):
        """reference layout: `points` gets the frame index as column 0; metadata / frame_id stay python lists."""
        ret = {'batch_size': len(batch_list)}
        ret['points'] = synth.collate([b['points'] for b in batch_list])
        ret['frame_id'] = np.array([b['frame_id'] for b in batch_list])
        ret['metadata'] = [b['metadata'] for b in batch_list]
        if 'gt_boxes' in batch_list[0]:
            # zero-padded to the longest frame (pcdet/datasets/dataset.py:260-266)
            m = max(b['gt_boxes'].shape[0] for b in batch_list)
            gt = np.zeros((len(batch_list), m, batch_list[0]['gt_boxes'].shape[1]), dtype=np.float32)
            for i, b in enumerate(batch_list):
                gt[i, :b['gt_boxes'].shape[0]] = b['gt_boxes']
            ret['gt_boxes'] = gt
            if 'instances_tf' in batch_list[0]:
                tf = np.zeros((len(batch_list), m) + batch_list[0]['instances_tf'].shape[1:], dtype=np.float32)
                tf[..., :3, :3] = np.eye(3, dtype=np.float32)
                for i, b in enumerate(batch_list):
                    tf[i, :b['instances_tf'].shape[0]] = b['instances_tf']
                ret['instances_tf'] = tf
        return ret

    def generate_prediction_dicts(self, batch_dict, pred_dicts, class_names, output_path=None):
        out = []
        for i, pd in enumerate(pred_dicts):
            labels = pd['pred_labels'].cpu().numpy()
            out.append({'frame_id': batch_dict['frame_id'][i], 'boxes_lidar': pd['pred_boxes'].cpu().numpy(),
                        'score': pd['pred_scores'].cpu().numpy(), 'pred_labels': labels,
                        'name': np.array(class_names)[labels - 1] if labels.size else np.zeros(0, dtype='<U8')})
        return out

    def evaluation(self, det_annos, class_names, **kwargs):
        n = sum(len(a['score']) for a in det_annos)
        return 'synthetic data: %d detections over %d frames (no ground truth, no mAP)\n' % (n, len(det_annos)), {'num_detections': n}


__all__ = {name: SyntheticV2XDataset for name in LAYOUT_OF}
__all__['SyntheticV2XDataset'] = SyntheticV2XDataset


class DistributedSampler(_DistributedSampler):
    """round-robin shard without shuffling in eval (reference: pcdet/datasets/__init__.py:31-51)"""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        super().__init__(dataset, num_replicas=num_replicas, rank=rank, shuffle=shuffle)


def build_dataloader(dataset_cfg, class_names, batch_size, dist, root_path=None, workers=0, seed=None, logger=None, training=True,
                     merge_all_iters_to_one_epoch=False, total_epochs=0):
    dataset = __all__.get(dataset_cfg.DATASET, SyntheticV2XDataset)(dataset_cfg=dataset_cfg, class_names=class_names,
                                                                   root_path=root_path, training=training, logger=logger)
    sampler = None
    if dist:
        rank, world = common_utils.get_dist_info()
        sampler = DistributedSampler(dataset, world, rank, shuffle=training)
    loader = DataLoader(dataset, batch_size=batch_size, pin_memory=True, num_workers=workers, shuffle=(sampler is None) and training,
                        collate_fn=dataset.collate_batch, drop_last=False, sampler=sampler, timeout=0)
    return dataset, loader, sampler
