"""Agent-sharded execution of the fusion configs on one node: one process per GPU, `torch.distributed` backend "nccl"
= RCCL over xGMI (gloo in tests).  The reference has no such path -- it simulates every exchange through its dataloader (F5) -- so
the contract is *bit-identical results to single-process execution* of the same model on the union of the inputs.

  early fusion (config 4)  the pillar mean / max reductions run over the union of all agents' points, so the only exact exchange is
      the RAW POINTS: one ragged all-gather (<= 1.9 MB per agent); afterwards frames are dealt round-robin to the ranks.
  mid fusion  (config 5)   rank r encodes ITS agents with the frozen BEV maker and applies the shared compressor (it precedes the
      warp and is per map, v2x_fusion_disco.py:85, so this is exact); ONE all-gather of the compressed (B, H, W, 128) maps
      (8.4 MB per agent and frame); every rank then runs the ego branch (VFE on all points -- F4 --, backbone, warp + fuse, head)
      for its share of the frames.  The ego branch needs every agent's points as well, hence the same ragged point gather first.
Inference only (the training path shards by frame: tools/train.py).
"""
import torch

from pcp_amd import ops

from ..utils import v2x_exchange as ex
from .bev_layers.bev_maker import BEVMaker


def _select_frames(points, frames):
    """rows of the given frames, renumbered 0..len(frames)-1 in that order (row order inside a frame is preserved)"""
    col = points[:, 0]
    out = []
    for j, f in enumerate(frames):
        rows = points[col == float(f)].clone()
        rows[:, 0] = float(j)
        out.append(rows)
    return torch.cat(out, 0).contiguous() if out else points[:0]


class AgentShardedEarlyFusion:
    def __init__(self, model, group=None):
        self.model, self.group = model, group

    @torch.no_grad()
    def __call__(self, local_points, batch_size, metadata):
        """local_points: (n, 1+C) CUDA rows of THIS rank's agents (all frames, ego frame).  Returns (frames, pred_dicts of those frames)."""
        world, rank = ex._world(self.group)
        union, _counts = ex.all_gather_v_rows(local_points, self.group)
        frames = ex.shard_frames(batch_size, world, rank)
        if not frames:
            return frames, []
        bd = {'points': _select_frames(union, frames), 'batch_size': len(frames), 'metadata': [metadata[f] for f in frames]}
        pred_dicts, _ = self.model(bd)
        return frames, pred_dicts


class AgentShardedMidFusion:
    def __init__(self, model, group=None, run_early_maker=False):
        """run_early_maker: bev_img_early feeds only the training distillation loss (v2x_fusion_disco.py:119); off at inference."""
        self.model, self.group, self.run_early_maker = model, group, run_early_maker
        self.makers = [m for m in model.module_list if isinstance(m, BEVMaker)]
        self.ego_chain = [m for m in model.module_list if not isinstance(m, BEVMaker)]

    @staticmethod
    def agents_of_rank(agent_ids, world, rank):
        remote = sorted(a for a in agent_ids if a != 1)
        return [a for i, a in enumerate(remote) if i % world == rank]

    @torch.no_grad()
    def __call__(self, local_points, batch_size, metadata):
        world, rank = ex._world(self.group)
        model = self.model
        union, _counts = ex.all_gather_v_rows(local_points, self.group)
        agent_ids = sorted({int(a) for md in metadata for a in md['se3_from_ego'].keys()} | {1})
        mine = self.agents_of_rank(agent_ids, world, rank)
        # ---- stage A: encode + compress this rank's agents ----------------------------------------------------------------------
        bd = {'points': union, 'batch_size': batch_size, 'metadata': metadata}
        for mk in self.makers:
            if mk.maker_type == 'early' and not self.run_early_maker:
                continue
            mk.only_agents = set(mine)
            try:
                bd = mk(bd)
            finally:
                mk.only_agents = None
        fusion = model.v2x_mid_fusion
        H = W = None
        local = {}
        for a, img in bd.get('bev_img', {}).items():
            comp = fusion.compress_maps(img)                                      # (last, H, W, cc)
            pad = torch.zeros((batch_size,) + tuple(comp.shape[1:]), dtype=comp.dtype, device=comp.device)
            pad[:comp.shape[0]] = comp
            local[a] = (pad, comp.shape[0])
            H, W = comp.shape[1], comp.shape[2]
        # ---- one exchange: fixed-shape stack per rank (slots = max agents per rank), ids + valid-frame counts alongside ---------------
        slots = max(1, -(-len([a for a in agent_ids if a != 1]) // world))
        dev = union.device
        if H is None:                                                              # this rank owns no agent: shape from the config
            H = W = int(model.dataset.grid_size[0]) // int(model.dense_head.feature_map_stride)
        stack = torch.zeros((slots, batch_size, H, W, fusion.cc), dtype=torch.float32, device=dev)
        ids = torch.full((slots, 2), -1, dtype=torch.int64, device=dev)
        for i, a in enumerate(sorted(local)):
            stack[i] = local[a][0]
            ids[i, 0], ids[i, 1] = a, local[a][1]
        all_stacks = ex.all_gather_maps(stack, self.group)
        all_ids = ex.all_gather_maps(ids, self.group)
        comp_all = {}
        for st, idt in zip(all_stacks, all_ids):
            for i, (a, last) in enumerate(idt.tolist()):
                if a >= 0:
                    comp_all[int(a)] = st[i][:int(last)]
        # ---- stage B: the ego branch on this rank's frames ---------------------------------------------------------------------------
        frames = ex.shard_frames(batch_size, world, rank)
        if not frames:
            return frames, []
        fr = torch.tensor(frames, device=dev)
        bd2 = {'points': _select_frames(union, frames), 'batch_size': len(frames), 'metadata': [metadata[f] for f in frames]}
        order = sorted(comp_all)                                                   # ascending agent id = the order BEVMaker inserts
        bd2['bev_img'] = {a: None for a in order}
        pre = {}
        for a in order:
            full = torch.zeros((batch_size, H, W, fusion.cc), dtype=torch.float32, device=dev)
            full[:comp_all[a].shape[0]] = comp_all[a]
            pre[a] = full.index_select(0, fr).contiguous()
        bd2['bev_img_compressed'] = pre
        for m in self.ego_chain:
            bd2 = m(bd2)
        pred_dicts, _ = model.post_processing(bd2)
        return frames, pred_dicts
