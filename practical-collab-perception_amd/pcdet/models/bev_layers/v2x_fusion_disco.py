"""DiscoNet mid fusion on gfx950 (reference: pcdet/models/bev_layers/v2x_fusion_disco.py:8-126).

Same parameter tree (compressor.{0,1,3}, pixel_weightor.{conv1_1,bn1_1,conv1_2,bn1_2,conv1_4}, decompressor.{0,1,3}).
Per frame batch: compress ego + each agent map (MFMA convs), warp each compressed agent map into the ego frame
(pcp_warp_nearest, one launch per (agent, frame), theta built on the host exactly like :32-35), evaluate the
pixel weightor on [ego | agent] WITHOUT materialising the concat (dual-source 1x1 conv), softmax over agents + weighted
sum in one kernel, decompress.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from pcp_amd import fusion_host, lib, ops

from ..convnet import pack_conv_module
from ..packed import PackedModule, train_tape


class PixelWeightedFusionSoftmax(nn.Module):
    """parameter container (reference :8-26)"""

    def __init__(self, channel):
        super().__init__()
        self.conv1_1 = nn.Conv2d(channel * 2, 64, kernel_size=1, stride=1, padding=0)
        self.bn1_1 = nn.BatchNorm2d(64)
        self.conv1_2 = nn.Conv2d(64, 16, kernel_size=1, stride=1, padding=0)
        self.bn1_2 = nn.BatchNorm2d(16)
        self.conv1_4 = nn.Conv2d(16, 1, kernel_size=1, stride=1, padding=0)


def transform_bev_img(dst_se3_src, bev_in_src, pc_range_min, pix_size):
    """API twin of the reference helper (:29-45) for a single (C, H, W) CUDA map; returns the warped (C, H, W) map."""
    assert bev_in_src.dim() == 3
    C, H, W = bev_in_src.shape
    src = bev_in_src.permute(1, 2, 0).contiguous()
    cp = (C + 3) // 4 * 4
    if cp != C:
        pad = src.new_zeros((H, W, cp))
        pad[..., :C] = src
        src = pad
    dst = torch.empty_like(src)
    ops.warp_nearest(src, dst, fusion_host.warp_theta(dst_se3_src, H, W, pc_range_min, pix_size), cp)
    return dst[..., :C].permute(2, 0, 1)


class V2XMidFusionDisco(PackedModule):
    def __init__(self, model_cfg, in_channel=384):
        super().__init__()
        cc = model_cfg.COMPRESSED_CHANNELS
        self.compressor = nn.Sequential(
            nn.Conv2d(in_channel, cc, kernel_size=3, stride=1, padding=1, bias=False), nn.BatchNorm2d(cc), nn.ReLU(inplace=True),
            nn.Conv2d(cc, cc, kernel_size=3, stride=1, padding=1))
        self.pixel_weightor = PixelWeightedFusionSoftmax(cc)
        self.decompressor = nn.Sequential(
            nn.Conv2d(cc, in_channel, kernel_size=3, stride=1, padding=1, bias=False), nn.BatchNorm2d(in_channel),
            nn.ReLU(inplace=True), nn.Conv2d(in_channel, in_channel, kernel_size=3, stride=1, padding=1))
        self.pc_min = model_cfg.get('PC_RANGE_MIN', -51.2)
        self.pix_size = model_cfg.get('FINAL_BEV_PIXEL_SIZE', 0.2 * 4)
        self.model_cfg = model_cfg
        self.cc = cc
        self.loss_dict = {'loss_distill': 0.0}

    def _build_packed(self):
        pw = self.pixel_weightor
        return dict(c0=pack_conv_module(self.compressor[0], self.compressor[1], relu=True),
                    c1=pack_conv_module(self.compressor[3], None, relu=False),
                    w1=pack_conv_module(pw.conv1_1, pw.bn1_1, relu=True), w2=pack_conv_module(pw.conv1_2, pw.bn1_2, relu=True),
                    w3=pack_conv_module(pw.conv1_4, None, relu=True),
                    d0=pack_conv_module(self.decompressor[0], self.decompressor[1], relu=True),
                    d1=pack_conv_module(self.decompressor[3], None, relu=False),
                    wf=self._fold_weightor())

    # one launch for the whole weightor + softmax + weighted sum (pcp_disco_weight_fuse) instead of three pointwise launches per map and
    # k_softmax_fuse; False (PCP_DISCO_FUSED_WEIGHTOR=0): the launch-per-stage form, kept for A/B runs and the equality test
    fused_weightor = os.environ.get('PCP_DISCO_FUSED_WEIGHTOR', '1') != '0'

    def _fold_weightor(self):
        from ..convnet import _fold
        pw = self.pixel_weightor
        w1, b1 = _fold(pw.conv1_1, pw.bn1_1, 0)
        w2, b2 = _fold(pw.conv1_2, pw.bn1_2, 0)
        w3, b3 = _fold(pw.conv1_4, None, 0)
        f = lambda t: t.reshape(t.shape[0], -1).float().contiguous()
        return dict(w1=f(w1), b1=b1.float().contiguous(), w2=f(w2), b2=b2.float().contiguous(), w3=f(w3).reshape(-1).contiguous(),
                    b3=b3.float().reshape(1).contiguous())

    def _compress(self, pk, x_nhwc, out=None):
        h = pk['c0'].run(x_nhwc)
        if out is None:
            # a compressed map is what goes on the wire and into the fp32 warp: float32 whatever precision the convolutions ran in
            out = torch.empty(tuple(h.shape[:3]) + (pk['c1'].cout,), dtype=torch.float32, device=h.device)
        return pk['c1'].run(h, out=out)

    def compress_maps(self, bev_img_nchw):
        """the shared compressor on one agent's (B', 384, H, W) map -> (B', H, W, cc) NHWC (eval): what a remote GPU sends"""
        return self._compress(self.packed(), ops.as_nhwc(bev_img_nchw))

    def _weight(self, pk, ego, other, wbuf, col):
        # relu(conv1_4(relu(bn(conv1_2(relu(bn(conv1_1(cat[ego, other]))))))))  -> column `col` of wbuf
        h = ops.pointwise(ego, pk['w1'].w, pk['w1'].b, lib.PW_PLAIN, 2 * self.cc, 64, pk['w1'].cout_pad, relu=True, x2=other,
                          k_split=self.cc)
        h = pk['w2'].run(h)
        pk['w3'].run(h, out=wbuf, out_ch_off=col)

    def _forward_train(self, batch_dict):
        from ..train_path import FusionTrain
        from pcp_amd.train_layers import Act
        if getattr(self, '_pcp_train', None) is None:
            self._pcp_train = FusionTrain(self)
        self.invalidate_packed()
        out, loss = self._pcp_train.forward(Act(ops.as_nhwc(batch_dict['spatial_features_2d'])), batch_dict['bev_img'],
                                            batch_dict['metadata'], batch_dict.get('bev_img_early', None))
        if loss is not None:
            self.loss_dict['loss_distill'] = loss[0]
        batch_dict['spatial_features_2d'] = ops.nchw_view(out.t)
        train_tape(batch_dict).append(('v2x_mid_fusion', self._pcp_train.backward))
        return batch_dict

    @staticmethod
    def _warp_pairs(agent_order, metadata):
        """(agent, frame) pairs that are warped, in launch order: agents as batch_dict['bev_img'] lists them, frames ascending, pairs whose frame
        does not list the agent left out"""
        return [(a, b) for a in agent_order for b, meta in enumerate(metadata) if a in meta['se3_from_ego']]

    def theta_array(self, metadata, agent_order, h, w):
        """(pairs, 6) float32: the affines a static forward hands to the pose table as 'fusion.thetas', from the metadata alone"""
        pairs = self._warp_pairs(agent_order, metadata)
        th = fusion_host.warp_thetas([metadata[b]['se3_from_ego'][a] for a, b in pairs], h, w, self.pc_min, self.pix_size)
        return np.asarray(th, dtype=np.float32).reshape(len(pairs), 6)

    def forward(self, batch_dict):
        if self.training:
            return self._forward_train(batch_dict)
        pk = self.packed()
        ego_in = ops.as_nhwc(batch_dict['spatial_features_2d'])
        self._last_map_hw = (int(ego_in.shape[1]), int(ego_in.shape[2]))
        B, H, W, _ = ego_in.shape
        agents = list(batch_dict['bev_img'].items())
        n_maps = 1 + len(agents)
        dev = ego_in.device
        # ego + warped agents.  The warp writes every pixel of a present (agent, frame) pair (zeros outside the agent's map): only ABSENT pairs
        # are filled with zeros (none in a batch where every agent sees every frame -- no 200-MB fill per step)
        stack = torch.empty((n_maps, B, H, W, self.cc), dtype=torch.float32, device=dev)
        self._compress(pk, ego_in, out=stack[0])
        fuse_now = self.fused_weightor and self.cc == 128 and n_maps <= 16 and pk['wf']['w1'].shape == (64, 2 * self.cc) \
            and pk['wf']['w2'].shape == (16, 64)
        wbuf = None
        if not fuse_now:
            wbuf = torch.zeros((B, H, W, max(4, (n_maps + 3) // 4 * 4)), dtype=torch.float32, device=dev)
            self._weight(pk, stack[0], stack[0], wbuf, 0)
        pre = batch_dict.get('bev_img_compressed', None)      # agent-sharded execution: maps compressed on the agent's own GPU
        warps = []                                             # (source map, destination map, theta) of every (agent, frame) pair
        # the 2 x 3 affine of every pair, all at once on the host (the poses are metadata: nothing on the device is waited for)
        pairs = self._warp_pairs([agent_idx for agent_idx, _img in agents], batch_dict['metadata'])
        theta_list = fusion_host.warp_thetas([batch_dict['metadata'][b]['se3_from_ego'][a_] for a_, b in pairs], H, W, self.pc_min, self.pix_size)
        thetas = dict(zip(pairs, theta_list))
        for a, (agent_idx, bev_img) in enumerate(agents, start=1):
            comp = pre[agent_idx] if pre is not None else self._compress(pk, ops.as_nhwc(bev_img))
            for b_idx, meta in enumerate(batch_dict['metadata']):
                if agent_idx not in meta['se3_from_ego'] or b_idx >= comp.shape[0]:
                    stack[a, b_idx].zero_()
                    continue
                warps.append((comp[b_idx], stack[a, b_idx], thetas[(agent_idx, b_idx)]))
            if not fuse_now:
                ops.warp_nearest_batch(warps, self.cc)         # this agent's frames in one launch, then its weight logits
                warps = []
                self._weight(pk, stack[0], stack[a], wbuf, a)
        table = batch_dict.get('_pcp_pose_table', None)
        theta_dev = None
        if table is not None and warps:
            # graph mode: the affines live in device memory the runner refreshes before every replay (theta_array() computes the same rows in
            # the same order from the metadata alone); only the fused weightor's single launch is captured this way
            assert fuse_now and len(warps) == len(pairs), 'graph mode: every listed (agent, frame) pair is warped in the one fused launch'
            theta_dev = table.slot('fusion.thetas', np.asarray([th for _s, _d, th in warps], dtype=np.float32).reshape(len(warps), 6))
        ops.warp_nearest_batch(warps, self.cc, theta_dev=theta_dev)   # fused weightor: every pair of the forward in ONE launch
        live = batch_dict.get('_pcp_agent_live', None)
        if live is not None and agents:
            # hipGraph mode (BEVMaker static agent discovery): the maps of agents without rows / of frames behind an agent's last row do not
            # exist in the reference (bev_maker.py:156-190) -- zeroed from the device-side flags
            assert fuse_now, 'static agent discovery runs with the fused weightor'
            idx = [int(agent_idx) * B + b_idx if agent_idx in meta['se3_from_ego'] else -1
                   for agent_idx, _img in agents for b_idx, meta in enumerate(batch_dict['metadata'])]
            ops.zero_maps_unless(stack[1:].view(len(agents) * B, H, W, self.cc), idx, live)
        fused = torch.empty((B, H, W, self.cc), dtype=torch.float32, device=dev)
        if fuse_now:
            wf = pk['wf']
            # an agent without a single row is not in the reference's bev_img at all (:83): its map leaves the softmax (flag of its frame 0)
            live_index = None if live is None else [-1] + [int(agent_idx) * B for agent_idx, _img in agents]
            ops.disco_weight_fuse([stack[a] for a in range(n_maps)], wf['w1'], wf['b1'], wf['w2'], wf['b2'], wf['w3'], wf['b3'], self.cc, fused,
                                  live_index=live_index, live=live)
        else:
            ops.softmax_fuse([stack[a] for a in range(n_maps)], wbuf, self.cc, fused)
        # the published map stays float32 also under PCP_CONV_ALGO=bf16 (the sharded detector and the heads read it as float)
        out = pk['d1'].run(pk['d0'].run(fused), out=torch.empty((B, H, W, pk['d1'].cout), dtype=torch.float32, device=dev))
        batch_dict['spatial_features_2d'] = ops.nchw_view(out)
        return batch_dict
