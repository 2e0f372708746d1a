"""BaseBEVBackbone on gfx950 (reference: pcdet/models/backbones_2d/base_bev_backbone.py:6-112).

Same parameter tree (blocks.{i}.{1,2,4,5,...}, deblocks.{i}.{0,1}) so checkpoints load unchanged; the forward launches one
fused conv+BN+ReLU MFMA kernel per layer on NHWC buffers and writes the three up-sampled branches straight into their
channel windows of the 384-channel output (no torch.cat copy).
"""
import numpy as np
import torch
import torch.nn as nn

from pcp_amd import ops, pack

from ..convnet import pack_conv_module
from ..packed import PackedModule, train_tape


class BaseBEVBackbone(PackedModule):
    def __init__(self, model_cfg, input_channels):
        super().__init__()
        self.model_cfg = model_cfg
        if self.model_cfg.get('LAYER_NUMS', None) is not None:
            layer_nums, layer_strides = list(self.model_cfg.LAYER_NUMS), list(self.model_cfg.LAYER_STRIDES)
            num_filters = list(self.model_cfg.NUM_FILTERS)
            assert len(layer_nums) == len(layer_strides) == len(num_filters)
        else:
            layer_nums = layer_strides = num_filters = []
        if self.model_cfg.get('UPSAMPLE_STRIDES', None) is not None:
            upsample_strides = list(self.model_cfg.UPSAMPLE_STRIDES)
            num_upsample_filters = list(self.model_cfg.NUM_UPSAMPLE_FILTERS)
            assert len(upsample_strides) == len(num_upsample_filters)
        else:
            upsample_strides = num_upsample_filters = []
        self.layer_strides = layer_strides
        self.upsample_strides = upsample_strides
        c_in_list = [input_channels] + num_filters[:-1]
        self.blocks = nn.ModuleList()
        self.deblocks = nn.ModuleList()
        for idx in range(len(layer_nums)):
            # index 0 is the explicit ZeroPad2d(1) of the reference: it keeps the parameter indices (1, 2, 4, 5, ...) identical
            layers = [nn.ZeroPad2d(1),
                      nn.Conv2d(c_in_list[idx], num_filters[idx], kernel_size=3, stride=layer_strides[idx], padding=0, bias=False),
                      nn.BatchNorm2d(num_filters[idx], eps=1e-3, momentum=0.01), nn.ReLU()]
            for _ in range(layer_nums[idx]):
                layers += [nn.Conv2d(num_filters[idx], num_filters[idx], kernel_size=3, padding=1, bias=False),
                           nn.BatchNorm2d(num_filters[idx], eps=1e-3, momentum=0.01), nn.ReLU()]
            self.blocks.append(nn.Sequential(*layers))
            if len(upsample_strides) > 0:
                us = upsample_strides[idx]
                if us >= 1:
                    up = nn.ConvTranspose2d(num_filters[idx], num_upsample_filters[idx], int(us), stride=int(us), bias=False)
                else:
                    k = int(np.round(1 / us))
                    up = nn.Conv2d(num_filters[idx], num_upsample_filters[idx], k, stride=k, bias=False)
                self.deblocks.append(nn.Sequential(up, nn.BatchNorm2d(num_upsample_filters[idx], eps=1e-3, momentum=0.01), nn.ReLU()))
        c_in = sum(num_upsample_filters)
        if len(upsample_strides) > len(layer_nums):
            self.deblocks.append(nn.Sequential(
                nn.ConvTranspose2d(c_in, c_in, int(upsample_strides[-1]), stride=int(upsample_strides[-1]), bias=False),
                nn.BatchNorm2d(c_in, eps=1e-3, momentum=0.01), nn.ReLU()))
        self.num_bev_features = c_in
        self.num_upsample_filters = num_upsample_filters

    def _build_packed(self):
        blocks = []
        for seq in self.blocks:
            mods = list(seq)
            convs = []
            i = 1
            while i < len(mods):
                convs.append(pack_conv_module(mods[i], mods[i + 1], relu=True))
                i += 3
            blocks.append(convs)
        deblocks = [pack_conv_module(seq[0], seq[1], relu=True) for seq in self.deblocks]
        # the first layer once more in the layout of pcp_sparse_conv3x3_s2 (it can run from the pillar list instead of the dense canvas)
        sparse0 = None
        c0 = list(self.blocks[0])[1] if len(self.blocks) else None
        if c0 is not None and c0.kernel_size == (3, 3) and c0.stride == (2, 2) and c0.in_channels == 64 and c0.out_channels <= 64 \
                and c0.out_channels % 4 == 0:
            from ..convnet import _fold
            w, b = _fold(c0, list(self.blocks[0])[2], out_axis=0)
            sparse0 = pack.pack_conv3x3_sparse_s2(w, b) + (c0.out_channels,)
        return dict(blocks=blocks, deblocks=deblocks, sparse0=sparse0)

    def _forward_train(self, data_dict):
        from ..train_path import BackboneTrain
        from pcp_amd.train_layers import Act
        if getattr(self, '_pcp_train', None) is None:
            self._pcp_train = BackboneTrain(self)
        self.invalidate_packed()
        out = self._pcp_train.forward(Act(ops.as_nhwc(data_dict['spatial_features'])))
        data_dict['spatial_features_2d'] = ops.nchw_view(out.t)
        train_tape(data_dict).append(('backbone_2d', self._pcp_train.backward))
        return data_dict

    def forward(self, data_dict):
        if self.training:
            return self._forward_train(data_dict)
        pk = self.packed()
        sf = data_dict['spatial_features']
        first_done = False
        if sf is None:
            # pipeline mode: the VFE handed over the pillar list instead of a dense canvas (dynamic_pillar_vfe.py, sparse_first_layer)
            stash = data_dict['_pcp_vfe']
            if pk['sparse0'] is None:
                raise RuntimeError('the VFE skipped the dense canvas but this backbone has no sparse first layer '
                                   '(needs Conv2d(64, <= 64, 3, stride 2)); unset sparse_first_layer')
            wsp, bsp, c0 = pk['sparse0']
            from ..convnet import _plain_bf16
            nxt = pk['blocks'][0][1] if len(pk['blocks'][0]) > 1 else None
            to_bf16 = _plain_bf16() and nxt is not None and getattr(nxt, 'mp', None) is not None       # bf16 loop: the next layer reads bf16
            x = ops.sparse_conv3x3_s2(stash['pillar_rows'], stash['vox'], wsp, bsp, c0, relu=True,
                                      out_dtype=torch.bfloat16 if to_bf16 else torch.float32)
            in_h = stash['vox'].grid.ny
            first_done = True
        else:
            x = ops.as_nhwc(sf)
            in_h = x.shape[1]
        n_levels = len(pk['blocks'])
        ups, out = [], None
        ch_off = 0
        for i in range(n_levels):
            for j, conv in enumerate(pk['blocks'][i]):
                if i == 0 and j == 0 and first_done:
                    continue
                x = conv.run(x)
            stride = int(in_h / x.shape[1])
            data_dict['spatial_features_%dx' % stride] = ops.nchw_view(x)
            if len(pk['deblocks']) > 0:
                de = pk['deblocks'][i]
                if out is None:
                    us = self.upsample_strides[i]
                    oh = x.shape[1] * int(us) if us >= 1 else x.shape[1] // int(np.round(1 / us))
                    ow = x.shape[2] * int(us) if us >= 1 else x.shape[2] // int(np.round(1 / us))
                    # bf16 loop: a caller whose consumer reads bf16 (the DiscoNet compressor of a frozen teacher) asks for the
                    # concatenated map in bf16 -- the up-sampling layers then write it directly (pcp_mp_pointwise)
                    odt = torch.float32
                    if data_dict.get('_pcp_bf16_map', False) and x.dtype == torch.bfloat16 and all(getattr(d, 'mp', None) is not None
                                                                                                  for d in pk['deblocks'][:n_levels]):
                        odt = torch.bfloat16
                    out = torch.empty((x.shape[0], oh, ow, sum(self.num_upsample_filters[:n_levels])), dtype=odt, device=x.device)
                de.run(x, out=out, out_ch_off=ch_off)
                ch_off += de.cout
            else:
                ups.append(x)
        if out is None:
            out = ups[0] if len(ups) == 1 else torch.cat(ups, dim=3)
        if len(pk['deblocks']) > n_levels:
            out = pk['deblocks'][-1].run(out)
        data_dict['spatial_features_2d'] = ops.nchw_view(out)
        return data_dict
