"""Train-mode execution of the trainable branch of config 5: ego VFE -> scatter -> BEV backbone ->
V2XMidFusionDisco -> CenterHead, forward AND backward, on the HIP kernels of include/pcp_hip_train.h.

No torch.autograd inside: each *Train object mirrors one reference module, keeps what its backward needs, and exposes
`backward(grad) -> grad of its input`.  The detector strings the backward closures together behind the loss tensor it
returns (a one-node autograd graph), so the reference's loop -- loss.backward(); clip_grad_norm_; optimizer.step()
(tools/train_utils/train_utils.py:49-58) -- drives it unchanged and finds the gradients in param.grad.

Reference modules: dynamic_pillar_vfe.py:35-46,94-147; pointpillar_scatter.py:14-37; base_bev_backbone.py:30-112;
v2x_fusion_disco.py:47-126; center_head.py:13-47,75-96,270-300,377-392.
"""
import os

import numpy as np
import torch

from pcp_amd import fusion_host, lib, ops, pack
from pcp_amd import train_layers as tl
from pcp_amd import train_ops as tops
from pcp_amd.train_layers import Act, ConvBNAct, ensure_grad


# PCP_PFN_BF16_ROWS=1 (bf16 loop only): the per-point 64-channel rows around the second PFN Linear as bf16, that Linear and its gradient
# GEMMs on the bf16 pointwise kernels.  Off by default: measured on config 5 it takes 0.27 ms of kernel time off the iteration but moves the
# third iteration of the 20-iteration loss curve 2.9 % away from the fp32 loop (2.5 % is the bound the test holds the bf16 loop to) -- the
# pillar max and the BatchNorm over 1.4 M points are where bf16 rows cost most; profiles/r04_pfn_bf16_rows.txt
PFN_BF16_ROWS = os.environ.get('PCP_PFN_BF16_ROWS', '0') == '1'


def _empty(shape, dev, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=dev)


# ---------------------------------------------------------------------------------------------------------------------
# VFE + scatter
# ---------------------------------------------------------------------------------------------------------------------

class VFETrain:
    def __init__(self, vfe):
        self.m = vfe
        self.vec0 = self.vec1 = None
        self.saved = None
        self._step = -1

    def _weights(self):
        if self._step == tl.StepClock.step:
            return self._w
        l0, l1 = self.m.pfn_layers[0], self.m.pfn_layers[1]
        w0 = l0.linear.weight.detach()                       # (32, F)
        dev = w0.device
        F = w0.shape[1]
        fw = 16 if F <= 16 else 32                              # feature rows are padded to 16 / 32 floats (pcp_pfn_train_features)
        w0p = torch.zeros((32, fw), dtype=torch.float32, device=dev)
        w0p[:, :F] = w0
        w1 = l1.linear.weight.detach()                       # (64, 64)
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
        self._w = dict(w0=pack.pack_plain(w0p, z(32)), w1=pack.pack_plain(w1.contiguous(), z(64)),
                       w1t=pack.pack_plain(w1.t().contiguous(), z(64)), F=F, fw=fw)
        if tl.mp_mode() and PFN_BF16_ROWS and dev.type == 'cuda':
            for k in ('w1', 'w1t'):
                self._w[k + '_mp'] = (self._w[k][0].to(torch.bfloat16), self._w[k][1], self._w[k][2])
        self._step = tl.StepClock.step
        return self._w

    def forward(self, batch_dict):
        m = self.m
        points = batch_dict['points']
        if points.dtype != torch.float32 or not points.is_contiguous():
            points = points.float().contiguous()
        batch_size = batch_dict['batch_size']
        dev = points.device
        grid = ops.make_grid(m.point_cloud_range, m.voxel_size, m.grid_size, batch_size)
        vox = ops.voxelize(points, grid, want_inverse=False, want_counts=False)
        ops.voxelize_sort_pillar_rows(vox)                                  # reproducible row order for the per-point GEMMs (atomic slots otherwise)
        P, Nk = (int(v) for v in vox.counters[:2].tolist())                # host sync: row counts of the per-point GEMMs
        n = max(points.shape[0], 1)
        w = self._weights()
        fbuf = _empty((n, w['fw']), dev)
        slot_pillar = _empty((n,), dev, torch.int32)
        tops.pfn_train_features(points, vox, m.num_raw_point_features, fbuf, slot_pillar)
        nx, ny = m.grid_size[0], m.grid_size[1]
        # bf16 loop: the canvas is born bf16 (the first backbone layer and its weight gradient read bf16): half the zero fill, no cast
        canvas = torch.zeros((batch_size, ny, nx, 64), dtype=torch.bfloat16 if tl.mp_mode() else torch.float32, device=dev)
        pf = _empty((max(P, 1), 64), dev)
        l0, l1 = m.pfn_layers[0], m.pfn_layers[1]
        if Nk > 0:
            fk = fbuf[:Nk]
            x0 = ops.pointwise(fk, w['w0'][0], w['w0'][1], lib.PW_PLAIN, w['fw'], 32, w['w0'][2], relu=False)
            self.vec0 = tops.bn_train_stats(x0, 32, l0.norm.weight.detach(), l0.norm.bias.detach(), l0.norm.eps, l0.norm.momentum,
                                            l0.norm.running_mean, l0.norm.running_var, vec=self.vec0)
            rdt = torch.bfloat16 if 'w1_mp' in w else torch.float32
            in1 = _empty((Nk, 64), dev, rdt)
            arg0 = _empty((max(P, 1), 32), dev, torch.int32)
            tops.pfn_train_mid(vox, x0, self.vec0, in1, arg0)
            if 'w1_mp' in w:
                x1 = tops.mp_pointwise(in1, w['w1_mp'][0], w['w1_mp'][1], lib.PW_PLAIN, 64, 64, w['w1_mp'][2], relu=False)
            else:
                x1 = ops.pointwise(in1, w['w1'][0], w['w1'][1], lib.PW_PLAIN, 64, 64, w['w1'][2], relu=False)
            self.vec1 = tops.bn_train_stats(x1, 64, l1.norm.weight.detach(), l1.norm.bias.detach(), l1.norm.eps, l1.norm.momentum,
                                            l1.norm.running_mean, l1.norm.running_var, vec=self.vec1)
            arg1 = _empty((max(P, 1), 64), dev, torch.int32)
            tops.pfn_train_out(vox, x1, self.vec1, pf, arg1, canvas)
            for nrm in (l0.norm, l1.norm):
                tl.bump_batches_tracked(nrm)
            self.saved = dict(vox=vox, Nk=Nk, P=P, fk=fk, x0=x0, in1=in1, arg0=arg0, x1=x1, arg1=arg1)
        else:
            self.saved = None
        batch_dict['voxel_features'] = batch_dict['pillar_features'] = pf[:P]
        batch_dict['voxel_coords'] = vox.voxel_coords[:P]
        batch_dict['_pcp_vfe'] = dict(canvas=canvas, vox=vox)
        return batch_dict

    def backward(self, dcanvas):
        """dcanvas: (B, ny, nx, 64) NHWC gradient of the BEV canvas."""
        if self.saved is None:
            return None
        s = self.saved
        l0, l1 = self.m.pfn_layers[0], self.m.pfn_layers[1]
        if dcanvas.dtype == torch.bfloat16 and not dcanvas.is_contiguous():
            dcanvas = dcanvas.contiguous()                # the bf16 loop hands the canvas gradient over as bf16: read as it is
        dev = dcanvas.device
        w = self._weights()
        Nk = s['Nk']
        dz1 = _empty((Nk, 64), dev, s['x1'].dtype)
        tops.pfn_train_route_out_grad(s['vox'], Nk, s['arg1'], dz1, dcanvas=dcanvas)
        tops.bn_act_backward(dz1, s['x1'], 64, self.vec1, True, ensure_grad(l1.norm.weight), ensure_grad(l1.norm.bias))
        tops.pointwise_wgrad(tops.rowmap(dz1, 64), tops.rowmap(s['in1'], 64), Nk, ensure_grad(l1.linear.weight))
        if dz1.dtype == torch.bfloat16:
            din1 = tops.mp_pointwise(dz1, w['w1t_mp'][0], w['w1t_mp'][1], lib.PW_PLAIN, 64, 64, w['w1t_mp'][2], relu=False)
        else:
            din1 = ops.pointwise(dz1, w['w1t'][0], w['w1t'][1], lib.PW_PLAIN, 64, 64, w['w1t'][2], relu=False)
        da0 = _empty((Nk, 32), dev)
        tops.pfn_train_route_mid_grad(s['vox'], din1, s['arg0'], da0)
        tops.bn_act_backward(da0, s['x0'], 32, self.vec0, True, ensure_grad(l0.norm.weight), ensure_grad(l0.norm.bias))
        g0 = torch.empty((32, w['fw']), dtype=torch.float32, device=dev)
        tops.pointwise_wgrad(tops.rowmap(da0, 32), tops.rowmap(s['fk'], w['fw']), Nk, g0)
        ensure_grad(l0.linear.weight).copy_(g0[:, :w['F']])
        return None


# ---------------------------------------------------------------------------------------------------------------------
# BEV backbone
# ---------------------------------------------------------------------------------------------------------------------

class BackboneTrain:
    def __init__(self, bb):
        self.m = bb
        self.blocks = []
        for bi, seq in enumerate(bb.blocks):
            mods = list(seq)
            layers, i = [], 1
            while i < len(mods):
                layers.append(ConvBNAct(mods[i], mods[i + 1], relu=True, name='blocks.%d.%d' % (bi, i)))
                i += 3
            self.blocks.append(layers)
        self.deblocks = [ConvBNAct(seq[0], seq[1], relu=True, name='deblocks.%d' % i) for i, seq in enumerate(bb.deblocks)]
        if len(self.deblocks) != len(self.blocks):
            raise NotImplementedError('training path covers one deblock per block (all five configs)')
        self.offs = None

    def forward(self, x):
        """x: Act of the canvas.  Returns Act of the concatenated up-sampled map."""
        offs, out = [], None
        ch = 0
        for layers, de in zip(self.blocks, self.deblocks):
            for layer in layers:
                x = layer.forward(x)
            if out is None:
                shp = de.out_shape(x)
                # bf16 loop: the up-sampling layers write the concatenated map as bf16 (its consumers are bf16 3x3 layers)
                odt = torch.bfloat16 if (x.t.dtype == torch.bfloat16 and all(d.mp_pointwise_capable() for d in self.deblocks)) else torch.float32
                out = _empty(shp + (sum(d.cout for d in self.deblocks),), x.t.device, odt)
            de.forward(x, out=Act(out, ch, de.cout))
            offs.append(ch)
            ch += de.cout
        self.offs = offs
        return Act(out, 0, ch)

    def backward(self, dout):
        """dout: Act, gradient of the concatenated map.  Returns Act gradient of the canvas."""
        g_next = None
        for bi in range(len(self.blocks) - 1, -1, -1):
            de = self.deblocks[bi]
            g = de.backward(Act(dout.t, dout.off + self.offs[bi], de.cout))
            if g_next is not None:
                tops.accumulate(g.t, g_next.t, g.c, dst_ch_off=g.off, src_ch_off=g_next.off)
            for layer in reversed(self.blocks[bi]):
                g = layer.backward(g)
            g_next = g
        return g_next


# ---------------------------------------------------------------------------------------------------------------------
# DiscoNet mid fusion
# ---------------------------------------------------------------------------------------------------------------------

class FusionTrain:
    def __init__(self, fu):
        self.m = fu
        c, d, pw = fu.compressor, fu.decompressor, fu.pixel_weightor
        self.mk_comp = lambda tag: (ConvBNAct(c[0], c[1], relu=True, name='compressor.0/' + tag),
                                    ConvBNAct(c[3], None, relu=False, name='compressor.3/' + tag))
        self.mk_weight = lambda tag: (ConvBNAct(pw.conv1_1, pw.bn1_1, relu=True, name='conv1_1/' + tag),
                                      ConvBNAct(pw.conv1_2, pw.bn1_2, relu=True, name='conv1_2/' + tag))
        self.comp_ego = self.mk_comp('ego')
        self.d0 = ConvBNAct(d[0], d[1], relu=True, name='decompressor.0')
        self.d1 = ConvBNAct(d[3], None, relu=False, name='decompressor.3')
        self.comp_agents, self.weights = {}, {}
        self.saved = None

    def forward(self, ego_in, bev_img, metadata, bev_early):
        fu = self.m
        cc = fu.cc
        B, H, W, _ = ego_in.t.shape
        dev = ego_in.t.device
        agents = list(bev_img.items())
        n_maps = 1 + len(agents)
        if n_maps > 8:
            raise NotImplementedError('fusion training kernels are built for <= 8 maps (config 5 has 6)')
        # [ego | warped agent] per map.  Every pixel of a present (agent, frame) pair is written by the warp (zeros outside the agent's map) and
        # the ego halves by copies: only the agent half of an ABSENT pair needs a fill (none in a batch where every agent sees every frame)
        # bf16 loop: the stacked maps are bf16 (written by the bf16 compressor, read by the weightor's first 1x1 layer on pcp_mp_pointwise and by
        # the storage-typed fusion kernels; softmax / weighted sum / gradients stay float32)
        mdt = torch.bfloat16 if (tl.mp_mode() and cc % 8 == 0) else torch.float32
        esz = 2 if mdt == torch.bfloat16 else 4
        cats = [torch.empty((B, H, W, 2 * cc), dtype=mdt, device=dev) for _ in range(n_maps)]
        c0, c1 = self.comp_ego
        c1.forward(c0.forward(ego_in), out=Act(cats[0], 0, cc))
        ego = cats[0][..., :cc]
        cats[0][..., cc:].copy_(ego)
        pairs = [(agent_idx, b_idx) for agent_idx, _img in agents for b_idx, meta in enumerate(metadata) if agent_idx in meta['se3_from_ego']]
        thetas = dict(zip(pairs, fusion_host.warp_thetas([metadata[b]['se3_from_ego'][a_] for a_, b in pairs], H, W, fu.pc_min, fu.pix_size)))
        for a, (agent_idx, img) in enumerate(agents, start=1):
            if agent_idx not in self.comp_agents:
                self.comp_agents[agent_idx] = self.mk_comp('agent%d' % agent_idx)
            a0, a1 = self.comp_agents[agent_idx]
            comp = a1.forward(a0.forward(Act(ops.as_nhwc(img))), out_dtype=mdt)   # BatchNorm sees this agent's batch (train mode), no gradient
            cats[a][..., :cc].copy_(ego)
            for b_idx, meta in enumerate(metadata):
                if agent_idx not in meta['se3_from_ego'] or b_idx >= comp.t.shape[0]:
                    cats[a][b_idx, :, :, cc:].zero_()
                    continue
                ops.warp_nearest(comp.t[b_idx], cats[a][b_idx], thetas[(agent_idx, b_idx)], cc, dst_ch_off=cc)
        h2 = []
        for a in range(n_maps):
            if a not in self.weights:
                self.weights[a] = self.mk_weight('map%d' % a)
            w1, w2 = self.weights[a]
            h2.append(w2.forward(w1.forward(Act(cats[a], 0, 2 * cc))).t)
        pw = fu.pixel_weightor
        logits = torch.zeros((B, H, W, 8), dtype=torch.float32, device=dev)
        tops.disco_weight_logits(h2, pw.conv1_4.weight.detach().reshape(-1).contiguous(), pw.conv1_4.bias.detach(), logits)
        map_ptrs = [cats[0].data_ptr()] + [cats[a].data_ptr() + esz * cc for a in range(1, n_maps)]
        fused = _empty((B, H, W, cc), dev)
        ops.softmax_fuse_raw(map_ptrs, logits, cc, 2 * cc, fused, map_dtype=mdt)
        out = self.d1.forward(self.d0.forward(Act(fused)), out_dtype=torch.float32)      # fp32: the distillation loss and the head read it
        self.saved = dict(cats=cats, h2=h2, logits=logits, map_ptrs=map_ptrs, n_maps=n_maps, mdt=mdt)
        loss = None
        self.dgrad_distill = None
        if bev_early is not None:
            early = ops.as_nhwc(bev_early)
            self.dgrad_distill = _empty(tuple(out.t.shape), dev)
            loss = tops.distill_loss(out.t, early, out.c, weight=10.0, dfused=self.dgrad_distill)
        return out, loss

    def backward(self, dout):
        s = self.saved
        fu = self.m
        cc = fu.cc
        dev = dout.t.device
        if self.dgrad_distill is not None:
            tops.accumulate(dout.t, self.dgrad_distill, dout.c, dst_ch_off=dout.off)
        g = self.d0.backward(self.d1.backward(dout), dx_dtype=torch.float32)   # dL/d fused (B, H, W, cc), fp32 for pcp_disco_fuse_backward
        n_maps = s['n_maps']
        B, H, W = g.t.shape[0], g.t.shape[1], g.t.shape[2]
        d_ego = _empty((B, H, W, cc), dev)
        dh2 = [_empty((B, H, W, 16), dev) for _ in range(n_maps)]
        pw = fu.pixel_weightor
        gw4 = ensure_grad(pw.conv1_4.weight)
        gb4 = ensure_grad(pw.conv1_4.bias)
        tops.disco_fuse_backward(s['map_ptrs'], 2 * cc, cc, s['logits'], g.t, s['h2'], pw.conv1_4.weight.detach().reshape(-1).contiguous(),
                                 d_ego, dh2, gw4.view(-1), gb4, map_dtype=s['mdt'])
        for a in range(n_maps):
            w1, w2 = self.weights[a]
            dcat = w1.backward(w2.backward(Act(dh2[a]), accumulate=a > 0), accumulate=a > 0)
            tops.accumulate(d_ego, dcat.t, cc)
            if a == 0:
                tops.accumulate(d_ego, dcat.t, cc, src_ch_off=cc)
        c0, c1 = self.comp_ego
        return c0.backward(c1.backward(Act(d_ego)))


# ---------------------------------------------------------------------------------------------------------------------
# CenterHead
# ---------------------------------------------------------------------------------------------------------------------

class HeadTrain:
    def __init__(self, head):
        self.m = head
        if len(head.heads_list) != 1:
            raise NotImplementedError('training kernels cover one detection head (all five configs)')
        sh = head.shared_conv
        self.shared = ConvBNAct(sh[0], sh[1], relu=True, name='shared_conv')
        self.names = head.head_names[0]
        h = head.heads_list[0]
        self.seqs = [getattr(h, n) for n in self.names]
        if not all(len(s) == 2 for s in self.seqs):
            raise NotImplementedError('training kernels cover NUM_HM_CONV = num_conv = 2 (all five configs)')
        self.stage1 = [ConvBNAct(s[0][0], s[0][1], relu=True, name='%s.0' % n) for s, n in zip(self.seqs, self.names)]
        self.outs = [s[1].weight.shape[0] for s in self.seqs]
        self.offs = [int(v) for v in np.concatenate([[0], np.cumsum(self.outs)])]
        self.c = sh[0].weight.shape[0]
        self.ld = max(16, (self.offs[-1] + 3) // 4 * 4)
        self._step = -1
        self.saved = None

    def _stage2_forms(self):
        if self._step == tl.StepClock.step:
            return self._forms
        c, n = self.c, len(self.seqs)
        dev = self.seqs[0][1].weight.device
        wg = torch.cat([s[1].weight.detach().float() for s in self.seqs], 0)                       # (n_out, c, 3, 3)
        bg = torch.cat([s[1].bias.detach().float() for s in self.seqs], 0)
        grouped = (wg.permute(0, 2, 3, 1).reshape(wg.shape[0], 9, c).contiguous(), bg.contiguous())
        # data gradient: conv3x3 from the ld-channel gradient buffer to the n*c stage-1 channels, block-diagonal flipped weights
        wt = torch.zeros((n * c, self.ld, 3, 3), dtype=torch.float32, device=dev)
        for i, s in enumerate(self.seqs):
            wt[c * i:c * (i + 1), self.offs[i]:self.offs[i + 1]] = s[1].weight.detach().flip(2, 3).transpose(0, 1)
        bw = pack.pack_conv3x3(wt, torch.zeros(n * c, dtype=torch.float32, device=dev))
        # data gradient of the n stage-1 convs in one launch: weights concatenated along the contraction axis
        w1t = torch.cat([s[0][0].weight.detach().flip(2, 3).transpose(0, 1) for s in self.seqs], 1).contiguous()   # (c, n*c, 3, 3)
        z = torch.zeros(c, dtype=torch.float32, device=dev)
        bw1 = dict(direct=pack.pack_conv3x3(w1t, z), wino=pack.pack_conv3x3_winograd(w1t, z))
        self._forms = dict(grouped=grouped, bw2=bw, bw1=bw1)
        self._step = tl.StepClock.step
        return self._forms

    def forward(self, x):
        """x: Act (B, H, W, 384).  Returns the raw head-map buffer (B, H, W, ld)."""
        s = self.shared.forward(x)
        B, H, W, _ = s.t.shape
        dev = s.t.device
        n, c = len(self.seqs), self.c
        mid = _empty((B, H, W, n * c), dev)
        for i, layer in enumerate(self.stage1):
            layer.forward(s, out=Act(mid, c * i, c))
        f = self._stage2_forms()
        buf = torch.zeros((B, H, W, self.ld), dtype=torch.float32, device=dev)
        ops.conv3x3_grouped_small(mid, f['grouped'][0], f['grouped'][1], self.offs, buf)
        self.saved = dict(mid=mid, s=s)
        return buf

    def backward(self, dhead):
        """dhead: (B, H, W, ld) gradient of the raw maps.  Returns Act gradient of the head input (B, H, W, 384)."""
        sv = self.saved
        n, c = len(self.seqs), self.c
        dev = dhead.device
        f = self._stage2_forms()
        mid = sv['mid']
        # final convs: weight gradient of the block-diagonal (ld, n*c) conv, then the diagonal blocks
        dwfull = _empty((self.ld, n * c, 3, 3), dev)
        if tl.mp_mode() and (n * c) % 8 == 0 and self.ld % 8 == 0:
            # bf16 loop: the 320 -> 16 weight gradient on the bf16 GEMM as well (two casts + ~45 us against 207 us on the fp32 kernel)
            tops.mp_conv3x3_wgrad(tl.as_bf16(mid), tl.as_bf16(dhead), n * c, self.ld, 1, dwfull)
        else:
            tops.conv3x3_wgrad(mid, dhead, n * c, self.ld, 1, dwfull)
        dbias = _empty((self.ld,), dev)
        tops.colsum(dhead, self.ld, dbias)
        for i, sq in enumerate(self.seqs):
            ensure_grad(sq[1].weight).copy_(dwfull[self.offs[i]:self.offs[i + 1], c * i:c * (i + 1)])
            ensure_grad(sq[1].bias).copy_(dbias[self.offs[i]:self.offs[i + 1]])
        w, b, cp = f['bw2']
        dmid = ops.conv3x3(dhead, w, b, self.ld, n * c, cp, stride=1, relu=False)
        for i, layer in enumerate(self.stage1):
            layer.backward(Act(dmid, c * i, c), need_dx=False)                  # BN+ReLU backward in place, wgrad
        ds = _empty(tuple(sv['s'].t.shape), dev)
        ConvBNAct._run3x3(f['bw1'], dmid, n * c, c, 1, ds, 0, 0)
        return self.shared.backward(Act(ds))


# ---------------------------------------------------------------------------------------------------------------------
# AnchorHeadSingle (MODEL.NAME PointPillar)
# ---------------------------------------------------------------------------------------------------------------------

class AnchorHeadTrain:
    """conv_cls / conv_box / conv_dir_cls (1x1 convs with bias, anchor_head_single.py:17-37) as ONE pointwise GEMM over the concatenated
    output channels; the data gradient is the same kernel on the transposed matrix, the weight gradient one pixel-contraction GEMM."""

    def __init__(self, head):
        self.m = head
        self.convs = [head.conv_cls, head.conv_box] + ([head.conv_dir_cls] if head.conv_dir_cls is not None else [])
        self.outs = [c.weight.shape[0] for c in self.convs]
        self.offs = [int(v) for v in np.concatenate([[0], np.cumsum(self.outs)])]
        self.cin = self.convs[0].weight.shape[1]
        self.cout = self.offs[-1]
        self.ld = pack.round_up(self.cout, 16)              # the gradient buffer is a contraction operand of the data-gradient GEMM
        self._step = -1
        self.saved = None

    def _forms(self):
        if self._step == tl.StepClock.step:
            return self._f
        w = torch.cat([c.weight.detach().float().reshape(c.weight.shape[0], -1) for c in self.convs], 0)      # (cout, cin)
        b = torch.cat([c.bias.detach().float() for c in self.convs], 0)
        wt = w.new_zeros((self.cin, self.ld))
        wt[:, :self.cout] = w.t()
        self._f = dict(fw=pack.pack_plain(w, b), bw=pack.pack_plain(wt, w.new_zeros(self.cin)))
        self._step = tl.StepClock.step
        return self._f

    def forward(self, x):
        """x: Act (B, H, W, cin).  Returns the raw head buffer (B, H, W, ld); channels past cout are zero."""
        f = self._forms()
        if x.t.dtype != torch.float32:                     # bf16 loop: the backbone's map arrives as bf16, these 1x1 convs stay fp32
            x = Act(tl.as_f32(x.t, x.off, x.c), 0, x.c)
        B, H, W, _ = x.t.shape
        buf = torch.zeros((B, H, W, self.ld), dtype=torch.float32, device=x.t.device)
        w, b, cp = f['fw']
        ops.pointwise(x.t, w, b, lib.PW_PLAIN, self.cin, self.cout, cp, relu=False, out=buf, in_ch_off=x.off)
        self.saved = x
        return buf

    def backward(self, dhead):
        """dhead: (B, H, W, ld), padding channels zero.  Returns Act gradient of the head input."""
        x = self.saved
        dev = dhead.device
        dw = _empty((self.ld, self.cin), dev)
        tops.pointwise_wgrad(tops.rowmap(dhead, self.ld), tops.rowmap(x.t, self.cin, x.off), x.rows, dw)
        db = _empty((self.ld,), dev)
        tops.colsum(dhead, self.ld, db)
        for i, c in enumerate(self.convs):
            ensure_grad(c.weight).view(self.outs[i], self.cin).copy_(dw[self.offs[i]:self.offs[i + 1]])
            ensure_grad(c.bias).copy_(db[self.offs[i]:self.offs[i + 1]])
        w, b, cp = self._forms()['bw']
        dx = _empty(tuple(x.t.shape[:-1]) + (self.cin,), dev)
        ops.pointwise(dhead, w, b, lib.PW_PLAIN, self.ld, self.cin, cp, relu=False, out=dx)
        return Act(dx)
