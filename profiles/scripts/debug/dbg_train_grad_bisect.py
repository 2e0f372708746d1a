"""Diagnostic (not a test; run by hand on the GPU box: python tests/dbg_train_grad_bisect.py): per-parameter gradient error of the HIP
training step vs the float64 / float32 CPU oracle on the g7 fixture, plus a bisection of the fusion backward against oracle probes."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd'))
import test_gpu_train_e2e as T          # noqa: E402
from helpers import load_golden         # noqa: E402

g = load_golden('g7_train.npz')
names = [str(n) for n in g['trainable']]
model = T._build(g)
batch, metadata = T._batch(g)
torch.set_num_threads(16)
from oracle import train as otr
from helpers import arch_of
meta = g['meta']
arch = otr.add_train_arch(arch_of(meta), meta['model'])
from pcp_amd import synth
st = otr.make_state(synth.fill_state_dict(meta['state_shapes']))
st = {k: (v.detach().double().requires_grad_(v.requires_grad) if (v.dtype == torch.float32 and not k.startswith('bev_maker')) else v) for k, v in st.items()}
probe = {}
loss_p, _, aux_p = otr.train_forward(g['points'], g['gt_boxes'], metadata, st, arch, probe=probe)
loss_p.backward()
PROBE = dict(d_out=aux_p['fused'].grad, d_mid=probe['decomp_mid'].grad, d_fused=probe['fused_in'].grad, d_ego=probe['ego_compressed'].grad,
             d_bb=aux_p['backbone_out'].grad, mid=probe['decomp_mid'].detach(), fused=probe['fused_in'].detach())
g64, loss64, _, _ = T._oracle_grads(g, metadata, torch.float64)
g32, loss32, _, _ = T._oracle_grads(g, metadata, torch.float32)
model.train()
import pcdet.models.train_path as TP
_orig = TP.FusionTrain.backward
def spy(self, dout):
    def cmp(tag, mine, ref):
        mine = mine.permute(0, 3, 1, 2).double().cpu()
        print('   probe %-8s rel err %.3e (scale %.3e)' % (tag, float((mine - ref).abs().max()) / float(ref.abs().max()), float(ref.abs().max())))
    cmp('d_out', dout.t.clone(), PROBE['d_out'])
    g1 = self.d1.backward(dout)
    cmp('d_mid', g1.t.clone(), PROBE['d_mid'])
    cmp('mid(fwd)', self.d1.saved[0].t, PROBE['mid'])
    g0 = self.d0.backward(g1)
    cmp('d_fused', g0.t, PROBE['d_fused'])
    self.d1.backward = lambda d: g1
    self.d0.backward = lambda d: g0
    r = _orig(self, dout)
    cmp('d_bb', r.t, PROBE['d_bb'])
    return r
TP.FusionTrain.backward = spy
ret, tb, _ = model(batch)
ret['loss'].backward()
params = dict(model.named_parameters())
print('loss hip %.7f  f64 %.7f  f32 %.7f' % (float(ret['loss'].detach()), loss64, loss32))
for n in names:
    mine = params[n].grad.detach().double().cpu()
    ex = g64[n]
    sc = float(ex.abs().max())
    print('%-55s scale %.3e  hip %.2e  f32cpu %.2e' % (n, sc, float((mine - ex).abs().max()) / max(sc, 1e-12),
                                                       float((g32[n] - ex).abs().max()) / max(sc, 1e-12)))
