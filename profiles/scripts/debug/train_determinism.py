"""run the first iterations of the full-size DiscoNet training loop TWICE from the same state and report the first iteration whose loss or
parameter digest differs between the runs (per PCP_CONV_ALGO)"""
import os, sys
import numpy as np
import pytest
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, os.path.join(REPO, 'practical-collab-perception_amd'))
sys.path.insert(0, REPO)
import torch
import test_gpu_train_e2e as T


class MP:
    def setenv(self, k, v):
        os.environ[k] = v


def run(algo, iters):
    g, model, opt, ocfg, batch, build_scheduler = T._full_size_disco(MP(), algo)
    sched, _ = build_scheduler(opt, 20, 1, -1, ocfg)
    out = []
    for it in range(iters):
        sched.step(it)
        model.train()
        opt.zero_grad()
        ret, tb, _disp = model(batch())
        model.update_global_step()
        ret['loss'].backward()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        opt.clip_grad_norm(ocfg.GRAD_NORM_CLIP)
        opt.step()
        out.append((float(ret['loss'].detach()), grads))
    del model, opt
    torch.cuda.empty_cache()
    return out


for algo in sys.argv[1:] or ['auto', 'bf16']:
    a, b = run(algo, 4), run(algo, 4)
    for it, ((la, ga), (lb, gb)) in enumerate(zip(a, b)):
        bad = [n for n in ga if not torch.equal(ga[n], gb[n])]
        print('%-6s iteration %d: loss %.9f / %.9f  %s' % (algo, it, la, lb, 'gradients bitwise equal' if not bad else 'DIFFERENT gradients: %d tensors, first %s' % (len(bad), bad[:6])), flush=True)
        if bad:
            break
