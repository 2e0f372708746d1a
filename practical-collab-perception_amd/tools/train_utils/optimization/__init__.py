"""build_optimizer / build_scheduler with the reference's signatures (tools/train_utils/optimization/__init__.py:11-62) for the
`adam_onecycle` recipe every V2X-Sim config uses (OPTIMIZATION section of v2x_pointpillar_disco.yaml:175-194).

MI355X-first design: all trainable parameters live in ONE flat fp32 buffer (param.data become views of it, param.grad views of
a flat gradient buffer), so
  * gradient clipping is one squared-norm reduction (pcp_grad_sqnorm) whose result never leaves the device,
  * the optimizer step is ONE launch (pcp_adam_step: clip coefficient, decoupled weight decay, Adam moments, update),
  * data-parallel training needs ONE RCCL all-reduce of the flat gradient buffer per step (19 MB for config 5).
The reference's OptimWrapper splits BN / non-BN groups but applies the same lr, wd (bn_wd=True, true_wd=True) and betas to
both (fastai_optim.py:104-122), so one flat group is arithmetically identical.
"""
import math

import numpy as np
import torch

from pcp_amd import train_layers as tl
from pcp_amd import train_ops as tops


def annealing_cos(start, end, pct):
    return end + (start - end) / 2 * (np.cos(np.pi * pct) + 1)


def all_reduce_flat_gradient(flat_g, group=None):
    """the ONE collective of a data-parallel training step: sum of the flat fp32 gradient over ranks (RCCL on GPUs, gloo in the CPU
    tests); the 1 / world averaging of DistributedDataParallel is folded into the optimizer kernel's grad_scale"""
    if _collectives_on(group):
        torch.distributed.all_reduce(flat_g, group=group)
        return 1.0 / torch.distributed.get_world_size(group)
    return 1.0


def _collectives_on(group=None):
    """more than one rank -- or PCP_FORCE_COLLECTIVES=1 in an initialised one-rank group (the RCCL path exercised on a one-GPU box)"""
    import os
    d = torch.distributed
    if not (d.is_available() and d.is_initialized()):
        return False
    return d.get_world_size(group) > 1 or os.environ.get('PCP_FORCE_COLLECTIVES', '0') == '1'


class OverlappedFlatReduce:
    """The two-bucket gradient all-reduce of one optimizer: the TAIL of the flat gradient (the parameters whose backward finishes first)
    is reduced asynchronously while the rest of the backward pass runs, the head follows in finish().

    Correct for any number of backward() calls per step (round 4's in-place form was not): the tail is reduced in a COPY, so the live
    buffer keeps this rank's own sums.  One backward since the last finish(): the reduced copy replaces the tail and only the head is
    reduced in finish().  A second backward (gradient accumulation) adds to the live buffer after the copy was taken: finish() then drops
    the copy and reduces the whole buffer once.  abandon() (zero_grad without a step) waits for the copy and drops it.
    Works on any tensor torch.distributed can reduce (the CPU test drives it with gloo)."""

    def __init__(self, flat_g, tail_off, group=None):
        self.flat_g, self.tail_off, self.group = flat_g, int(tail_off), group
        self.work, self.copy, self.backwards = None, None, 0
        self.started = 0                                   # asynchronous tail reductions issued (tests)

    def grad_ready(self):
        """the tail's gradients of one backward pass are complete (queued)"""
        self.backwards += 1
        if self.backwards == 1:
            self.copy = self.flat_g[self.tail_off:].clone()
            self.work = torch.distributed.all_reduce(self.copy, group=self.group, async_op=True)
            self.started += 1

    def finish(self):
        """reduces what is left; returns the 1 / world scale the optimizer kernel folds in"""
        world = torch.distributed.get_world_size(self.group)
        if self.work is None:
            torch.distributed.all_reduce(self.flat_g, group=self.group)
        elif self.backwards == 1:
            torch.distributed.all_reduce(self.flat_g[:self.tail_off], group=self.group)
            self.work.wait()
            self.flat_g[self.tail_off:].copy_(self.copy)
        else:
            self.work.wait()                               # the copy misses the later backward passes: one full reduction instead
            torch.distributed.all_reduce(self.flat_g, group=self.group)
        self.work, self.copy, self.backwards = None, None, 0
        return 1.0 / world

    def abandon(self):
        if self.work is not None:
            self.work.wait()
        self.work, self.copy, self.backwards = None, None, 0


class FlatAdamOneCycle:
    """optimizer.zero_grad() / .step() / .lr / .mom like the reference's OptimWrapper; `clip_grad_norm(max_norm)` replaces
    torch.nn.utils.clip_grad_norm_ (it only records max_norm: the scaling happens inside the fused step)."""

    def __init__(self, model, wd=0.01, beta2=0.99, eps=1e-8, process_group=None):
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev = self.params[0].device
        if dev.type != 'cuda':
            raise RuntimeError('the fused optimizer runs on the GPU: move the model first (model.cuda())')
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]            # 16-byte aligned views
        total = sum(sizes)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p, sz in zip(self.params, sizes):
            view = self.flat_p[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[off:off + p.numel()].view(p.shape)
            off += sz
        self.sqnorm = torch.zeros(1, dtype=torch.float64, device=dev)
        self.wd, self.beta2, self.eps = wd, beta2, eps
        self.lr, self.mom = 0.0, 0.9
        self.t = 0
        self.max_norm = None
        self.process_group = process_group
        self.world = torch.distributed.get_world_size(process_group) if (torch.distributed.is_available() and
                                                                          torch.distributed.is_initialized()) else 1
        self.ref_order = reference_param_order(model, self.params)

    # ---- gradient all-reduce overlapped with the backward pass (two buckets) -------------------------------------------------------------
    def attach_overlap(self, model, first_done=('dense_head', 'v2x_mid_fusion')):
        """Splits the flat gradient where the parameters of the modules that finish FIRST in the backward pass begin (`first_done`, in
        registration order the LAST ones: ... backbone | fusion | head) and all-reduces that tail asynchronously as soon as their
        backward closures have been queued; the head of the buffer follows in step().  A no-op without a multi-rank group (or
        PCP_FORCE_COLLECTIVES=1), or when those parameters are not one contiguous tail of the flat buffer."""
        self._tail_off, self._tail_trigger, self._overlap = None, None, None
        if not _collectives_on(self.process_group):
            return False
        owner = {}
        for mname, mod in model.named_children():
            for p in mod.parameters():
                owner[id(p)] = mname
        off, tail_start, seen_tail = 0, None, False
        for p in self.params:
            in_tail = owner.get(id(p)) in first_done
            if in_tail and tail_start is None:
                tail_start = off
            if seen_tail and not in_tail:
                return False                               # not contiguous: keep the single all-reduce
            seen_tail = seen_tail or in_tail
            off += (p.numel() + 3) // 4 * 4
        if tail_start is None or tail_start == 0:
            return False
        present = [n for n in first_done if hasattr(model, n) and getattr(model, n) is not None]
        self._tail_off = tail_start
        self._tail_trigger = present[-1]                   # the tape runs in reverse registration order: the last listed finishes last
        self._overlap = OverlappedFlatReduce(self.flat_g, tail_start, self.process_group)
        model._pcp_grad_ready_hook = self._grad_ready      # carried by the loss node of THIS model's forwards only (centerpoint.hip_loss)
        return True

    def detach_overlap(self, model):
        """back to the single all-reduce in step()"""
        if self._overlap is not None:
            self._overlap.abandon()
        self._tail_off, self._tail_trigger, self._overlap = None, None, None
        if getattr(model, '_pcp_grad_ready_hook', None) == self._grad_ready:
            model._pcp_grad_ready_hook = None

    def _grad_ready(self, name):
        if self._overlap is None or name != self._tail_trigger:
            return
        self._overlap.grad_ready()
        self.overlapped_reductions = self._overlap.started

    def zero_grad(self):
        off = 0
        for p in self.params:                                               # re-attach views a foreign zero_grad(set_to_none) dropped
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off:off + n].view(p.shape)
            off += (n + 3) // 4 * 4
        if getattr(self, '_overlap', None) is not None:
            self._overlap.abandon()                                         # a step that was skipped: its tail reduction is dropped
        tops_zero(self.flat_g)

    def clip_grad_norm(self, max_norm):
        """Records the clipping threshold; returns the device scalar holding the squared global norm (valid after step())."""
        self.max_norm = float(max_norm)
        return self.sqnorm

    def step(self):
        if getattr(self, '_overlap', None) is not None:
            # bucket 2 (head + fusion) has been in flight since their backward was queued; bucket 1 is the rest
            scale = self._overlap.finish()
        else:
            scale = all_reduce_flat_gradient(self.flat_g, self.process_group)   # RCCL, one bucket; DDP-style averaging via grad_scale
        self.t += 1
        if self.max_norm is not None:
            tops.grad_sqnorm(self.flat_g, out=self.sqnorm)
        tops.adam_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.lr, self.mom, self.beta2, self.eps, self.wd, self.t,
                       max_norm=self.max_norm or 0.0, sqnorm=self.sqnorm if self.max_norm is not None else None, grad_scale=scale)
        tl.StepClock.tick()                                                 # packed weight forms are stale now

    def grad_norm(self):
        """host value of the last global gradient norm (syncs; logging only)"""
        return math.sqrt(float(self.sqnorm.item())) / self.world

    def state_dict(self):
        return {'t': self.t, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq, 'lr': self.lr, 'mom': self.mom}

    def load_state_dict(self, sd):
        if 'state' in sd and 'param_groups' in sd:              # a reference checkpoint: torch.optim.Adam's dict under fastai's OptimWrapper
            return self._load_reference_state(sd)
        self.t = int(sd['t'])
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.lr, self.mom = float(sd['lr']), float(sd['mom'])
        return True

    def _load_reference_state(self, sd):
        """maps the reference optimizer's per-parameter moments into the flat buffers.  Its parameter numbering (fastai_optim.py:16-27,
        104-122): leaf modules in registration order, non-BatchNorm leaves first, BatchNorm leaves second, trainable tensors only.  Any
        count or shape mismatch leaves the moments at zero and says so (the model weights are unaffected)."""
        import logging
        log = logging.getLogger(__name__)
        ids = [i for g in sd['param_groups'] for i in g['params']]
        order = self.ref_order
        ok = len(ids) == len(order)
        if ok:
            for k, pi in zip(ids, order):
                st = sd['state'].get(k)
                if st is not None and tuple(st['exp_avg'].shape) != tuple(self.params[pi].shape):
                    ok = False
                    break
        if not ok:
            log.warning('optimizer_state of this checkpoint does not line up with the model (reference torch-Adam format, %d tensors vs %d): '
                        'moments start from zero', len(ids), len(order))
            return False
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        t = 0
        for k, pi in zip(ids, order):
            st = sd['state'].get(k)
            if st is None:
                continue
            n = self.params[pi].numel()
            self.exp_avg[offs[pi]:offs[pi] + n].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[offs[pi]:offs[pi] + n].copy_(st['exp_avg_sq'].reshape(-1))
            t = max(t, int(st['step'].item() if torch.is_tensor(st['step']) else st['step']))
        self.t = t
        return True


def reference_param_order(model, params):
    """indices into `params` in the order the reference's optimizer numbers its tensors (see _load_reference_state)"""
    def leaves(m):
        ch = list(m.children())
        return sum((leaves(c) for c in ch), []) if ch else [m]
    bn = (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)
    index = {id(p): i for i, p in enumerate(params)}
    groups = ([], [])
    for leaf in leaves(model):
        for p in leaf.parameters():
            if id(p) in index:
                groups[1 if isinstance(leaf, bn) else 0].append(index[id(p)])
    return groups[0] + groups[1]


def tops_zero(t):
    from pcp_amd import ops
    ops.fill_zero(t)


class OneCycle:
    """learning_schedules_fastai.py:44-77: cosine lr low -> max over the first pct_start, then max -> low / 1e4; beta1 moms[0] ->
    moms[1] -> moms[0].  step(it) is called BEFORE each iteration (train_utils.py:39)."""

    def __init__(self, optimizer, total_step, lr_max, moms, div_factor, pct_start):
        self.optimizer, self.total_step = optimizer, total_step
        self.lr_max, self.moms, self.div_factor, self.pct_start = lr_max, list(moms), div_factor, pct_start
        low = lr_max / div_factor
        a1 = int(pct_start * total_step)
        self.lr_phases = ((0, a1, low, lr_max), (a1, total_step, lr_max, low / 1e4))
        self.mom_phases = ((0, a1, self.moms[0], self.moms[1]), (a1, total_step, self.moms[1], self.moms[0]))
        optimizer.lr, optimizer.mom = low, self.moms[0]

    def step(self, step):
        for s, e, a, b in self.lr_phases:
            if step >= s:
                self.optimizer.lr = float(annealing_cos(a, b, (step - s) / (e - s)))
        for s, e, a, b in self.mom_phases:
            if step >= s:
                self.optimizer.mom = float(annealing_cos(a, b, (step - s) / (e - s)))


def build_optimizer(model, optim_cfg):
    if optim_cfg.OPTIMIZER != 'adam_onecycle':
        raise NotImplementedError('the fused optimizer implements adam_onecycle (the recipe of every V2X-Sim config), got %s' % optim_cfg.OPTIMIZER)
    opt = FlatAdamOneCycle(model, wd=optim_cfg.WEIGHT_DECAY, beta2=0.99)
    opt.attach_overlap(model)           # multi-rank groups only: head + fusion gradients reduced under the backbone's backward
    return opt


def build_scheduler(optimizer, total_iters_each_epoch, total_epochs, last_epoch, optim_cfg):
    total_steps = total_iters_each_epoch * total_epochs
    sched = OneCycle(optimizer, total_steps, optim_cfg.LR, list(optim_cfg.MOMS), optim_cfg.DIV_FACTOR, optim_cfg.PCT_START)
    return sched, None
