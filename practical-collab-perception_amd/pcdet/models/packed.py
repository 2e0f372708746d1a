"""Mixin that caches the HIP-side form of a module's weights (BatchNorm folded, repacked for the MFMA tiles) and drops the
cache whenever the parameters may have changed: .to()/.cuda() (_apply), load_state_dict, train()."""
import torch.nn as nn


def _flush_counters(module, prefix, keep_vars):
    """state_dict pre-hook: the deferred num_batches_tracked increments of the training path land before the counters are read"""
    from pcp_amd.train_layers import flush_batches_tracked
    flush_batches_tracked()


class PackedModule(nn.Module):
    def __init__(self):
        super().__init__()
        self._pcp_cache = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._weights_replaced())
        self.register_state_dict_pre_hook(_flush_counters)

    def _weights_replaced(self):
        """load_state_dict: the folded inference weights AND the per-step packed forms of the train-mode layers are stale; BatchNorm
        counter increments collected before the load no longer apply to the loaded counters"""
        self.invalidate_packed()
        from pcp_amd.train_layers import StepClock, drop_pending_batches_tracked
        drop_pending_batches_tracked(self)
        StepClock.tick()

    def invalidate_packed(self):
        self._pcp_cache = None
        for m in self.children():
            if isinstance(m, PackedModule):
                m.invalidate_packed()

    def _apply(self, fn, *args, **kwargs):
        self._pcp_cache = None
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode=True):
        if mode != self.training:          # a frozen teacher flipped to eval() every forward keeps its packed weights
            self._pcp_cache = None
        return super().train(mode)

    def packed(self):
        """dict built lazily by the subclass's _build_packed(); valid until the next invalidation."""
        if self._pcp_cache is None:
            self._pcp_cache = self._build_packed()
        return self._pcp_cache

    def _build_packed(self):
        raise NotImplementedError


def require_eval_hip(module, what):
    if module.training:
        raise NotImplementedError(
            '%s: this module has no training kernels (the HIP training path covers config 5: DynPillarVFE, PointPillarScatter, '
            'BaseBEVBackbone, V2XMidFusionDisco, CenterHead) -- call model.eval()' % what)


def train_tape(batch_dict):
    """list of backward closures, appended in forward order by the train-mode modules; the detector runs it reversed."""
    tape = batch_dict.get('_pcp_tape')
    if tape is None:
        from .detectors.centerpoint import Tape
        tape = batch_dict['_pcp_tape'] = Tape()
    return tape
