"""Reference-compatible plugin surface (`pcdet` API of OpenPCDet v0.6.0 as forked by quan-dao/practical-collab-perception)
for the PointPillars collaborative-perception hot path, backed by hand-written gfx950 kernels (pcp_amd / libpcp_hip.so).

Same registry names, constructor signatures, batch_dict keys and state-dict keys as the reference, so
tools/test.py-style callers and published checkpoints work unchanged; there is no CUDA, cuDNN, torch_scatter or spconv
dependency and no CPU fallback.
"""
__version__ = '0.6.0+pcp_amd'
