"""pcp_amd -- runtime of the MI355X-native PointPillars hot path: ctypes binding of libpcp_hip.so (lib), tensor-level op
wrappers (ops), weight folding / packing (pack), host-side warp parameters (fusion_host) and the synthetic-input
generator (synth).  The reference-compatible plugin surface lives next to it in the `pcdet` package."""
__version__ = '0.1.0'
