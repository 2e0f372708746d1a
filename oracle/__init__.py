"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy for the integer/index work, torch-CPU fp32 for the dense contractions, plain C for the
rotated-IoU NMS) of the reference's PointPillars hot path.  Each function cites the reference file:line it follows.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(practical-collab-perception_amd/) never does, and fails loudly when its HIP library is missing.

Pinning: the reference has no tests of its own for this path, so the oracle is pinned against
outputs of the reference itself, produced in the CPU container by tests/golden/make_golden.py (reference imported
read-only from /root/reference) and committed as tests/golden/*.npz, and against oracle/_ref (the reference's own
iou3d_cpu.cpp compiled where it lies).  See tests/test_oracle_pins.py.
"""
