#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -q -x -k "pfn or single_agent or disco_mid or full_size or degenerate" 2>&1 | tail -3
python profiles/scripts/r02_pfn_order.py 2>/dev/null | grep random
python profiles/scripts/r02_pfn_order.py 2>/dev/null | grep random
