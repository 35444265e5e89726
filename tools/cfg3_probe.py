#!/usr/bin/env python
"""BASELINE cfg 3 alone (the `cfg3` block of bench.py: released architecture, 8 views, cross_entropy_linemod, 128 renders per
iteration scored on the fused engine), as a workload for rocprofv3:

    rocprofv3 --kernel-trace --stats --output-format csv -d out -o cfg3 -- python tools/cfg3_probe.py
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA ... -- python tools/cfg3_probe.py

Prints the block as JSON."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if os.environ.get('LF_HIP_LIB'):                                    # A/B of two BUILDS of the library (tool only)
    from latentfusion_amd import _lib
    _lib.LIB_PATH = os.environ['LF_HIP_LIB']

a = argparse.Namespace(cfg3_iters=int(sys.argv[1]) if len(sys.argv) > 1 else 10)
print(json.dumps(bench.cfg3_report(a, 'cuda:0')))
