#!/usr/bin/env python
"""BASELINE cfg 5 alone (the `cfg5` block of bench.py: one bf16-autocast generator training step, 32 + 8 views, SYN(128,16)):
    python tools/cfg5_probe.py [steps=8]
Prints the block as JSON."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

a = argparse.Namespace(cfg5_steps=int(sys.argv[1]) if len(sys.argv) > 1 else 8, size=128, channels=16)
print(json.dumps(bench.cfg5_report(a, 'cuda:0')))
