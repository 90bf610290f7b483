#!/usr/bin/env python3
"""The HBM-bound streaming kernel exactly as bench.py's `roofline_hbm` times it (one cross-covariance column over 1M x 256
rows, m = 21): the program tools/profile_r5.sh runs under rocprofv3 for the kernel's duration and its HBM counters."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

print(json.dumps(bench.hbm_stream_probe(torch.device("cuda", 0))))
