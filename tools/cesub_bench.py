#!/usr/bin/env python3
"""The change-estimation-subset path (monolithic score_generic_kernel) at the sizes bench.py times it: Iris-shaped 150 x 4
and USPS-shaped 9298 x 256, subset of 5, k = 4 (tools/profile_r5.sh cesub)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import bench  # noqa: E402

dev = torch.device("cuda", 0)
print(json.dumps(bench.cesub_workload(dev, 150, 4, rounds=5, length_scale=0.5)))
print(json.dumps(bench.cesub_workload(dev, 9298, 256, length_scale=3.0)))
