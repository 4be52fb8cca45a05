#!/usr/bin/env python3
"""The gossip training leg of bench.py on its own (for rocprofv3 --kernel-trace --stats):
    python tools/train_gossip_only.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    print(json.dumps(bench.train_gossip_leg(torch.device("cuda", 0))))
