#!/usr/bin/env python3
"""The training leg of bench.py on its own (for rocprofv3 --kernel-trace --stats):
    python tools/train_leg_only.py [stride] [precision]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    stride = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
    r = bench.train_leg(torch.device("cuda", 0), stride=stride, precision=prec)
    print(json.dumps(r))
