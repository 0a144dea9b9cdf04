"""BASELINE configs C3 and C5 as JSON objects with a `roofline` each (the code lives in bench.py: `config_c3`, `config_c5`; the
default `python bench.py` run emits both under `configs`):

    python tools/config_roofline.py c3     # LargeVis N = 1M, D = 128, kNN width 15 (perplexity 5), 500 iterations
    python tools/config_roofline.py c3b    # the same with kNN width 45
    python tools/config_roofline.py c5     # symmetric entropic affinity N = 200k, D = 64, perplexity 30: dual iterations
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    for w in sys.argv[1:] or ["c3", "c5"]:
        if w == "c3":
            print(json.dumps(bench.config_c3(dev)), flush=True)
        elif w == "c3b":
            print(json.dumps(bench.config_c3(dev, width=45)), flush=True)
        elif w == "c5":
            print(json.dumps(bench.config_c5(dev)), flush=True)
