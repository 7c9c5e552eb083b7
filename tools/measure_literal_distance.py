#!/usr/bin/env python3
"""Prints the end-to-end distance of the HIP network to the reference-literal numerics (tests/test_gpu_network.py literal_distance) as JSON.
Run on the GPU box: python tools/measure_literal_distance.py [samples]"""
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests")]
import test_gpu_network as t  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
print(json.dumps([t.literal_distance(n), t.literal_distance(n // 4, weight_scale=16.0)], indent=1))
