"""Kernel times of the wider tensor-product descriptors of the bench line (bench.measure_tensor_forms): coefficient gradients (Q3),
a vector-valued space ((Q2)^3) and the Helmholtz operator on Q6 / Q7 (column chunks, plane-wise point weights)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

print(json.dumps(bench.measure_tensor_forms(3, 1), indent=1))
