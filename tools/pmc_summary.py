#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch for kernels matching a pattern."""
import csv
import glob
import sys
from collections import defaultdict


def main(pattern, *dirs):
    for d in dirs:
        for f in glob.glob(d + "/*counter_collection.csv"):
            acc = defaultdict(list)
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if pattern in row["Kernel_Name"]:
                        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
            for k, v in sorted(acc.items()):
                print(f"{k:<28} n={len(v):<3} mean={sum(v) / len(v):.6g}")


if __name__ == "__main__":
    main(sys.argv[1], *sys.argv[2:])
