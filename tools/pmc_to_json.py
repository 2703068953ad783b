#!/usr/bin/env python
"""Turn rocprofv3 --pmc CSVs (tools/pmc.sh) into a small JSON with per-kernel HBM traffic.
HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KB: MI355X_MICROARCH.md (HBM section) -- on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced streaming reads; WRITE_SIZE is taken as reported."""
import csv
import glob
import json
import sys
from collections import defaultdict


def main(tag, out):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"gpurun_out/{tag}_pmc_*/*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0]
                if name.startswith("wrap_"):
                    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for k, c in acc.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        d = {"counters_mean_per_dispatch": m}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            d["hbm_bytes_per_launch"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
            d["fetch_kb_raw"], d["write_kb_raw"] = m["FETCH_SIZE"], m["WRITE_SIZE"]
        res[k] = d
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in res.items()}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
