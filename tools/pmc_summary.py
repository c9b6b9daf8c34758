"""Summarise rocprofv3 --pmc counter-collection CSVs: mean counter value per launch, per kernel.

Usage: python tools/pmc_summary.py OUT.json DIR [DIR ...]
Every DIR is the -d directory of one `rocprofv3 --pmc ... --output-format csv` pass.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"<.*$", "", name)           # drop template arguments
    return name.strip()


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    a = acc[k][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    res = {k: {c: round(v[0] / v[1], 2) for c, v in sorted(cs.items())} | {"launches": max(v[1] for v in cs.values())}
           for k, cs in sorted(acc.items())}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print("wrote", out, "kernels:", len(res))


if __name__ == "__main__":
    main()
