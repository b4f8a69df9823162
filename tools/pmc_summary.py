"""Aggregate rocprofv3 --pmc CSVs: mean counter value per dispatch of the RX kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "rx512_kernel"
acc = defaultdict(list)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if kern in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-28s mean %.6g  (n=%d, min %.6g, max %.6g)" % (k, sum(v) / len(v), len(v), min(v), max(v)))
