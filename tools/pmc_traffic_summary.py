"""Average a rocprofv3 --pmc counter over the launches of the 256x256 bf16 GEMM kernel: `python tools/pmc_traffic_summary.py <counter_collection.csv>`."""
import collections
import csv
import sys

acc, n = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_bf16_nt_256_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Counter_Name"]] += 1
for c in acc:
    print(c, acc[c] / n[c], "per launch over", n[c], "launches")
