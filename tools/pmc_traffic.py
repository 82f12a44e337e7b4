"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dirF> -- <cmd>
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dirW> -- <cmd>
    python tools/pmc_traffic.py <dirF> <dirW> <out.json> "<cmd>" [passes of the workload in <cmd>]

Units and corrections as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 B, so it is DOUBLED; WRITE_SIZE is taken as is.
Output: {"command", "kernels": {name: {"launches", "fetch_bytes", "write_bytes", "traffic_bytes"}}} (means per launch)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def mean_per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = mean_per_kernel(sys.argv[1], "FETCH_SIZE"), mean_per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"command": sys.argv[4] if len(sys.argv) > 4 else "", "units": "bytes per launch (mean); FETCH_SIZE KiB x 1024 x 2 (gfx950 "
       "correction), WRITE_SIZE KiB x 1024", "kernels": {}}
for k in sorted(set(fetch) & set(write), key=lambda k: -(fetch[k][0] * fetch[k][1])):
    fb, wb = 2 * 1024 * fetch[k][0], 1024 * write[k][0]
    out["kernels"][k[:160]] = {"launches": fetch[k][1], "fetch_bytes": round(fb), "write_bytes": round(wb), "traffic_bytes": round(fb + wb)}
if len(sys.argv) > 5:
    npass = int(sys.argv[5])
    out["passes_of_the_workload"] = npass
    out["collected"] = ("MI355X, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace "
                        "(tools/pmc_traffic.py)")
    out["total_traffic_bytes_per_pass"] = round(sum(v["traffic_bytes"] * v["launches"] for v in out["kernels"].values()) / npass)
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in list(out["kernels"].items())[:12]:
    print(f"{v['launches']:6d} x  fetch {v['fetch_bytes'] / 1e6:9.2f} MB  write {v['write_bytes'] / 1e6:9.2f} MB   {k[:100]}")
