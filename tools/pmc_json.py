#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs (one directory per pass) -> the per-launch JSON summary kept under profiles/.
    python tools/pmc_json.py <kernel substring> <out.json> <config json> <note> <dir> [<dir> ...]
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B for wide
coalesced streaming reads), WRITE_SIZE is taken as is."""
import collections, csv, glob, json, sys
pat, out, config, note, dirs = sys.argv[1], sys.argv[2], json.loads(sys.argv[3]), sys.argv[4], sys.argv[5:]
acc = collections.defaultdict(list)
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in acc.items()}
rec = {"kernel": pat, "config": config, "launches": {k: len(v) for k, v in acc.items()}, "counters_mean_per_launch": mean, "note": note}
if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
    fr, wr = mean["FETCH_SIZE"] * 1024, mean["WRITE_SIZE"] * 1024
    rec["per_launch_bytes"] = {"fetch_raw": fr, "fetch_x2": 2 * fr, "write": wr, "traffic_raw": fr + wr, "traffic_corrected": 2 * fr + wr}
if "TCC_HIT_sum" in mean and "TCC_MISS_sum" in mean:
    rec["l2_hit_rate"] = mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec, indent=1))
