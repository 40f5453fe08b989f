"""Reduce rocprofv3 --pmc counter CSVs to per-launch HBM traffic for one kernel.
    python tools/pmc_parse.py <dir with *_counter_collection.csv (searched recursively)> <kernel name substring> [skip_first]
Prints JSON: mean FETCH_SIZE / WRITE_SIZE (KiB) per dispatch and bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB -- the gfx950
correction of MI355X_MICROARCH.md's HBM section (FETCH_SIZE tallies 128-B requests at 64 B)."""
import csv, glob, json, os, sys

root, kname = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
vals = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per_disp = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if kname not in row.get("Kernel_Name", ""):
                continue
            key = (row["Counter_Name"], int(row["Dispatch_Id"]))
            per_disp[key] = per_disp.get(key, 0.0) + float(row["Counter_Value"])
    by_counter = {}
    for (c, d), v in sorted(per_disp.items(), key=lambda kv: kv[0][1]):
        by_counter.setdefault(c, []).append(v)
    for c, lst in by_counter.items():
        lst = lst[skip:] if len(lst) > skip else lst
        vals.setdefault(c, []).extend(lst)
out = {c: sum(v) / len(v) for c, v in vals.items()}
res = {"kernel": kname, "launches": {c: len(v) for c, v in vals.items()}, "fetch_kib": out.get("FETCH_SIZE"), "write_kib": out.get("WRITE_SIZE")}
if res["fetch_kib"] is not None and res["write_kib"] is not None:
    res["bytes"] = int((2 * res["fetch_kib"] + res["write_kib"]) * 1024)
print(json.dumps(res))
