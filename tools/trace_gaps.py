"""Timeline of the stream kernel from a rocprofv3 --kernel-trace CSV: per dispatch start/end (us, relative), queue, overlap
with the previous dispatch.  python tools/trace_gaps.py <dir> [kernel substring] [first] [count]"""
import csv, glob, os, sys
root = sys.argv[1]; name = sys.argv[2] if len(sys.argv) > 2 else "stream_kernel"
first = int(sys.argv[3]) if len(sys.argv) > 3 else 100; count = int(sys.argv[4]) if len(sys.argv) > 4 else 16
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if name in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
rows.sort()
t0 = rows[first][0]
prev_end = None
for s, e, q in rows[first:first + count]:
    print(f"q{q} start {(s - t0) / 1e3:8.1f} end {(e - t0) / 1e3:8.1f} dur {(e - s) / 1e3:6.1f}" + (f"  start-prev_end {(s - prev_end) / 1e3:7.1f}" if prev_end else ""))
    prev_end = e
n = len(rows)
span = (rows[-20][0] - rows[40][0]) / (n - 60) / 1e3
print(f"{n} dispatches, mean period {span:.1f} us, mean duration {sum(e - s for s, e, _ in rows[40:-20]) / (n - 60) / 1e3:.1f} us")
