"""Stamp profiles/traffic_latest.json with the commit it was measured at (the GPU box has no .git): run in the repo after
copying gpurun_out/prof_<tag>/summary/traffic_latest.json into profiles/, BEFORE committing anything that touches the kernels.
bench.py reports the figure only while the kernel sources still hash to kernel_source_sha16."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench
p = os.path.join(root, "profiles", "traffic_latest.json")
d = json.load(open(p))
d["measured_at_commit"] = subprocess.check_output(["git", "-C", root, "rev-parse", "HEAD"], text=True).strip()
if d.get("kernel_source_sha16") != bench.kernel_source_hash():
    print("warning: the kernel sources differ from the ones the profile was taken with:", d.get("kernel_source_sha16"), "!=", bench.kernel_source_hash())
json.dump(d, open(p, "w"), indent=1)
print(p, "<-", d["measured_at_commit"])
