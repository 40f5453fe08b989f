"""Instruction histogram of the LK march loop of one stream_kernel / lk_level_kernel instantiation.

usage: python tools/isa_hist.py [R] [MODE] [stream|level]   (compiles lk_level.hip to ISA under /tmp, needs hipcc only)
The march is unrolled three times, so counts are divided by 3 to give one 256-column row step.
"""
import collections, os, re, subprocess, sys

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
MODE = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kind = sys.argv[3] if len(sys.argv) > 3 else "stream"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "cuda_optical_flow_2_amd", "csrc")
out = "/tmp/ofx_isa.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950",
                "-I" + os.path.join(root, "include"), "-I" + csrc, "-S", "--cuda-device-only", "-o", out,
                os.path.join(csrc, ("lk_inst_stream_" if kind == "stream" else "lk_inst_levels_") + ("c" if MODE == 0 else ("fast" if int(os.environ.get("OFX_ISA_FAST", "0")) else "f")) + ".hip")] + os.environ.get("OFX_BUILD_DEFS", "").split(), check=True, stderr=subprocess.DEVNULL)
FAST = int(os.environ.get("OFX_ISA_FAST", "0"))   # 1: the <= 1 ulp solve instantiation
name = f"stream_kernelILi{R}ELi{MODE}ELb{FAST}ELb{int(os.environ.get('OFX_ISA_DMA', '0'))}ELi0EEE" if kind == "stream" else f"lk_level_kernelILi{R}ELi{MODE}ELb0ELb{FAST}EEE"
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
while "NumVgprs" not in lines[end]:
    end += 1
body = lines[start:end + 12]
for l in body:
    if re.search(r"NumVgprs|ScratchSize|Occupancy|TotalNumSgprs", l):
        print(l.strip())
# the march: the loop that contains v_dot2c (the biggest one)
heads = [i for i, l in enumerate(body) if "Loop Header" in l and "Depth=1" in l]
best = None
for h in heads:
    label = body[h].split(":")[0]
    # the loop's blocks are tagged "in Loop: Header=<label>"; it ends with the last of them (up to the next label)
    tag = "Header=" + label.lstrip(".L") + " "
    tagged = [i for i, l in enumerate(body) if tag in l + " "]
    last = max([h] + tagged)
    while last + 1 < len(body) and not body[last + 1].startswith(".LBB") and "s_endpgm" not in body[last + 1]:
        last += 1
    seg = body[h:last + 1]
    n = sum("v_dot2c" in l for l in seg)
    if best is None or n > best[0]:
        best = (n, seg)
seg = best[1]
ops = collections.Counter()
for l in seg:
    l = l.strip()
    if not l or l.startswith(";") or l.startswith("."):
        continue
    ops[l.split()[0]] += 1
valu = sum(c for o, c in ops.items() if o.startswith("v_"))
salu = sum(c for o, c in ops.items() if o.startswith("s_"))
print(f"march loop: {len(seg)} lines; per step (x1/3): VALU {valu/3:.0f}  SALU {salu/3:.0f}  vmem {sum(c for o,c in ops.items() if o.startswith('global_'))/3:.1f}")
for o, c in ops.most_common(45):
    print(f"  {c/3:6.1f}  {o}")
