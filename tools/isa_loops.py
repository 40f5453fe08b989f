"""Instruction histogram of the LK march loops of one kernel in an ISA listing (hipcc -S --cuda-device-only).
usage: python tools/isa_loops.py file.s <kernel name substring> [unroll=3]
Prints NumVgprs / scratch, then for every loop that holds v_dot2c: VALU / SALU / memory instructions per step (loop body / unroll),
split into the two issue classes of tools/ubench/valu_rates.hip (cheap: plain 32-bit add / sub / and / or / xor / mov / fp32 add, mul, fma;
full: everything else) and an estimate of the VALU time per step from the measured rates (1.15 / 1.85 ns, v_rcp_f64 6.85)."""
import collections, re, sys
path, name = sys.argv[1], sys.argv[2]
unroll = int(sys.argv[3]) if len(sys.argv) > 3 else 3
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
e2 = end
while "NumVgprs" not in lines[e2]:
    e2 += 1
for l in lines[e2 - 3:e2 + 14]:
    if re.search(r"NumVgprs|ScratchSize|Occupancy|TotalNumSgprs", l):
        print(l.strip())
body = lines[start:end]
heads = [i for i, l in enumerate(body) if "Loop Header" in l and "Depth=1" in l] + [len(body)]
CHEAP = {"v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32", "v_mov_b32_e32", "v_add_f32_e32", "v_sub_f32_e32",
         "v_mul_f32_e32", "v_fma_f32", "v_fmac_f32_e32", "v_add_u32_e64", "v_sub_u32_e64", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
for a, b in zip(heads, heads[1:]):
    seg = body[a:b]
    ops = collections.Counter()
    for l in seg:
        l = l.strip()
        if not l or l.startswith(";") or l.startswith("."):
            continue
        ops[l.split()[0]] += 1
    if not ops["v_dot2c_i32_i16_e32"]:
        continue
    valu = sum(c for o, c in ops.items() if o.startswith("v_"))
    cheap = sum(c for o, c in ops.items() if o in CHEAP)
    salu = sum(c for o, c in ops.items() if o.startswith("s_"))
    mem = sum(c for o, c in ops.items() if o.startswith(("buffer_", "global_", "scratch_", "ds_", "flat_")))
    ns = (cheap * 1.15 + (valu - cheap - ops["v_rcp_f64_e32"]) * 1.85 + ops["v_rcp_f64_e32"] * 6.85) / unroll
    print(f"loop @{a}: per step VALU {valu/unroll:.0f} (cheap {cheap/unroll:.0f})  SALU {salu/unroll:.0f}  mem {mem/unroll:.1f}  scratch {sum(c for o,c in ops.items() if o.startswith('scratch_'))/unroll:.1f}  ~{ns:.0f} ns VALU")
    print("   " + "  ".join(f"{o}:{c/unroll:.1f}" for o, c in ops.most_common(28)))
