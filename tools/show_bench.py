"""one-screen summary of a bench.py JSON line:  python tools/show_bench.py gpurun_out/<dir>/bench.json"""
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"value {d['value']} {d['unit']}  ms/step {d['ms_per_step']}  frames/step {d['frames_per_step']}  check {d['self_check']}")
print(f"roofline[{r.get('kind')}] frac {r['frac']} launch {r['avg_launch_us']} us  whole pair: frac {r['whole_pair']['frac']} {r['whole_pair']['kernel_us_per_pair']} us/pair  traffic {r.get('traffic')}")
if "power" in d:
    print("power", d["power"].get("board_w"), "W of", d["power"].get("limit_w"), " sclk", d["power"].get("sclk_mhz"), "MHz")
for k, v in r["whole_pair"]["launches"].items():
    print(f"   {k:12s} {v['avg_us']:9.2f} us x {v['launches_per_pair']}")
for k, v in d.get("extra", {}).items():
    if not isinstance(v, dict):
        continue
    if "value" in v or "roofline" in v:
        rr = v.get("roofline", {})
        print(f"{k:34s} {str(v.get('value')):>10s}  frac {rr.get('frac')}  us {rr.get('avg_launch_us', rr.get('kernel_us_per_pair'))}" + (f"  warm {v['warm_ring']['value']} frac {v['warm_ring']['roofline']['frac']}" if "warm_ring" in v else "")
              + (f"  us/pair {v.get('us_per_pair')} {v.get('launches_us')}" if "us_per_pair" in v else ""))
    else:
        for kk, vv in v.items():
            if isinstance(vv, dict):
                print(f"{k}.{kk}: " + ", ".join(f"{a}={b}" for a, b in vv.items() if not isinstance(b, (dict, str)) or a == "kernel")[:230])
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print("cpu", c["value"], c["unit"], c["kind"], "all cores", c.get("all_cores", {}).get("value"))
