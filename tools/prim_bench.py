"""timing of the stand-alone window-sum entry points (bench.py's extra.primitives leg alone): python tools/prim_bench.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

class A:  # the arguments Run needs
    pass
sys.argv = ["bench.py", "--iters", "1"]
args = bench.parse_args()
bench.plan_stream(args)
args.ring = 4
run = bench.Run(args)
print(json.dumps({k: {"us": v["avg_launch_us"], "frac": v["frac"]} for k, v in run.leg_primitives().items()}))
