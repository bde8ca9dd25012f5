#!/usr/bin/env python3
"""gpurun_out/pmc_c (tools/pmc.sh, the SQ_INSTS_VALU_*_F64 pass) -> profiles/RND_final_fp64.json: fp64 wave-instruction
counts per trade of the pricing kernel; bench.py turns them into TFLOP/s with its own kernel time (roofline.fp64_valu)."""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _identity
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1e6
agg, kernel = collections.defaultdict(list), None
for f in sorted(glob.glob("gpurun_out/pmc_c/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "price_" in r["Kernel_Name"]:
            m = re.search(r"(price_\w+<[^>]*>)", r["Kernel_Name"])
            kernel = m.group(1) if m else r["Kernel_Name"]
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in agg.items()}
out = {"kernel": kernel, "trades_per_launch": n, "launches": len(next(iter(agg.values()), [])),
       "per_launch": mean,
       "per_trade": {"fma_f64": mean.get("SQ_INSTS_VALU_FMA_F64", 0.0) / n, "mul_f64": mean.get("SQ_INSTS_VALU_MUL_F64", 0.0) / n,
                     "add_f64": mean.get("SQ_INSTS_VALU_ADD_F64", 0.0) / n, "trans_f64": mean.get("SQ_INSTS_VALU_TRANS_F64", 0.0) / n,
                     "valu_total": mean.get("SQ_INSTS_VALU", 0.0) / n},
       "note": "wave-instructions (64 lanes each); FMA counts 2 flop per lane, the others 1"}
_identity.stamp(out, "gpurun_out/pmc_c.log")
json.dump(out, open(f"profiles/{rnd}_final_fp64.json", "w"), indent=1)
print(json.dumps(out, indent=1))
