#!/usr/bin/env python3
"""Copy the summaries of one tools/round_refresh.sh run (gpurun_out/*_TAG*) into profiles/ under the round's prefix.

usage: python tools/collect_profiles.py TAG rNN      (e.g. r02c r02)"""
import csv, glob, json, os, shutil, subprocess, sys
tag, rnd = sys.argv[1], sys.argv[2]
G, P = "gpurun_out", "profiles"
copies = {
    f"bench_{tag}.json": "final_bench.json", f"bench_{tag}_ongrid.json": "final_bench_ongrid.json",
    f"bench_{tag}_flatfwd.json": "final_bench_flatfwd.json", f"bench_{tag}_linfwd.json": "final_bench_linfwd.json",
    f"bench_{tag}_linfwd_200k.json": "final_bench_linfwd_200k.json",
    f"bench_{tag}_config2_delta_100k.json": "final_bench_config2_delta_100k.json",
    f"bench_{tag}_delta_1m.json": "final_bench_delta_1m.json", f"bench_{tag}_value_1m.json": "final_bench_value_1m.json",
    f"bench_{tag}_long_legs.json": "long_legs_bench.json", f"bench_{tag}_payment_lag.json": "payment_lag_bench.json",
    f"bench_{tag}_xccy.json": "xccy_bench.json", f"bench_{tag}_mixed_book.json": "mixed_book_bench.json",
    f"bench_{tag}_curve_build.json": "curve_build_bench.json", f"ablate_{tag}.log": "final_ablations.txt",
    f"pmc_{tag}.txt": "final_pmc_counters.txt", f"pmc_{tag}_lag.txt": "payment_lag_pmc_counters.txt", f"pmc_{tag}_fp64.json": "final_fp64.json",
    f"stamps_{tag}.txt": "final_phase_stamps.txt", f"stamps_{tag}_lag.txt": "payment_lag_phase_stamps.txt",
    f"bench_{tag}_many_pillars.json": "many_pillars_bench.json", f"bench_{tag}_payment_lag_linfwd.json": "payment_lag_linfwd_bench.json",
    f"ablate_{tag}_wide.log": "wide_ablations.txt", f"pmc_{tag}_wide.txt": "wide_pmc_counters.txt",
    f"routing_audit_{tag}.txt": "routing_audit.txt",
    f"bench_{tag}_aggregate_only.json": "aggregate_only_bench.json", f"bench_{tag}_aggregate_only_delta.json": "aggregate_only_delta_bench.json",
    f"bench_{tag}_aggregate_tool.json": "aggregate_only_paths.json", f"ab_calls_{tag}.txt": "final_call_variants.txt",
    f"bench_{tag}_mixed_book_aggregate_only.json": "mixed_book_aggregate_only_bench.json",
    f"bench_{tag}_aggregate_only_legs.json": "aggregate_only_legs_bench.json",
}
for src, dst in copies.items():
    if os.path.exists(f"{G}/{src}"):
        lines = [l for l in open(f"{G}/{src}") if "amdgpu.ids" not in l]
        open(f"{P}/{rnd}_{dst}", "w").writelines(lines)
    else:
        print("missing", src)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
for path, name in ((f"prof_{tag}_lag", "payment_lag"), (f"prof_{tag}_xccy", "xccy"), (f"prof_{tag}_wide", "wide")):
    try:
        shutil.copy(newest(f"{G}/{path}/trace/*/*kernel_stats.csv"), f"{P}/{rnd}_{name}_kernel_stats.csv")
    except ValueError:
        print("missing", path)
# HBM traffic of the payment-lag bench (FETCH_SIZE / WRITE_SIZE passes; read side doubled, see tools/profile_summary.py)
try:
    out = {"tag": tag, "workload": "tools/bench_long_legs.py 200000 lag"}
    for name in ("fetch", "write"):
        vals = {}
        for r in csv.DictReader(open(newest(f"{G}/prof_{tag}_lag/{name}/*/*counter_collection.csv"))):
            if "price_" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in vals.items():
            out[k + "_KiB_per_launch"] = sum(v) / len(v)
    rd, wr = 2.0 * out.get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024, out.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024
    out.update(read_bytes_corrected=rd, write_bytes=wr, hbm_bytes_per_launch=rd + wr,
               note="read = 2 x FETCH_SIZE x 1024 (gfx950 half-count correction), write = WRITE_SIZE x 1024")
    json.dump(out, open(f"{P}/{rnd}_payment_lag_traffic.json", "w"), indent=1)
except ValueError:
    print("missing lag traffic passes")
# ... and of the wide route (40 pillars, 100 000 trades, every output stored)
try:
    out = {"tag": tag, "workload": "ABLATE_ONLY=stored tools/ablate_wide.py 40 100000 offgrid",
           "algorithmic_bytes_per_launch": 100000 * (8 * (1 + 40 + 40 * 40) + 16 * 15.5 + 32 * 15.5 + 40)}
    for name in ("fetch", "write"):
        vals = {}
        for r in csv.DictReader(open(newest(f"{G}/prof_{tag}_wide/{name}/*/*counter_collection.csv"))):
            if "price_" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in vals.items():
            out[k + "_KiB_per_launch"] = sum(v) / len(v)
    rd, wr = 2.0 * out.get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024, out.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024
    out.update(read_bytes_corrected=rd, write_bytes=wr, hbm_bytes_per_launch=rd + wr,
               note="read = 2 x FETCH_SIZE x 1024 (gfx950 half-count correction), write = WRITE_SIZE x 1024")
    json.dump(out, open(f"{P}/{rnd}_wide_traffic.json", "w"), indent=1)
except ValueError:
    print("missing wide traffic passes")
subprocess.check_call([sys.executable, "tools/profile_summary.py", tag, rnd])
for a, b in ((f"{P}/{rnd}_{tag}_kernel_stats.csv", f"{P}/{rnd}_final_kernel_stats.csv"), (f"{P}/{rnd}_{tag}_traffic.json", f"{P}/{rnd}_final_traffic.json")):
    os.replace(a, b)
