#!/bin/bash
# Round 4: memory-path counters of the store streams - the bench kernel's, a fill's, the microbenchmark's (separate PMC
# passes, counters only).  Which stall separates 5.3 TB/s (persistent waves) from 6.8 TB/s (one store per wave)?
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/pmc_stores
rm -rf $O; mkdir -p $O
P1="SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"
P2="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
P3="TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCC_BUSY_sum TCC_WRITE_sum TCC_EA0_WRREQ_LEVEL_sum"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/micro$i -- /root/repo/tools/exp_store_stream > $O/micro$i.log 2>&1
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/bench$i -- python3 /root/repo/tools/ab_calls.py 1000000 1 > $O/bench$i.log 2>&1
done
cd /root/repo
python3 - <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_stores/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("adr::(anonymous namespace)::", "").replace("void ", "")[:70]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if not any(s in k for s in ("price_fast", "fill_kernel", "stream_kernel", "march", "Fill", "range_kernel", "balance")):
        continue
    print(k, "launches", max(len(v) for v in agg[k].values()))
    for c, v in sorted(agg[k].items()):
        print(f"    {c:44s} {sum(v) / len(v):18.0f}")
PY
