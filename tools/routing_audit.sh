cd /tmp && export TMPDIR=/tmp
for spec in "xccy3:tools/bench_xccy.py 100000 3" "lag3:tools/bench_long_legs.py 200000 lag 3" "long7:tools/bench_long_legs.py 100000 long 7" "long3:tools/bench_long_legs.py 100000 long 3" "longlag7:tools/bench_long_legs.py 100000 longlag 7" "longlag3:tools/bench_long_legs.py 100000 longlag 3" "manypillars:tools/bench_many_pillars.py 20000" "laglinfwd7:tools/bench_long_legs.py 100000 lag 7 agg linfwd"; do
  name=${spec%%:*}; cmd=${spec#*:}
  if [ "${cmd##* }" = linfwd ]; then export ADR_BENCH_INTERP=2; else unset ADR_BENCH_INTERP; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/audit_$name -- python3 /root/repo/$cmd > /root/repo/gpurun_out/audit_$name.log 2>&1
  f=$(ls /root/repo/gpurun_out/audit_$name/*/*kernel_stats.csv | tail -1)
  echo "== $name"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "price_" in r["Name"] or "curve_df" in r["Name"]:
        print("   %-95s calls %4s avg %9.1f us" % (r["Name"].replace("adr::(anonymous namespace)::","").split("(adr::")[0][:95], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
