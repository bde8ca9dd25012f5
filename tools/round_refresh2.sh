#!/bin/bash
# Run on the GPU box after tools/round_refresh.sh TAG (a second call: one gpurun call is limited to 20 minutes): the rest of
# the round's evidence.  tools/collect_profiles.py TAG rNN picks both parts up.
TAG=${1:-final}
cd /root/repo
# round 3, second half: the wide route (33-64 pillars), payment lag under LINEAR_FWD_RATES, the routing audit
python tools/bench_many_pillars.py 2>/dev/null > gpurun_out/bench_${TAG}_many_pillars.json || exit 1
ADR_BENCH_INTERP=2 python tools/bench_long_legs.py 200000 lag 2>/dev/null > gpurun_out/bench_${TAG}_payment_lag_linfwd.json || exit 1
ADR_BENCH_INTERP=2 python tools/bench_long_legs.py 100000 longlag 2>/dev/null >> gpurun_out/bench_${TAG}_payment_lag_linfwd.json || exit 1
# (this call runs on a fresh box: the first call's file is not there to append to)
python tools/bench_long_legs.py 200000 lag 2>/dev/null > gpurun_out/bench_${TAG}_payment_lag.json || exit 1
python tools/bench_long_legs.py 100000 longlag 2>/dev/null >> gpurun_out/bench_${TAG}_payment_lag.json || exit 1
(python tools/ablate_wide.py 40 100000 offgrid; python tools/ablate_wide.py 40 100000 ongrid; python tools/ablate_wide.py 64 100000 offgrid) 2>/dev/null | grep pillars > gpurun_out/ablate_${TAG}_wide.log || exit 1
ABLATE_ONLY=stored bash tools/pmc_wide.sh 40 && python tools/pmc_summary.py 100000 > gpurun_out/pmc_${TAG}_wide.txt || exit 1
bash tools/profile_wide.sh $TAG || exit 1
bash tools/routing_audit.sh > gpurun_out/routing_audit_${TAG}.txt 2>&1 || exit 1
echo refresh2-done
