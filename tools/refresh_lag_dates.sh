#!/bin/bash
# Evidence for the opt-in date-record payment-lag kernel (kernels_lag.hip) next to the default variant: bench lines, SQ
# counters per trade, phase stamps.  Run on the GPU box; output under gpurun_out/.
cd /root/repo
python tools/bench_long_legs.py 200000 lag > gpurun_out/lagcmp_default.json 2>/dev/null
python tools/bench_long_legs.py 100000 longlag >> gpurun_out/lagcmp_default.json 2>/dev/null
ADR_LAG_KERNEL=dates python tools/bench_long_legs.py 200000 lag > gpurun_out/lagcmp_dates.json 2>/dev/null
ADR_LAG_KERNEL=dates python tools/bench_long_legs.py 100000 longlag >> gpurun_out/lagcmp_dates.json 2>/dev/null
export ADR_LAG_KERNEL=dates
bash tools/pmc_lag.sh && python tools/pmc_summary.py 200000 > gpurun_out/pmc_lag_dates.txt
if [ -f variants_stamps.so ]; then
  ADRATES_HIP_LIB=/root/repo/variants_stamps.so python tools/stamps_lag.py > gpurun_out/stamps_lag_dates.txt 2>/dev/null
  unset ADR_LAG_KERNEL
  ADRATES_HIP_LIB=/root/repo/variants_stamps.so python tools/stamps_lag.py > gpurun_out/stamps_lag_default.txt 2>/dev/null
fi
echo lag-dates-done
