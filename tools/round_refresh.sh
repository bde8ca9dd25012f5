#!/bin/bash
# Run on the GPU box: the round's evidence in one call - GPU parity tests, the bench line (with the CPU baseline),
# the other reported configurations, ablations, rocprofv3 kernel stats + HBM traffic passes and the PMC counters.
# Everything lands in gpurun_out/; tools/profile_summary.py TAG rNN then copies the summaries into profiles/.
TAG=${1:-final}
cd /root/repo
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_$TAG.log 2>&1 || { tail -20 gpurun_out/gpu_tests_$TAG.log; exit 1; }
tail -1 gpurun_out/gpu_tests_$TAG.log
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || exit 1
python bench.py --kind ongrid --cpu-baseline-seconds 0 > gpurun_out/bench_${TAG}_ongrid.json 2>/dev/null || exit 1
python bench.py --interp FLAT_FWD_RATES --cpu-baseline-seconds 0 > gpurun_out/bench_${TAG}_flatfwd.json 2>/dev/null || exit 1
python bench.py --interp LINEAR_FWD_RATES --trades 200000 --cpu-baseline-seconds 0 > gpurun_out/bench_${TAG}_linfwd_200k.json 2>/dev/null || exit 1
python bench.py --trades 100000 --requests value,delta --cpu-baseline-seconds 0 > gpurun_out/bench_${TAG}_config2_delta_100k.json 2>/dev/null || exit 1
python bench.py --trades 1000000 --requests value,delta --cpu-baseline-seconds 0 > gpurun_out/bench_${TAG}_delta_1m.json 2>/dev/null || exit 1
python bench.py --trades 1000000 --requests value --cpu-baseline-seconds 0 > gpurun_out/bench_${TAG}_value_1m.json 2>/dev/null || exit 1
python tools/ablate.py > gpurun_out/ablate_$TAG.log 2>&1 || exit 1
# round 4: the aggregate-only mode (Portfolio.compute's ladder alone) as its own bench lines, interleaved call timing
python bench.py --aggregate-only --cpu-baseline-seconds 0 > gpurun_out/bench_${TAG}_aggregate_only.json 2>/dev/null || exit 1
python bench.py --aggregate-only --requests value,delta --cpu-baseline-seconds 0 > gpurun_out/bench_${TAG}_aggregate_only_delta.json 2>/dev/null || exit 1
python tools/bench_aggregate.py > gpurun_out/bench_${TAG}_aggregate_tool.json 2>/dev/null || exit 1
python tools/ab_calls.py > gpurun_out/ab_calls_$TAG.txt 2>/dev/null || exit 1
python tools/bench_long_legs.py > gpurun_out/bench_${TAG}_long_legs.json 2>/dev/null || exit 1
python tools/bench_curve_build.py > gpurun_out/bench_${TAG}_curve_build.json 2>/dev/null || exit 1
python tools/bench_long_legs.py 200000 lag > gpurun_out/bench_${TAG}_payment_lag.json 2>/dev/null || exit 1
python tools/bench_xccy.py 100000 3 > gpurun_out/bench_${TAG}_xccy.json 2>/dev/null || exit 1
python tools/bench_xccy.py 100000 7 >> gpurun_out/bench_${TAG}_xccy.json 2>/dev/null || exit 1
python bench.py --xccy-swaps 100000 --steps 10 --cpu-baseline-seconds 0 > gpurun_out/bench_${TAG}_mixed_book.json 2>/dev/null || exit 1
python bench.py --xccy-swaps 100000 --aggregate-only --steps 10 --cpu-baseline-seconds 0 > gpurun_out/bench_${TAG}_mixed_book_aggregate_only.json 2>/dev/null || exit 1
for mode in lag longlag long; do python tools/bench_long_legs.py 200000 $mode 7 aggonly 2>/dev/null; done > gpurun_out/bench_${TAG}_aggregate_only_legs.json || exit 1
python bench.py --interp LINEAR_FWD_RATES --cpu-baseline-seconds 0 > gpurun_out/bench_${TAG}_linfwd.json 2>/dev/null || exit 1
bash tools/profile.sh $TAG || exit 1
bash tools/pmc.sh && python tools/pmc_summary.py > gpurun_out/pmc_$TAG.txt || exit 1
python tools/pmc_fp64_summary.py ${RND:-r04} > gpurun_out/pmc_${TAG}_fp64.json || exit 1
# round 4: the PV + delta pass (lite kernel) at both sizes: kernel stats + FETCH / WRITE passes
BENCH_ARGS="--trades 100000 --requests value,delta" bash tools/profile.sh ${TAG}_delta100k || exit 1
BENCH_ARGS="--trades 1000000 --requests value,delta" bash tools/profile.sh ${TAG}_delta1m || exit 1
bash tools/profile_paths.sh $TAG || exit 1
bash tools/pmc_lag.sh && python tools/pmc_summary.py 200000 > gpurun_out/pmc_${TAG}_lag.txt || exit 1
if [ -f variants_stamps.so ]; then
  ADRATES_HIP_LIB=/root/repo/variants_stamps.so python tools/stamps.py > gpurun_out/stamps_$TAG.txt 2>/dev/null || exit 1
  ADRATES_HIP_LIB=/root/repo/variants_stamps.so python tools/stamps_lag.py > gpurun_out/stamps_${TAG}_lag.txt 2>/dev/null || exit 1
fi
echo refresh-done
