cd /tmp && export TMPDIR=/tmp
for L in tri hub; do
export ADR_LAYOUT=$L
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /root/repo/gpurun_out/pmc_${L}_a -- python3 /root/repo/bench.py --steps 2 --warmup 1 --cpu-baseline-seconds 0 > /root/repo/gpurun_out/pmc_${L}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d /root/repo/gpurun_out/pmc_${L}_b -- python3 /root/repo/bench.py --steps 2 --warmup 1 --cpu-baseline-seconds 0 > /root/repo/gpurun_out/pmc_${L}_b.log 2>&1
done
cd /root/repo && python tools/pmc_summary.py
