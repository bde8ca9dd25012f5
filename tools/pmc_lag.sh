#!/bin/bash
# PMC counters of the payment-lag variant of the fast kernel on its own workload (200 000 payment-lag OIS, PV + delta +
# gamma); two passes, counters only.  Summarise with: python tools/pmc_summary.py 200000
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/pmc_a /root/repo/gpurun_out/pmc_b
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /root/repo/gpurun_out/pmc_a -- python3 /root/repo/tools/bench_long_legs.py 200000 lag > /root/repo/gpurun_out/pmc_a.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d /root/repo/gpurun_out/pmc_b -- python3 /root/repo/tools/bench_long_legs.py 200000 lag > /root/repo/gpurun_out/pmc_b.log 2>&1
