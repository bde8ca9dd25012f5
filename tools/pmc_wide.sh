#!/bin/bash
# PMC passes over the wide route (33-64 pillars) on the GPU box: gpurun -- 'bash tools/pmc_wide.sh [pillars]'.
# Counters only (no sys / runtime traces).  Summarise with: python tools/pmc_summary.py 100000
P=${1:-40}
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/pmc_a /root/repo/gpurun_out/pmc_b /root/repo/gpurun_out/pmc_c
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /root/repo/gpurun_out/pmc_a -- python3 /root/repo/tools/ablate_wide.py $P 100000 offgrid > /root/repo/gpurun_out/pmc_a.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d /root/repo/gpurun_out/pmc_b -- python3 /root/repo/tools/ablate_wide.py $P 100000 offgrid > /root/repo/gpurun_out/pmc_b.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_FLAT --output-format csv -d /root/repo/gpurun_out/pmc_c -- python3 /root/repo/tools/ablate_wide.py $P 100000 offgrid > /root/repo/gpurun_out/pmc_c.log 2>&1
