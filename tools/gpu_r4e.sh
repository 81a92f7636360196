#!/bin/bash
# round 4: what the machine kernels wait for -- texture addresser / vector L1 / instruction cache counters
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4e
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA|TCP|SQ|SQC|TD)_[A-Za-z0-9_]+" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/r4e/counters.txt)
wc -l gpurun_out/r4e/counters.txt
B="--steps 1 --warmup 1 --no-cpu-baseline --no-gather --no-regions"
for PMC in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC" \
           "TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ SQC_TC_INST_REQ"; do
  echo "## $PMC"
  timeout 300 bash tools/pmc_once.sh r4e "$PMC" $B 2>&1 | grep -E "k_machine|rror|nvalid" | cut -c1-900
done
