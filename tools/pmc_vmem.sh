#!/bin/bash
# round 4: vector-memory latency / busy counters of the machine kernels, few counters per pass
cd "$GRAFT_REPO_ROOT" || exit 1
B="--steps 1 --warmup 1 --no-cpu-baseline --no-gather --no-regions"
for PMC in "TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"; do
  echo "## $PMC"
  timeout 300 bash tools/pmc_once.sh vmem "$PMC" $B 2>&1 | grep -E "k_machine|k_bin_probe|rror|nvalid" | cut -c1-400
done
