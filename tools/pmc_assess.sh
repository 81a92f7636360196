#!/bin/bash
# round 5: is k_assess (-s 1, 250 Mbp, 512 MiB filter) gather-bound?  Vector-memory counters of the kernel, two per pass
cd "$GRAFT_REPO_ROOT" || exit 1
B="--snv --bases 250e6 --contig-len 100000 --filter-bytes 536870912 --steps 1 --warmup 1 --no-cpu-baseline --no-gather --no-regions"
for PMC in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  echo "## $PMC"
  timeout 300 bash tools/pmc_once.sh assess "$PMC" $B 2>&1 | grep -E "k_assess|rror|nvalid" | cut -c1-400
done
