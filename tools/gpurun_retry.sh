#!/bin/bash
# (container side) gpurun with retries while no GPU slot / box is free: tools/gpurun_retry.sh <log> <timeout> '<command>'
LOG=$1; TMO=$2; CMD=$3
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$TMO" -- "$CMD" > "$LOG" 2>&1
  rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" "$LOG"; then exit $rc; fi
  sleep 45
done
exit 3
