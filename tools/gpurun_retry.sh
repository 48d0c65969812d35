#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (status "transient": nothing is charged):  tools/gpurun_retry.sh <timeout> '<command>' [log]
T=$1; CMD=$2; LOG=${3:-/tmp/gpurun_retry.log}
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$CMD" > $LOG 2>&1
  grep -q "status=transient" $LOG || break
  sleep 90
done
