#!/bin/bash
# profiles/gr.sh TIMEOUT 'command' -- gpurun with retries while the pod's GPU slots are busy (exit code 3: nothing charged)
T=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /tmp/gr_last.log 2>&1; rc=$?
  if grep -q "status=transient" /tmp/gr_last.log; then sleep 45; continue; fi
  cat /tmp/gr_last.log; exit $rc
done
cat /tmp/gr_last.log; exit 3
