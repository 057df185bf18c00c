#!/bin/bash
# Developer helper (build container): one gpurun call, re-submitted while the pod reports "no slot free" (exit 3: nothing ran, nothing charged).
#   tools/gpu_call.sh TAG TIMEOUT 'command'   -> gpurun_out/TAG_call.log
TAG=$1; TO=$2; CMD=$3
for try in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout $TO -- "$CMD" > gpurun_out/${TAG}_call.log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then break; fi
  sleep 90
done
exit $rc
