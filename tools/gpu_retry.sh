#!/bin/bash
# gpurun with patience: exit code 3 = no GPU slot free right now, nothing charged, nothing run -> wait and ask again.
# (Only that case is retried; a command that ran is never run again by this script.)
#   tools/gpu_retry.sh <timeout-seconds> '<command>'
T=$1; shift
for attempt in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
