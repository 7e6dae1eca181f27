#!/usr/bin/env bash
# gpurun, retried while the pod's GPU slots are busy (exit code 3: nothing charged)
#   tools/gpurun_retry.sh <timeout seconds> '<command>'
t=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
