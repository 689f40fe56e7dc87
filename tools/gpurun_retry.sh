#!/bin/bash
# tools/gpurun_retry.sh TIMEOUT_S 'command' : gpurun, retried while the pool reports no free slot (exit code 3: nothing charged)
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
