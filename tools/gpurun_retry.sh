#!/bin/bash
# gpurun with retries of the ACQUISITION only: exit code 3 = no box / slot free, nothing ran and
# nothing was charged.  Any other outcome (success, failure, timeout) is returned as is.
# Usage: tools/gpurun_retry.sh TIMEOUT 'command'
t=$1; shift
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 150
done
exit 3
