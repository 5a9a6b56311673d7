#!/bin/bash
# gpurun with retries ONLY while no box / slot is free (exit code 3: nothing ran, nothing charged).  usage: grun.sh TIMEOUT 'command'
t=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
