#!/bin/bash
# usage: tools/exp/gpurun_retry.sh <timeout> '<command>' <logfile>
# Calls gpurun; when NO box or slot is free (exit code 3: nothing ran, nothing charged) waits and asks again, up to 12 times.
# Any other outcome (the command ran, was refused, failed) is returned as it is: a GPU command is never re-run by this script.
t=$1; cmd=$2; log=$3
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$cmd" > "$log" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
