#!/bin/bash
# Run a command on the GPU box through gpurun, waiting for a free slot (exit code 3 = no box/slot, nothing charged).
# usage: tools/gpu.sh TIMEOUT 'command'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
