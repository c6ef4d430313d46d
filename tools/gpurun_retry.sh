#!/bin/bash
# gpurun client wrapper: retries only when no GPU slot / box was free (exit code 3: nothing ran, nothing charged).
# usage: tools/gpurun_retry.sh TIMEOUT 'command'
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 45
done
exit 3
