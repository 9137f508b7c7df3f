#!/bin/bash
# tools/gpurun_retry.sh TIMEOUT 'command' : gpurun, retried every three minutes while the pool answers "busy" (nothing charged)
t=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  if grep -q '"status": "transient"' gpurun_out/.last_call.json 2>/dev/null; then sleep 180; continue; fi
  exit $rc
done
exit 3
