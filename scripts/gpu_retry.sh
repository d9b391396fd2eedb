#!/bin/bash
# Wait for a free GPU slot: waits while another call of this session is still in flight, then re-issues the SAME gpurun call
# only while it is refused with exit code 3 ("no box or slot free", nothing charged, the command never started).  Any other
# outcome -- success, failure, timeout -- is final.
# usage: scripts/gpu_retry.sh <timeout_s> '<command>'
T="$1"; shift
for i in $(seq 1 60); do
  if /usr/local/graft/bin/gpurun --status 2>/dev/null | grep -q '"in_flight": 0'; then break; fi
  sleep 20
done
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
