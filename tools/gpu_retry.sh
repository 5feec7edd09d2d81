#!/bin/bash
# usage: tools/gpu_retry.sh <timeout_s> <script-in-repo>   -- retries while the pool is busy (exit 3); log in gpurun_out/retry.log
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- bash "$@" > /root/repo/gpurun_out/retry.log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
