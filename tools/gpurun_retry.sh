#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout> '<command>'   -- gpurun, retried while the pod's GPU slots are busy (nothing is charged for those)
git -C /root/repo rev-parse HEAD > /root/repo/.git_head 2>/dev/null
for i in $(seq 1 30); do
  out=$(/usr/local/graft/bin/gpurun --timeout "$1" -- "$2" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out"; exit 0
done
echo "$out"; exit 3
