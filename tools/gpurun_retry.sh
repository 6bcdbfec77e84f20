#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout> '<command>'   -- gpurun, retried while the pod's GPU slots are busy (nothing is charged for those)
git -C /root/repo rev-parse HEAD > /root/repo/.git_head 2>/dev/null
for i in $(seq 1 30); do
  out=$(/usr/local/graft/bin/gpurun --timeout "$1" -- "$2" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out"
  # gpurun's own verdict: "status=ok rc=N" -> N; refused / no box / killed -> non-zero
  rc=$(echo "$out" | sed -n 's/.*status=ok rc=\([0-9]*\).*/\1/p' | tail -1)
  [ -n "$rc" ] && exit "$rc"
  exit 1
done
echo "$out"; exit 3
