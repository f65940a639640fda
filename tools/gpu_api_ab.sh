#!/bin/bash
# host-array rollout rate (mujoco_amd.rollout.rollout -> mjhip_rollout) for several chunk lengths of the overlapped copies
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-api_ab}
mkdir -p "$OUT"
for c in 0 50 84 125; do
  MJHIP_ROLLOUT_CHUNK=$c timeout 600 python tools/api_rate.py 4096 250 2>&1 | grep "^chunk" | tee -a "$OUT/api_rate.txt"
done
