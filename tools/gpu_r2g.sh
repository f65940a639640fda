# wide PGS A/B in the testspeed regime + tests + default bench
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python tools/regime_stats.py 1000 2>&1 | grep "launch\|65, "
MJHIP_PGS_WIDE=0 python tools/regime_stats.py 1000 2>&1 | grep "launch"
bash tools/gpu_quick.sh
