#!/bin/bash
# round 3, final-build check: the whole GPU suite, then the profile set (tools/gpu_r3g.sh)
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3h_tests.log 2>&1; tail -3 gpurun_out/r3h_tests.log
bash tools/gpu_r3g.sh
