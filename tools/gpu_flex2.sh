#!/bin/bash
# stage profile of the flex configuration (needs tools/variants/libmjhip_prof.so)
MODEL=flex NENV=256 K=50 W=400 MJHIP_LIB=$PWD/tools/variants/libmjhip_prof.so timeout 600 python tools/stage_profile.py > gpurun_out/stage_profile_flex.txt 2>&1
head -40 gpurun_out/stage_profile_flex.txt
