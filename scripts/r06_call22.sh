#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06
mkdir -p $O
export TPA_NO_AUTOBUILD=1
timeout 600 python scripts/host_segments.py 100 2048 2 > $O/host_segments_2048.txt 2>&1
tail -25 $O/host_segments_2048.txt
CHI=512 timeout 600 python scripts/host_profile.py 8 3 > $O/host_profile_512.txt 2>&1
head -90 $O/host_profile_512.txt | cut -c1-180
