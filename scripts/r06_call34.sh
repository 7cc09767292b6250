#!/bin/bash
# flakiness check: the GPU suite twice more on one box
cd "${GRAFT_REPO_ROOT:-.}"
export TPA_NO_AUTOBUILD=1
for i in 1 2; do timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3; done
timeout 600 python scripts/eigh_fuzz.py 200 23 2>&1 | grep -v amdgpu.ids | grep "BAD\|trials" | head -20
