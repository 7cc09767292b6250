#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TPA_NO_AUTOBUILD=1
timeout 900 python scripts/eigh_fuzz.py 150 7 2>&1 | grep -v amdgpu.ids | tail -60
