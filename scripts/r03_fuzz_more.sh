#!/bin/bash
# a longer round of the differential fuzzers with fresh seeds (GPU box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 240 python scripts/soak_results.py 30 4000 2>&1 | grep -v amdgpu.ids | tail -4 || exit 1
timeout -k 10 400 python scripts/fuzz_automata.py 120 5000 2>&1 | grep -v amdgpu.ids | tail -3 || exit 1
timeout -k 10 300 python scripts/fuzz_automata.py 40 6000 wide 2>&1 | grep -v amdgpu.ids | tail -3 || exit 1
timeout -k 10 400 python scripts/fuzz_automata.py 8 7000 long 2>&1 | grep -v amdgpu.ids | tail -3 || exit 1
timeout -k 10 500 python scripts/soak.py 8 900 2>&1 | grep -v amdgpu.ids | tail -4 || exit 1
