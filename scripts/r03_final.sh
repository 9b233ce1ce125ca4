#!/bin/bash
# end-of-round measurements: profiles (bench line, rocprofv3 stats, traffic), SQ / TCP / L2 counters of the
# kernels with one batch alone, the other configurations and corpora
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash scripts/round_profiles.sh r03 > gpurun_out/r03/round_profiles.log 2>&1 || { tail -5 gpurun_out/r03/round_profiles.log; exit 1; }
bash scripts/pmc_groups.sh "k_spec_both|k_symbolize|k_compact_plain" --streams 1 > gpurun_out/r03/pmc_sq.txt 2>&1
bash scripts/pmc_tcp.sh > gpurun_out/r03/pmc_tcp.txt 2>&1
timeout -k 10 300 python scripts/configs.py > gpurun_out/r03/configs.txt 2>&1
timeout -k 10 300 python scripts/robust.py > gpurun_out/r03/robust.txt 2>&1
timeout -k 10 200 python scripts/tiny_docs.py > gpurun_out/r03/tiny.txt 2>&1
timeout -k 10 200 python scripts/big_stages.py 32 > gpurun_out/r03/big_stages.txt 2>&1
timeout -k 10 200 python scripts/stages3.py 65536 >> gpurun_out/r03/big_stages.txt 2>&1
for f in configs robust tiny big_stages; do tail -n 4 gpurun_out/r03/$f.txt; done
bash scripts/profile_config4.sh r03 > gpurun_out/r03/config4.log 2>&1; tail -4 gpurun_out/r03/config4.log
timeout -k 10 120 python scripts/transduce_latency.py > gpurun_out/r03/transduce_latency.txt 2>&1; tail -5 gpurun_out/r03/transduce_latency.txt
