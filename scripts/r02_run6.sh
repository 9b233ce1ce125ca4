#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_exact_and_replay.py -x -q -m gpu > gpurun_out/r02/t_bits2.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r02/t_bits2.log
timeout -k 10 200 python scripts/sweep2.py 128 1,3 2>&1 | tail -1
for v in B K1; do echo "== lib$v (compaction skipped)"; DATOK_EXP_SKIP=8 DATOK_GPU_LIB=$PWD/ab/lib$v.so timeout -k 10 200 python scripts/sweep2.py 128 1,3 2>&1 | tail -1; done
cd /tmp
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $ctrs | cut -c1-12 | tr ' ' '_')
  timeout -k 10 180 rocprofv3 --kernel-trace --pmc $ctrs -d $GRAFT_REPO_ROOT/gpurun_out/r02/pmc_$tag -o pmc -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 12 --warmup 2 --no-cpu-baseline --parity-docs 0 > /dev/null 2>&1; echo "pmc $tag rc=$?"
done
