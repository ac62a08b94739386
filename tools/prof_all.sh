cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
set -e
python3 bench.py --steps 30 --warmup 5 > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err
tail -c 1500 gpurun_out/bench_bf16.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1b_bf16 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r1b_bf16.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sq.log 2>&1
ls gpurun_out/pmc_fetch/*/ | head
