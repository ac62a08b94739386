#!/bin/bash
# Round artefacts on the GPU box: un-profiled bench lines (bf16 with the CPU baselines and mIoU-vs-ref, fp16, fp32, the
# plugin path), kernel-trace statistics of the same command (default two-stream mode and one stream), and the three
# separate --pmc passes.  Outputs under gpurun_out/; summarise with tools/pmc_summary.py, tools/per_layer.py, then copy what
# is to be judged into profiles/ (tools/collect_profiles.sh).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
set -e
mkdir -p gpurun_out
R=${1:-r4}
python3 bench.py --steps 30 --warmup 5 > gpurun_out/${R}_bench_bf16.json 2> gpurun_out/${R}_bench_bf16.err
tail -c 2500 gpurun_out/${R}_bench_bf16.json
python3 bench.py --steps 30 --warmup 5 --dtype f16 --no-cpu-baseline --no-eval > gpurun_out/${R}_bench_f16.json 2> gpurun_out/${R}_bench_f16.err
python3 bench.py --steps 10 --warmup 3 --dtype f32 --no-cpu-baseline --no-miou --no-loader --no-eval > gpurun_out/${R}_bench_f32.json 2> gpurun_out/${R}_bench_f32.err
python3 bench.py --steps 30 --warmup 5 --path plugin --no-cpu-baseline --no-miou --no-loader > gpurun_out/${R}_bench_plugin.json 2> gpurun_out/${R}_bench_plugin.err
python3 bench.py --steps 20 --warmup 5 --size 512 --channels 9 --batch 16 --dtype f16 --no-cpu-baseline --no-miou --no-loader --no-eval > gpurun_out/${R}_bench_cfg3_f16.json 2> gpurun_out/${R}_bench_cfg3_f16.err
python3 bench.py --steps 30 --warmup 5 --graph 1 --no-cpu-baseline --no-miou --no-loader --no-eval > gpurun_out/${R}_bench_graph.json 2> gpurun_out/${R}_bench_graph.err
echo lines done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_ks_bf16 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass --no-miou --no-loader --no-eval > gpurun_out/${R}_ks_bf16.log 2>&1
export FU_NO_SIDE_STREAM=1   # one stream: per-kernel durations without the two backward chains sharing the GPU
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_ks_bf16_serial -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass --no-miou --no-loader --no-eval > gpurun_out/${R}_ks_bf16_serial.log 2>&1
unset FU_NO_SIDE_STREAM
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_ks_f16 -- python3 bench.py --steps 10 --warmup 3 --dtype f16 --no-cpu-baseline --no-serial-pass --no-miou --no-loader --no-eval > gpurun_out/${R}_ks_f16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_ks_f32 -- python3 bench.py --steps 5 --warmup 2 --dtype f32 --no-cpu-baseline --no-serial-pass --no-miou --no-loader --no-eval > gpurun_out/${R}_ks_f32.log 2>&1
echo traces done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-serial-pass --no-miou --no-loader --no-eval > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-serial-pass --no-miou --no-loader --no-eval > gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-serial-pass --no-miou --no-loader --no-eval > gpurun_out/pmc_sq.log 2>&1
echo done
