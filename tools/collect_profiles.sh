#!/bin/bash
# Copies the judged summaries of tools/prof_all.sh from gpurun_out/ (scratch) into profiles/ (tracked).  usage: tools/collect_profiles.sh r2
set -e
R=${1:-r4}
cd "$(dirname "$0")/.."
for m in bf16 bf16_serial f16 f32; do
  f=$(ls -t gpurun_out/${R}_ks_${m}/*/*_kernel_stats.csv | head -1)
  cp "$f" profiles/${R}_bench_${m}_kernel_stats.csv
done
for l in bf16 f16 f32 plugin cfg3_f16 graph; do cp gpurun_out/${R}_bench_${l}.json profiles/${R}_bench_${l}_line.json; done
python3 tools/per_layer.py gpurun_out/${R}_ks_bf16 > profiles/${R}_bench_bf16_per_layer.txt
python3 tools/per_layer.py gpurun_out/${R}_ks_bf16_serial > profiles/${R}_bench_bf16_serial_per_layer.txt
python3 tools/pmc_summary.py gpurun_out bf16 > profiles/${R}_pmc_bf16.json
python3 tools/timeline.py gpurun_out/${R}_ks_bf16 > profiles/${R}_bench_bf16_timeline.txt
echo "collected into profiles/ (csrc_sha $(python3 -c 'import bench; print(bench.csrc_sha())'))"
