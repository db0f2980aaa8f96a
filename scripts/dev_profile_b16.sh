# per-kernel durations in the batch regime: 16 images on ONE stream (kernels of different streams would overlap)
set -x
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --images-per-gpu 16 --streams 1 2>/dev/null | cut -c1-260
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt16 -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --images-per-gpu 16 --streams 1 > /dev/null 2>&1
cd $R
find gpurun_out/prof_kt16 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_b16.csv
head -12 gpurun_out/kernel_stats_b16.csv | cut -c1-200
