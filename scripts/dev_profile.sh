set -x
cd /root/repo
mkdir -p gpurun_out
python bench.py > gpurun_out/r01_bench_b1.json 2> gpurun_out/b1.err
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --images-per-gpu 8 > gpurun_out/r01_bench_b8.json 2>/dev/null
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --images-per-gpu 32 > gpurun_out/r01_bench_b32.json 2>/dev/null
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --obj 40k > gpurun_out/r01_bench_40k.json 2>/dev/null
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2>&1
cd $R
find gpurun_out/prof_kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats.csv
python scripts/summarize_pmc.py $(find gpurun_out/prof_fetch gpurun_out/prof_write -name "*counter_collection.csv") > gpurun_out/pmc_summary.csv
head -30 gpurun_out/kernel_stats.csv
cat gpurun_out/pmc_summary.csv
tail -c 1500 gpurun_out/r01_bench_b1.json
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --images-per-gpu 64 > gpurun_out/r01_bench_b64.json 2>/dev/null
bash scripts/dev_profile_b16.sh > gpurun_out/profile_b16.log 2>&1
tail -c 400 gpurun_out/r01_bench_b8.json; tail -c 400 gpurun_out/r01_bench_b32.json; tail -c 400 gpurun_out/r01_bench_b64.json; tail -c 400 gpurun_out/r01_bench_40k.json
head -8 gpurun_out/kernel_stats_b16.csv
