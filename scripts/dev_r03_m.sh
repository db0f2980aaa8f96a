cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(o['value']), o['ms_per_step'], o['ms_per_step_min_max'], o['repeats'], round(o['batched']['value']), o['batched'].get('step_traffic_frac'))"; done
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(o['value']), o['ms_per_step'], o['repeats'])"
