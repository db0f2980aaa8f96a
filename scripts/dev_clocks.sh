cd ${GRAFT_REPO_ROOT:-/root/repo}
rocm-smi --showperflevel 2>&1 | tail -4
rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | head -6
(timeout 60 python bench.py --no-cpu-baseline --no-extras --steps 600000 --warmup 100 > /tmp/b.json 2>/dev/null &) 
sleep 9
for i in 1 2 3 4; do rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk" | head -3; rocm-smi --showpower 2>&1 | grep -i "power" | head -2; sleep 2; done
wait
sleep 20
tail -c 300 /tmp/b.json
