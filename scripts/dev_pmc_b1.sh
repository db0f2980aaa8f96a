# PMC passes over the one-image step (no graph): wave occupancy and wait breakdown per kernel
R=/root/repo
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
CMD="python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-graph"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc1_a -- $CMD > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_REQ_sum TCC_ATOMIC_sum SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc1_b -- $CMD > /dev/null 2>&1
cd $R
python scripts/summarize_pmc.py $(find gpurun_out/pmc1_a gpurun_out/pmc1_b -name "*counter_collection.csv") > gpurun_out/pmc1_summary.csv
python - <<'PY'
import csv, collections
d = collections.defaultdict(dict)
for r in csv.DictReader(open("gpurun_out/pmc1_summary.csv")):
    d[r["kernel"]][r["counter"]] = float(r["mean_per_launch"])
for k in ("k_xform", "k_stage2", "k_resolve", "k_loss", "k_pix_bwd", "k_vert_bwd"):
    c = d[k]
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print(f"{k:11s} dur {cyc/2400:6.1f} us  waves {c['SQ_WAVES']:7.0f}  resident waves {c['SQ_WAVE_CYCLES']*4/cyc:7.0f}  wave life {c['SQ_WAVE_CYCLES']*4/c['SQ_WAVES']/2400:5.2f} us  "
          f"wait {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.2f} stall {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.2f} active {c['SQ_ACTIVE_INST_ANY']/c['SQ_WAVE_CYCLES']:.2f}  "
          f"L2 req {c['TCC_REQ_sum']:8.0f} atomics {c['TCC_ATOMIC_sum']:7.0f}  valu {c['SQ_INSTS_VALU']:8.0f} salu {c['SQ_INSTS_SALU']:8.0f}")
PY
