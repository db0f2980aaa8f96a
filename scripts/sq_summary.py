"""profiles/rNN_sq_summary.md from the two SQ-counter summaries (scripts/dev/dev_profile_rNN.sh): `python scripts/sq_summary.py r03`."""
import csv, collections, sys
RN = sys.argv[1] if len(sys.argv) > 1 else "r02"
print(f"# SQ counters of the guidance step, round {RN[1:].lstrip('0')} (rocprofv3 --pmc, two passes, `scripts/profile_{RN}.sh` (round 3: `scripts/dev/dev_profile_r03.sh`))\n")
print("`python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras --no-graph [--images-per-gpu 8 --streams 1]`; means per launch.")
print("dur = GRBM_GUI_ACTIVE / 8 XCDs at 2.4 GHz (inflated by the counter collection: 1.5-2x the un-profiled kernel time);")
print("resident = SQ_WAVE_CYCLES x 4 / cycles (average waves in flight on the chip); wave life = SQ_WAVE_CYCLES x 4 / SQ_WAVES;")
print("wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of wave time waiting for memory / LDS / barriers); active = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES;")
print("VALU util = SQ_ACTIVE_INST_VALU x 4 / (cycles x 1024 SIMDs): the share of the chip's VALU issue slots in use during the launch.\n")
for t, path in (("one image (configs[1])", f"profiles/{RN}_rocprofv3_sq_counters_b1.csv"), ("8 images on one stream", f"profiles/{RN}_rocprofv3_sq_counters_b8_1stream.csv")):
    d = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        d[r["kernel"]][r["counter"]] = float(r["mean_per_launch"])
    print(f"## {t}\n\n| kernel | dur us | waves | resident | wave life us | wait | active | VALU util | VALU inst | SALU inst | LDS inst |\n|---|---|---|---|---|---|---|---|---|---|---|")
    for k in ("k_xform", "k_stage2", "k_resolve", "k_loss", "k_pix_bwd", "k_vert_bwd"):
        c = d[k]; cyc = c["GRBM_GUI_ACTIVE"] / 8
        print(f"| `{k}` | {cyc/2400:.1f} | {c['SQ_WAVES']:.0f} | {c['SQ_WAVE_CYCLES']*4/cyc:.0f} | {c['SQ_WAVE_CYCLES']*4/c['SQ_WAVES']/2400:.2f} | "
              f"{c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.2f} | {c['SQ_ACTIVE_INST_ANY']/c['SQ_WAVE_CYCLES']:.2f} | {c['SQ_ACTIVE_INST_VALU']*4/(cyc*1024):.3f} | "
              f"{c['SQ_INSTS_VALU']:.0f} | {c['SQ_INSTS_SALU']:.0f} | {c['SQ_INSTS_LDS']:.0f} |")
    print()
print("Reading: at one image every kernel spends most of its wave time waiting and uses a few per cent of the VALU issue slots -- the step is")
print("bound by chains of dependent memory round trips, not by arithmetic or bytes.  `k_stage2` holds the two roles SURVEY 8(d) calls ALU bound")
print("(edge-function tests of the scatter rasteriser, the ray-parity inside test) plus the nearest-neighbour role: it is the only kernel whose VALU")
print("utilisation is not negligible, and it roughly doubles with eight images in flight.")
