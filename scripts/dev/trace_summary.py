"""Per-kernel table of ONE iteration out of a rocprofv3 --kernel-trace csv: python scripts/dev/trace_summary.py <kernel_trace.csv> [marker kernel]
An iteration = from the last-but-one launch of the marker kernel (default k_vae_rowstat: once per foho_vae_fwd) to the last one."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_vae_rowstat"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
seg = rows[idx[-2]:idx[-1]]


def short(n):
    m = re.search(r"k_geo_gemm(8p|256|_d4|_pc)?ILi(\d+)E", n)
    if m:
        return f"gemm{m.group(1) or '128'}<{m.group(2)}>"
    m = re.search(r"(k_[a-z_0-9]+)", n)
    return m.group(1) if m else n[:48]


agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = (short(r["Kernel_Name"]), r["Grid_Size_X"])
    agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
wall = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
print(f"{len(seg)} launches, kernel time {tot / 1e6:.3f} ms, wall {wall / 1e6:.3f} ms")
for k, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{d / 1e3:9.1f} us {c:4d} calls {d / c / 1e3:8.1f} us avg  {k[0]} grid {k[1]}")
