"""Timeline of ONE iteration out of a rocprofv3 --kernel-trace csv, grouped into consecutive runs of the same kernel family:
python scripts/dev/trace_timeline.py <kernel_trace.csv> <marker kernel> [max rows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2]
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
seg = rows[idx[-2]:idx[-1]]
t0 = int(seg[0]["Start_Timestamp"])


def short(n):
    m = re.search(r"k_geo_gemm(8p|256|_d4|_pc)?ILi(\d+)E", n)
    if m:
        return f"gemm{m.group(1) or '128'}<{m.group(2)}>"
    m = re.search(r"(k_[a-z_0-9]+)", n)
    if m:
        return m.group(1)
    n = re.sub(r"^void ", "", n)
    m = re.search(r"(Cijk_\w{0,20}|[a-zA-Z_]+::[a-zA-Z_:() ]+<[^>]{0,90}|[a-z_A-Z0-9]+)", n)
    return (m.group(1) if m else n)[:110]


prev_end = t0
busy = 0
for r in seg[: int(sys.argv[3]) if len(sys.argv) > 3 else 100000]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1e3
    busy += en - st
    print(f"{(st - t0) / 1e3:10.1f} us  +{(en - st) / 1e3:8.1f}  gap {gap:7.1f}  {short(r['Kernel_Name'])}")
    prev_end = max(prev_end, en)
print(f"busy {busy / 1e6:.3f} ms of {(prev_end - t0) / 1e6:.3f} ms")
