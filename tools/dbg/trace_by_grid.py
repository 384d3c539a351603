"""Summarise a rocprofv3 --kernel-trace CSV by (kernel, grid size): launches, total / average duration, share of the kernel time --
the view that shows WHICH launches of a kernel run under-filled (a sponge kernel over 2^20 lanes against the same kernel over 2^14).
usage: trace_by_grid.py <kernel_trace.csv> [top N rows, default 60]  -> text on stdout"""
import csv, sys
rows = {}
total = 0.0
first, last = None, 0
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void mp2g::", "").replace("mp2g::", "")
        g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        d = (e - s) / 1e3
        a = rows.setdefault((k, g), [0, 0.0, 1e30, 0.0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
        total += d
        first = s if first is None else min(first, s)
        last = max(last, e)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
print(f"kernel time summed {total / 1e3:.1f} ms over a span of {(last - first) / 1e6:.1f} ms (overlap factor {total / 1e3 / ((last - first) / 1e6):.2f})")
print(f"{'kernel':62s} {'lanes':>10s} {'calls':>7s} {'total ms':>9s} {'share':>6s} {'avg us':>9s} {'min us':>9s} {'ns/lane':>8s}")
for (k, g), (c, t, mn, mx) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:n]:
    print(f"{k[:62]:62s} {g:10d} {c:7d} {t / 1e3:9.2f} {100 * t / total:5.1f}% {t / c:9.1f} {mn:9.1f} {1e3 * t / c / g:8.3f}")
