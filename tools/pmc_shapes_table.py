"""usage: pmc_shapes_table.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv>
Per-shape HBM-side bytes of tools/pmc_shapes.py's second dispatch of each shape next to the algorithmic operand bytes."""
import csv, sys
from pmc_shapes import SHAPES
def rows(path, counter):
    out = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and ("gemm" in r["Kernel_Name"] or "splitk" in r["Kernel_Name"]):
            out.append((int(r.get("Dispatch_Id", len(out))), r["Kernel_Name"], float(r["Counter_Value"]) * 1024))
    out.sort()
    return out
fe, wr = rows(sys.argv[1], "FETCH_SIZE"), rows(sys.argv[2], "WRITE_SIZE")
def per_shape(rs):
    # every shape = two identical groups of dispatches (GEMM [+ split-K reduce]); take the second group
    groups, i = [], 0
    for s in SHAPES:
        j = i
        names = []
        while j < len(rs):
            names.append(rs[j][1]); j += 1
            half = len(names) // 2
            if len(names) % 2 == 0 and names[:half] == names[half:]:
                nxt_same = j < len(rs) and rs[j][1] == names[0] and len(names) == 2 and False
                break
        half = (j - i) // 2
        groups.append(rs[i + half:j]); i = j
    return groups
gf, gw = per_shape(fe), per_shape(wr)
print(f"{'shape (kind, M, N, K)':44s} {'kernel(s)':34s} {'fetch raw MB':>12s} {'x2 MB':>8s} {'write MB':>9s} {'operands in MB':>14s} {'out MB':>7s}")
for s, a, b in zip(SHAPES, gf, gw):
    kind, bt, h, w_, cin, cout = s
    M = bt * h * w_; K = 9 * cin if kind == "conv" else cin
    nout = cout // 2 if kind == "geglu" else cout
    inb = (M * cin + cout * K) * 2 + (M * cout * 2 if kind == "linear" else 0)
    outb = M * nout * 2
    ks = "+".join(sorted({("pp" if "gemm_pp" in n else "ws" if "gemm_ws" in n else "reduce" if "splitk" in n else "4wave") for _, n, _ in a}))
    f = sum(v for _, _, v in a); w = sum(v for _, _, v in b)
    print(f"{str((kind, M, cout, K)):44s} {ks:34s} {f / 1e6:12.1f} {2 * f / 1e6:8.1f} {w / 1e6:9.1f} {inb / 1e6:14.1f} {outb / 1e6:7.1f}")
