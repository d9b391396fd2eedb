#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace --hip-trace directory of scripts/eigsh_trace_one.py and reports, for the LAST repetition: kernels,
wall time, GPU-busy time, launches per kernel name and the HIP calls made meanwhile."""
import csv, sys, glob, collections
d = sys.argv[1]
kt = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])))
ht = list(csv.DictReader(open(glob.glob(d + "/*/*hip_api_trace.csv")[0])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in kt)
gaps = [(ks[i + 1][0] - ks[i][1], i) for i in range(len(ks) - 1)]
big = sorted(gaps, reverse=True)[:2]; cut = max(i for g, i in big) + 1        # the two longest gaps separate the three repetitions
last = ks[cut:]
wall = last[-1][1] - last[0][0]; busy = sum(e - s for s, e, n in last)
print("last repetition: %d kernels, wall %.2f ms, GPU busy %.2f ms (%.0f %%)" % (len(last), wall / 1e6, busy / 1e6, 100 * busy / wall))
c = collections.Counter(); t = collections.Counter()
for s, e, n in last:
    nn = n.split("(")[0][:70]; c[nn] += 1; t[nn] += e - s
for nn, k in c.most_common(25):
    print("  %5d x %7.1f us  %s" % (k, t[nn] / k / 1e3, nn))
a, b = last[0][0], last[-1][1]
hc = collections.Counter(); hd = collections.Counter()
for r in ht:
    s = int(r["Start_Timestamp"])
    if a - 200000 <= s <= b:
        hc[r["Function"]] += 1; hd[r["Function"]] += int(r["End_Timestamp"]) - s
for f, k in hc.most_common(12):
    print("  HIP %6d x %7.1f us  %s" % (k, hd[f] / k / 1e3, f))
