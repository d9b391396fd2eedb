"""Summarise rocprofv3 --pmc csv outputs: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections, os
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    print("==", os.path.relpath(f, root))
    for (kn, cn), (s, n) in sorted(acc.items()):
        if "k_csr_spmv" in kn or "k_csr_il" in kn or "k_csr_nd" in kn or "k_kron2" in kn or "k_dwpass" in kn or "k_up" in kn:
            print("  %-60s %-24s mean/dispatch %.6g  (n=%d)" % (kn, cn, s / n, n))
