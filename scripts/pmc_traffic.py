#!/usr/bin/env python3
"""HBM-side traffic of one product per storage from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
(scripts/profile_round.sh): mean counter value per dispatch of every product kernel, corrected as
MI355X_MICROARCH.md prescribes (units of 1 KiB; FETCH_SIZE tallies 64 B per 128-B request on gfx950 -> x2),
summed over the kernels that make one product.  Prints a summary and writes pmc_traffic.json next to it."""
import collections, csv, glob, hashlib, json, os, sys
root = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


PRODUCT_SOURCES = ("exchange.hpp", "kernels.hpp", "product.hpp", "sector_dev.hpp", "sector_host.hpp")


def source_stamp():      # same digest as bench.py: the summary is only quoted while the product-kernel sources are the ones it was measured on
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dmft-lanc-ed_amd", "csrc")
    for f in PRODUCT_SOURCES:
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


STAMP = source_stamp()
PRODUCT = {"csr": ("k_csr_spmv2", "k_csr_il"), "kron": ("k_uppass", "k_dwpass"), "direct": ("k_uppass", "k_dwpass")}      # the on-the-fly tables are derived once per sector (first product), not per product
out = []
for st, prefixes in PRODUCT.items():
    per = {}
    for cn in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob(os.path.join(root, "pmc_%s_%s" % (st, cn), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != cn:
                    continue
                kn = r["Kernel_Name"].split("(")[0].replace("void edgpu::", "").replace("edgpu::", "")
                if any(p in kn for p in prefixes):
                    acc[kn][0] += float(r["Counter_Value"]); acc[kn][1] += 1
        per[cn] = {k: (s / n, n) for k, (s, n) in acc.items()}
    if not per["FETCH_SIZE"]:
        continue
    # a product (or fused iteration) launches each kernel name once
    tot = 0.0
    print("== storage=%s" % st)
    for kn in sorted(per["FETCH_SIZE"]):
        fs, n = per["FETCH_SIZE"][kn]; ws = per["WRITE_SIZE"].get(kn, (0.0, 0))[0]
        mult = 1
        b = (2 * fs + ws) * 1024 * mult
        # only the Lanczos-fused (epilogue) instantiations belong to the timed loop
        print("  %-50s FETCH_SIZE %.4g KiB  WRITE_SIZE %.4g KiB  -> %.3f GB per product (n=%d)" % (kn, fs, ws, b / 1e9, n))
        tot += b
    print("  total %.2f GB per product" % (tot / 1e9))
    # kernel time per product under rocprofv3 (--kernel-trace --stats of the same command): mean duration of the same kernels
    rp_ms = None
    stats = os.path.join(root, "bench_c3_%s_kernel_stats.csv" % st)
    if os.path.exists(stats):
        rp_ms = 0.0
        for r in csv.DictReader(open(stats)):
            kn = r["Name"].split("(")[0].replace("void edgpu::", "").replace("edgpu::", "")
            if kn in per["FETCH_SIZE"]:
                rp_ms += float(r["AverageNs"]) * 1e-6
        print("  kernel time per product under rocprofv3: %.3f ms" % rp_ms)
    out.append({"storage": st, "nbath": 15, "n_gpus": 1, "hbm_bytes_per_launch": tot, "source_stamp": STAMP, "kernels": sorted(per["FETCH_SIZE"]),
                "rocprof_kernel_ms_per_product": rp_ms,
                "file": "profiles/%s/pmc_traffic_summary.txt" % os.environ.get("PROFILE_ROUND_NAME", "r04"),
                "source": "profiles: (2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch, summed over the kernels of one product (%s); FETCH_SIZE doubled per MI355X_MICROARCH.md (64 B tallied per 128-B request); fabric-side count, Infinity-Cache hits included" % ", ".join(sorted(per["FETCH_SIZE"]))})
json.dump(out, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
