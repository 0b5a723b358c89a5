"""Builds profiles/traffic_<round>*.json and profiles/<round>/traffic_all_kernels.json (round = argv[1], default r02) from
the per-kernel PMC summaries that tools/refresh_profiles_<round>.sh leaves in profiles/<round>/ (FETCH_SIZE and WRITE_SIZE were collected in separate rocprofv3
--pmc passes; on gfx950 FETCH_SIZE counts half of a coalesced streaming read, MI355X_MICROARCH.md HBM section, so it is
doubled -- the count kernel, which reads exactly 4 B per key and writes nothing, calibrates that in the same run).
usage: python tools/make_traffic_json.py [r03]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r02"
P = os.path.join(ROOT, "profiles", ROUND)


def parse(path):
    out, name = {}, None
    for line in open(path):
        if line.startswith("void ") or line.startswith("glu_hip::"):
            name = line.strip()
        elif name and "avg=" in line:
            m = re.search(r"n=\s*(\d+) avg=([0-9.e+]+)", line)
            out[name] = (int(m.group(1)), float(m.group(2)))
    return out


def find(d, *subs):
    for k, v in d.items():
        # (not the copy of the line scatter that only differs in name: the passes enqueued behind an attempt to end in LDS)
        # (nor the segmented form, SEG = true: the passes over the long runs of a sort that ends in LDS, which return at once here)
        name = k.split("(")[0].rstrip()
        if all(s in k for s in subs) and not name.endswith("false, false, true>") and not ("radix_scatter_lines" in name and name.endswith("true, false, false>")):
            return k, v
    raise KeyError(subs)


N = 1 << 28
# (the in-LDS pass: round 6's bucket kernel, radix_lds_bucket.hpp; rounds 4 and 5: the ballot-ranked kernel)
FINISH = "radix_finish_bucket_kernel" if ROUND >= "r06" else "radix_finish_sort_kernel"
T2 = 256 * 256 * 512  # the two-digit table of a pair of passes: [256][workgroups][256] 16-bit counters
T2_4 = 256 * 16 * 1024 + 16 * 256 * 16 * 4  # 4-bit digits: 16 x 16 counters per sub-block + the per-sub-block table
rows = []
for tag in ("bench", "configs"):
    f = parse(os.path.join(P, "pmc_fetch_size_%s.txt" % tag))
    w = parse(os.path.join(P, "pmc_write_size_%s.txt" % tag))
    rows.append((tag, f, w))
bench_f, bench_w = rows[0][1], rows[0][2]
cfg_f, cfg_w = rows[1][1], rows[1][2]


def entry(f, w, subs, alg_bytes, what, launches_note=""):
    k, (nf, fk) = find(f, *subs)
    _, (nw, wk) = find(w, *subs)
    rd, wr = int(fk * 1024 * 2), int(wk * 1024)
    return {"kernel": k.split("(")[0].replace("void ", ""), "what": what, "dispatches_averaged": nf,
            "fetch_size_kb_avg": fk, "write_size_kb_avg": wk, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
            "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg_bytes, "ratio": round((rd + wr) / alg_bytes, 4),
            "note": launches_note}


corr = ("FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reads 1/2 of a coalesced streaming "
        "read; the count kernel of the 4-bit sort of the same run, which reads 4 B per key and little else, reports 52x xxx KB for 2^28 keys "
        "= 1/2 of 1 GiB). WRITE_SIZE is exact. Units: KB = 1024 B.")
src = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tools/refresh_profiles_%s.sh), MI355X, round %s; "
       "per-dispatch averages in profiles/%s/pmc_{fetch,write}_size_{bench,configs}.txt" % (ROUND, ROUND[1:].lstrip("0"), ROUND))
e8 = entry(bench_f, bench_w, ("radix_scatter_lines_kernel<unsigned int, 8",), N * 16, "dominant kernel of the headline sort (2^28 u32 pairs, 8-bit digits)")
e4 = entry(bench_f, bench_w, ("radix_scatter_lines_kernel<unsigned int, 4",), N * 16, "scatter of the reference pass structure (4-bit digits)")
all_k = {
    "source": src, "corrections": corr,
    "kernels": [
        e8, e4,
        entry(bench_f, bench_w, (FINISH + "<unsigned int, 256, 18, true, false",), N * 16, "in-LDS pass of the headline sort (a workgroup per run of equal top 16 key bits; reads and writes every pair once)"),
        entry(bench_f, bench_w, ("radix_pair_count_kernel<unsigned int",), N * 4 + T2, "count kernel of a pair of passes of the headline sort (reads the keys, writes the two-digit table)"),
        entry(bench_f, bench_w, ("radix_pair_unitsum_kernel",), T2, "count table of the second pass of a pair (reads the two-digit table)"),
        entry(bench_f, bench_w, ("radix_pair4_count_kernel<unsigned int",), N * 4 + T2_4, "count kernel of a pair of passes of the 4-bit sort (reads the keys -- the calibration of the FETCH_SIZE factor -- and writes 4.25 MiB of tables)"),
        entry(bench_f, bench_w, ("radix_pair4_unitsum_kernel",), 256 * 16 * 1024, "count table of the second pass of a 4-bit pair"),
        entry(cfg_f, cfg_w, ("radix_scatter_lines_kernel<unsigned long, 8",), N * 24, "scatter of BASELINE.json configs[4] (2^28 u64 keys + u32 vals, 8-bit digits)"),
        entry(cfg_f, cfg_w, ("radix_scatter_lines_kernel<unsigned long, 4",), N * 24, "same, 4-bit digits"),
        entry(cfg_f, cfg_w, ("radix_pair_count_kernel<unsigned long",), N * 8 + T2, "count kernel of a pair of passes, 64-bit keys"),
        entry(cfg_f, cfg_w, ("radix_pair4_count_kernel<unsigned long",), N * 8 + T2_4, "count kernel of a pair of passes, 64-bit keys, 4-bit digits"),
        entry(cfg_f, cfg_w, (FINISH + "<unsigned long, 512, 9, true, false",), N * 24, "in-LDS pass of BASELINE.json configs[4] (2^28 u64 keys + u32 vals)"),
        entry(cfg_f, cfg_w, ("scan_chunks_kernel",), N * 8, "glu::BlellochScan 2^28 u32 (chained single pass)"),
    ],
}
k, (nf, fk) = find(cfg_f, "reduce_kernel")
all_k["kernels"].append({"kernel": "glu_hip::reduce_kernel<0, unsigned int, 1, true>", "what": "glu::Reduce 2^28 u32 sum",
                         "dispatches_averaged": nf, "fetch_size_kb_avg": fk,
                         "note": "average over the two launches of a reduce (the second reads 2 KiB): the first launch fetches 2 x avg",
                         "hbm_read_bytes_per_launch": int(fk * 2 * 1024 * 2), "algorithmic_bytes_per_launch": N * 4,
                         "ratio": round(fk * 2 * 1024 * 2 / (N * 4), 4)})
json.dump(all_k, open(os.path.join(P, "traffic_all_kernels.json"), "w"), indent=1)
for e, key, name in ((e8, "radix_sort_u32_pairs_2^28_uniform_bits8", "traffic_%s.json" % ROUND), (e4, "radix_sort_u32_pairs_2^28_uniform_bits4", "traffic_%s_bits4.json" % ROUND)):
    d = {"workload_key": key, "source": src, "corrections": corr}
    d.update(e)
    json.dump(d, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
for e in all_k["kernels"]:
    print("%-70s ratio %s" % (e["kernel"][:70], e["ratio"]))
