"""Per-sort kernel durations from a rocprofv3 --kernel-trace CSV.
   python tools/trace_summary.py <kernel_trace.csv> [last N sorts]
       one line per sort (a sort ends with radix_finalize_kernel): C = leader count kernels that read keys, S = scatter launches
       that moved data, F = the in-LDS pass (us), the other kernels summed, and first start to last end
   python tools/trace_summary.py <kernel_trace.csv> --bench K
       what bench.py's timed region looks like in the trace: the last K sorts that ended in LDS (bench.py's K timed steps are
       the last sorts of its headline sort object; the comparison legs that follow use other pass structures) -- average,
       min and max of their scatter launches and of their in-LDS pass, to set beside bench.py's roofline object
   python tools/trace_summary.py <kernel_trace.csv> --sorts K
       the same for any key type (tools/sort_loop.py): the last K sorts that ended in LDS, per kind of kernel, the other
       kernels by name, and the time between first kernel start and last kernel end that no kernel of the sort covers"""
import csv, sys

def targs(n):
    """the template arguments of a kernel name"""
    a = n.split("(")[0].rstrip()
    return [x.strip() for x in a[a.index("<") + 1:a.rindex(">")].split(",")] if "<" in a else []


def behind(n):  # (the last template argument is true)
    t = targs(n)
    return bool(t) and t[-1] == "true"


def segmented(n):  # radix_scatter_lines_kernel<..., SEG = true, ...>: the long-run passes / a segmented sort (13 arguments, SEG the 13th)
    t = targs(n)
    return len(t) >= 13 and t[12] == "true"


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
bench = "--bench" in sys.argv
per_sorts = "--sorts" in sys.argv
last = int(sys.argv[sys.argv.index("--bench") + 1]) if bench else int(sys.argv[sys.argv.index("--sorts") + 1]) if per_sorts else (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
sorts, cur, other, t0 = [], [], 0.0, None
names, busy, t_end = {}, 0.0, None
all_scatter8 = []
for r in rows:
    n = r["Kernel_Name"]
    if "glu_hip::" not in n:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if t0 is None:
        t0 = int(r["Start_Timestamp"])
    if "radix_scatter_lines_kernel<unsigned int, 8" in n:
        all_scatter8.append((n, d))
    # (time covered by at least one kernel of the sort: kernels of one stream run one after the other)
    busy += (int(r["End_Timestamp"]) - max(int(r["Start_Timestamp"]), t_end or 0)) / 1e3 if (t_end or 0) < int(r["End_Timestamp"]) else 0.0
    t_end = max(t_end or 0, int(r["End_Timestamp"]))
    short = n.split("glu_hip::")[1].split("<")[0].split("(")[0]
    if "radix_finalize" in n:
        names[short] = names.get(short, 0.0) + d
        sorts.append((cur, other + d, (int(r["End_Timestamp"]) - t0) / 1e3, names, busy))
        cur, other, t0, names, busy, t_end = [], 0.0, None, {}, 0.0, None
    # (round 6: the kernels of the sequence not taken run on the sort object's side stream and wait there for room on the CUs -- a
    # launch that returns at once can show hundreds of microseconds beside a streaming kernel.  They are told apart by NAME: the
    # scatter of the passes behind an attempt has BEHIND_ATTEMPT = true as its last template argument, the leader's count kernel
    # is the instantiation that collects key bits.)
    elif "radix_scatter_lines" in n and d > 100 and not behind(n) and not segmented(n):
        cur.append(("S", d))
    elif ("radix_finish_sort" in n or "radix_finish_bucket" in n) and d > 100:
        cur.append(("F", d))
    elif "pair_count" in n and d > 100 and behind(n):  # (<KeyT, TILE, XF, COLLECT = true>: the first pass's)
        cur.append(("C", d))
    else:
        other += d
        names[short] = names.get(short, 0.0) + d
if per_sorts:
    def st(v):
        return "n %d  average %.1f us  min %.1f  max %.1f" % (len(v), sum(v) / len(v), min(v), max(v)) if v else "none"
    ended = [s for s in sorts if any(k == "F" for k, _ in s[0])]
    timed = ended[-last:]
    print("sorts in the trace: %d, of them ended in LDS: %d; the last %d of those:" % (len(sorts), len(ended), len(timed)))
    print("  scatter launches that moved data:   %s" % st([d for s in timed for k, d in s[0] if k == "S"]))
    print("  in-LDS pass (one per sort):         %s" % st([d for s in timed for k, d in s[0] if k == "F"]))
    print("  leader count kernel (one per sort): %s" % st([d for s in timed for k, d in s[0] if k == "C"]))
    print("  first kernel start to last kernel end per sort: %s" % st([s[2] for s in timed]))
    print("  of that, NOT under the leader's count, a scatter that moved data or the in-LDS pass: %s" % st([s[2] - sum(d for _, d in s[0]) for s in timed]))
    print("  of that, covered by no kernel at all (launch gaps): %s" % st([s[2] - s[4] for s in timed]))
    print("  the other kernels by name, average us per sort from launch to end -- side-stream kernels wait for room beside the streaming")
    print("  kernels, so these overlap them and do not add up to the figure above:")
    agg = {}
    for s in timed:
        for k, d in s[3].items():
            agg[k] = agg.get(k, 0.0) + d / len(timed)
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1]):
        print("    %-40s %7.1f" % (k, d))
    sys.exit(0)
if not bench:
    for cur, other, span, _, _ in sorts[-last:]:
        print(" ".join("%s%d" % (k, round(d)) for k, d in cur), " other kernels %d us, first start to last end %d us" % (other, span))
    sys.exit(0)


def stat(v):
    return "n %d  average %.1f us  min %.1f  max %.1f" % (len(v), sum(v) / len(v), min(v), max(v)) if v else "none"


ended = [s for s in sorts if any(k == "F" for k, _ in s[0])]
timed = ended[-last:]
plain = [d for n, d in all_scatter8 if not behind(n)]
behind_launches = [d for n, d in all_scatter8 if behind(n)]
print("radix_scatter_lines_kernel<u32, 8, ...> launches in the whole process (calibration sorts of prepare, warm-up, timed steps, the")
print("four-pass comparison leg):  under its plain name %s" % stat(plain))
print("  under the name of the passes enqueued behind an attempt to end in LDS (BEHIND_ATTEMPT = true; they return at once when the")
print("  attempt was accepted; on the side stream they wait for room beside the streaming kernels): %s" % stat(behind_launches))
print("sorts that ended in LDS: %d; the last %d of them (bench.py's timed steps):" % (len(ended), len(timed)))
print("  scatter launches (two per sort):  %s" % stat([d for s in timed for k, d in s[0] if k == "S"]))
print("  in-LDS pass (one per sort):       %s" % stat([d for s in timed for k, d in s[0] if k == "F"]))
print("  leader count kernel (one):        %s" % stat([d for s in timed for k, d in s[0] if k == "C"]))
print("  first start to last end NOT under the leader's count, a scatter that moved data or the in-LDS pass, per sort: %s" % stat([s[2] - sum(d for _, d in s[0]) for s in timed]))
print("  first kernel start to last kernel end per sort:  %s" % stat([s[2] for s in timed]))
