"""Copies what tools/refresh_profiles_r05.sh left in gpurun_out/r05/ into profiles/r05/ and writes the summaries that are read off
several of those files (bench_events_ab.txt, force_dist_summary.txt), then rebuilds profiles/traffic_r05*.json.
   python tools/collect_profiles_r05.py"""
import glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r05"), os.path.join(ROOT, "profiles", "r05")
os.makedirs(DST, exist_ok=True)
keep = ["bench_n1.json", "bench_n1_under_rocprof.json", "bench_n1_kernel_stats.csv", "bench_n1_timed_region_from_trace.txt",
        "pmc_fetch_size_bench.txt", "pmc_write_size_bench.txt", "pmc_fetch_size_configs.txt", "pmc_write_size_configs.txt",
        "configs_single_gpu.txt", "c5_timed_region_from_trace.txt", "c5_kernel_stats.csv", "c5_loop.txt",
        "c5_timed_region_from_trace_baseline.txt", "c5_kernel_stats_baseline.csv",
        "force_dist_2p27_kernel_stats.csv", "bench_force_dist_2p27.json", "bench_force_dist_2p27_under_rocprof.json",
        "distributions_2p28.txt", "size_ladder_pairs.txt", "bench_ladder_reference_format.txt",
        "finish_stamps_u32.txt", "finish_stamps_u64_rank16.txt", "finish_stamps_u64_all_rounds.txt",
        "fuzz_library.txt", "fuzz_one_object.txt", "fuzz_segments.txt", "pytest_gpu_head.txt"]
keep += [os.path.basename(f) for f in glob.glob(os.path.join(SRC, "force_dist_*.json"))]
for f in keep:
    if os.path.exists(os.path.join(SRC, f)):
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, f))
    else:
        print("missing:", f)

# ---- what the events cost the timed region
rows = []
for name, what in (("bench_n1_events_everywhere", "events at every kernel boundary of every pass (round 4: --full-kernel-events)"),
                   ("bench_n1_events_light", "events only around the scatter launches and the in-LDS pass (round 5, the default)"),
                   ("bench_n1_no_events", "no per-kernel events (--no-kernel-events: no roofline object)")):
    p = os.path.join(SRC, name + ".json")
    if os.path.exists(p):
        d = json.load(open(p))
        sd = d.get("step_device_ms") or {}
        rows.append("%-90s ms_per_step %.4f   device ms per step: median %s  min %s" % (what, d["ms_per_step"], sd.get("median"), sd.get("min")))
open(os.path.join(DST, "bench_events_ab.txt"), "w").write(
    "python bench.py --no-cpu-baseline --no-alt, three runs in one gpurun call on one MI355X (tools/refresh_profiles_r05.sh): what the HIP events\n"
    "recorded inside the timed region cost it.  2^28 uint32 key + value pairs, 10 timed sorts.\n\n" + "\n".join(rows) + "\n")

# ---- a rank's compute of the sharded sort
names = [("force_dist_world1", "world 1 (256 buckets on the rank: runs of 2048)"),
         ("force_dist_world1_three_passes", "the same by three segmented passes (GLU_HIP_SEG_LDS_FINISH=0: round 4's local sort)"),
         ("force_dist_as_rank_of_2", "key range of one rank of 2 (128 buckets: runs of 4096)"),
         ("force_dist_as_rank_of_4", "key range of one rank of 4 (64 buckets: runs of 8192)"),
         ("force_dist_as_rank_of_8", "key range of one rank of 8 (32 buckets: runs of 16384) -- default: no attempt, three passes"),
         ("force_dist_as_rank_of_8_tile17408", "  the same, runs taken whole by a 1024 x 17 tile, one workgroup per CU (GLU_HIP_SEG_MAX_GEO=5)"),
         ("force_dist_as_rank_of_8_split2", "  the same, every run split over 2 workgroups of a 9216-pair tile (GLU_HIP_SEG_SPLIT_MAX=3 GLU_HIP_SEG_SPLIT_GEO=4)"),
         ("force_dist_as_rank_of_8_split4", "  the same, every run split over 4 workgroups of a 4608-pair tile (GLU_HIP_SEG_SPLIT_MAX=3)")]
out = ["A rank's compute of the sharded sort on ONE GPU: python bench.py --force-dist --log2-keys 27 --no-one-gpu --pipeline-depth 1 [--as-rank-of R]",
       "(the N > 1 code path at world size 1; --as-rank-of R draws the keys from the key range one rank of R owns, so that the local sort has",
       "the runs it would have there).  2^27 pairs, phases_ms_rank0 per sort; the exchange is a local copy here (2 GiB of HBM traffic).",
       "NOT a multi-GPU measurement: no multi-GPU box has been available to any session.", ""]
for f, what in names:
    p = os.path.join(SRC, f + ".json")
    if not os.path.exists(p):
        continue
    d = json.load(open(p))
    ph, l = d["phases_ms_rank0"], d.get("local_sort_in_lds_rank0", {})
    out.append("%-118s partition %.3f  local sort %.3f  (+ copy %.3f)   in LDS: %s  tile %s x split %s  longest run %s  verified %s" % (
        what, ph["partition"], ph["local_sort"], ph["all_to_all"], "yes" if l.get("accepted") else "no", l.get("tile"), l.get("split"), l.get("longest_run"), d.get("verified")))
open(os.path.join(DST, "force_dist_summary.txt"), "w").write("\n".join(out) + "\n")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_traffic_json.py"), "r05"])
print(open(os.path.join(DST, "bench_events_ab.txt")).read())
print(open(os.path.join(DST, "force_dist_summary.txt")).read())
