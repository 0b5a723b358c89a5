"""Copies what tools/refresh_profiles_r06.sh and tools/r06_validate.sh left in gpurun_out/r06/ into profiles/r06/, writes
force_dist_summary.txt, and rebuilds profiles/traffic_r06*.json (tools/make_traffic_json.py r06).
   python tools/collect_profiles_r06.py"""
import glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r06"), os.path.join(ROOT, "profiles", "r06")
os.makedirs(DST, exist_ok=True)
keep = ["bench_n1.json", "bench_n1_default_flags.json", "bench_n1_under_rocprof.json", "bench_n1_kernel_stats.csv",
        "bench_n1_timed_region_from_trace.txt", "pmc_fetch_size_bench.txt", "pmc_write_size_bench.txt", "pmc_fetch_size_configs.txt",
        "pmc_write_size_configs.txt", "configs_single_gpu.txt", "c5_timed_region_from_trace.txt", "c5_kernel_stats.csv", "c5_loop.txt",
        "last_sort_kernels_2p28.txt", "last_sort_kernels_2p28_u64.txt", "refused_sort_kernels_three_values.txt", "three_values_sort_kernels.txt", "zipf_sort_kernels.txt",
        "distinct_1000_sort_kernels.txt", "distinct_2p20_sort_kernels.txt", "zeros_0p01_sort_kernels.txt", "zeros_1_sort_kernels.txt", "caller_pairs_probe.txt", "fuzz_heavy.txt", "distributions_2p28.txt", "distributions_2p28_u64.txt", "size_ladder_pairs.txt",
        "bench_ladder_reference_format.txt", "finish_bucket_bench.txt", "finish_midsize_any.txt",
        "smoke_head.txt", "pytest_gpu_head.txt", "fuzz_library.txt", "fuzz_library_large.txt", "fuzz_one_object.txt", "fuzz_segments.txt"]
keep += [os.path.basename(f) for f in glob.glob(os.path.join(SRC, "force_dist_*.json"))]
for f in keep:
    if os.path.exists(os.path.join(SRC, f)):
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, f))
    else:
        print("missing:", f)
names = [("force_dist_world1", "world 1 (256 buckets on the rank: runs of 2048)"),
         ("force_dist_as_rank_of_2", "key range of one rank of 2 (128 buckets: runs of 4096)"),
         ("force_dist_as_rank_of_4", "key range of one rank of 4 (64 buckets: runs of 8192)"),
         ("force_dist_as_rank_of_8", "key range of one rank of 8 (32 buckets: runs of 16384) -- default: no attempt, three passes")]
out = ["A rank's compute of the sharded sort on ONE GPU: python bench.py --force-dist --log2-keys 27 --no-one-gpu --pipeline-depth 1 [--as-rank-of R]",
       "(the N > 1 code path at world size 1; --as-rank-of R draws the keys from the key range one rank of R owns, so that the local sort has",
       "the runs it would have there).  2^27 pairs, phases_ms_rank0 per sort; the exchange is a local copy here (2 GiB of HBM traffic).",
       "NOT a multi-GPU measurement: no multi-GPU box has been available to any session.", ""]
for f, what in names:
    p = os.path.join(SRC, f + ".json")
    if not os.path.exists(p):
        continue
    d = json.loads(open(p).read().strip().splitlines()[-1])
    ph, l = d["phases_ms_rank0"], d.get("local_sort_in_lds_rank0", {})
    out.append("%-118s partition %.3f  local sort %.3f  (+ copy %.3f)   in LDS: %s  tile %s x split %s  longest run %s  verified %s" % (
        what, ph["partition"], ph["local_sort"], ph["all_to_all"], "yes" if l.get("accepted") else "no", l.get("tile"), l.get("split"), l.get("longest_run"), d.get("verified")))
open(os.path.join(DST, "force_dist_summary.txt"), "w").write("\n".join(out) + "\n")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_traffic_json.py"), "r06"])
print(open(os.path.join(DST, "force_dist_summary.txt")).read())
