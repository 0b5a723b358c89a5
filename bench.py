"""bench.py -- Mkeys/s of the hot path (glu::RadixSort::operator(), reference glu/RadixSort.hpp:273-334) on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Both forms work for N > 1: the plain command (no WORLD_SIZE in the environment) starts the second form as a CHILD process
before anything touches a GPU, forwards rank 0's JSON line and returns the child's exit code (launch_ranks below).

A "step" is one full sort of one batch of synthetic unsorted input that is already resident in HBM.
  N = 1   BASELINE.json configs[2]: 2^28 uniform-random uint32 keys + uint32 values (vals = iota), in place in the
          caller's two arrays, scratch pre-allocated (the reference's benchmark does the same,
          test/radix_sort_tests.cpp:187).  Every step sorts its own pristine copy of the input, so no restore copy
          sits inside the timed region.
  N > 1   BASELINE.json configs[3]: 2^27 pairs per GPU (2^30 at N = 8), rank r holds slice r; top-8-bit bucket
          partition -> one all-to-all (RCCL over xGMI) -> local sort (gl-radix-sort_amd/glu_hip/dist.py).
Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (the scatter pass) from HIP-event timings
taken inside the timed region; `cpu_baseline` is std::sort on the host cores over a bounded sample.
For every N, `value` / `ms_per_step` are ONE SORT AT A TIME (K sorts enqueued back to back on one stream), so that the
1/2/4/8-GPU curve compares like with like; N > 1 also reports the two-sorts-in-flight throughput as `value_depth2` and
`one_gpu` = this very run's single-GPU figures (2^28 pairs on every rank's own GPU, depth 1 and depth 2), with
`speedup_vs_1gpu_depth1` / `speedup_vs_1gpu_depth2` taken against them.
Exit codes: 0 = a line was printed (a line that carries `native_error` is the torch.distributed transport's measurement:
the native transport raised or hung and was abandoned -- every rank then leaves with 0 so that the launcher keeps the
line); anything else = no valid line.
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s)
KEY_BYTES, VAL_BYTES = 4, 4


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--log2-keys", type=int, default=None, help="pairs per GPU = 2^this (default 28 at N=1, 27 at N>1)")
    p.add_argument("--digit-bits", type=int, default=None, help="4 or 8 (default: library default)")
    p.add_argument("--keys", default="uniform", choices=["uniform", "zero"],
                   help="uniform = headline; zero = the reference README's benchmark input")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample-log2", type=int, default=None,
                   help="time the CPU baseline on a generated 2^this sample instead of the GPU workload's own array")
    p.add_argument("--no-verify", action="store_true")
    p.add_argument("--no-kernel-events", action="store_true",
                   help="do not record per-kernel HIP events in the timed region (roofline fields become null)")
    p.add_argument("--full-kernel-events", action="store_true",
                   help="record events at EVERY kernel boundary inside the timed region (the round-4 behaviour; for the A/B in "
                        "profiles/r05/bench_events_ab.txt) instead of only around the kernels that move data")
    p.add_argument("--no-alt", action="store_true", help="skip the extra 4-bit-digit (reference pass structure) measurement")
    p.add_argument("--pipeline-depth", type=int, default=2,
                   help="N>1: consecutive independent sorts in flight in the line's timed region (own stream, buffers and "
                        "communicator each; the one-at-a-time figure is reported beside it as value_depth1); 1 = one at a time only")
    p.add_argument("--reserved-cus", type=int, default=8,
                   help="N>1, pipelined measurement: CUs the sort kernels leave to the RCCL kernels of the other sort in flight")
    p.add_argument("--transport", default="native", choices=["native", "torch"],
                   help="N>1: native = the whole sharded sort inside libglu_hip.so (glu_dist_*: its own RCCL communicator, one grouped "
                        "exchange); torch = torch.distributed collectives around the same C-ABI device work")
    p.add_argument("--no-transport-fallback", action="store_true",
                   help="N>1, native transport: do not measure the torch.distributed transport first (it is the line rank 0 prints "
                        "if the native run raises or hangs)")
    p.add_argument("--no-one-gpu", action="store_true",
                   help="N>1: skip the single-GPU figures of the same run (one_gpu, speedup_vs_1gpu_*)")
    p.add_argument("--one-gpu-log2", type=int, default=None,
                   help="N>1: pairs of the single-GPU figures = 2^this (default 28, the N = 1 workload; pairs per GPU + 1 in a rehearsal)")
    p.add_argument("--force-dist", action="store_true",
                   help="run the multi-GPU code path (partition + all-to-all + local sort) even with one rank")
    p.add_argument("--as-rank-of", type=int, default=1,
                   help="with --force-dist: the keys are drawn from the key range ONE rank of an R-GPU sort owns (its 256 / R "
                        "buckets), so that this GPU's local sort has the runs it would have there (2^27 pairs at R = 8: 16384 per "
                        "(bucket, next byte)); a rehearsal of a rank's compute, not a line for the scaling curve")
    p.add_argument("--rehearse-one-gpu", action="store_true",
                   help="NOT a measurement: run the N>1 code path with all ranks on GPU 0 (gloo process group, glu_dist over "
                        "the file transport named by GLU_HIP_RCCL_LIB, tests/cpp/mock_rccl.cpp) to check that the launch, the "
                        "collectives and the JSON line work before an 8-GPU node runs them")
    return p.parse_args()


def make_input(torch, n, kind, seed, device, index_base=0, as_rank_of=1):
    """keys: uniform over the full [0, 2^32) (bit 31 set half of the time), vals: global index mod 2^32;
    both held as int32 bit patterns.  as_rank_of = R > 1: uniform over the first 1 / R of the key space (the buckets the first
    rank of R owns)."""
    g = torch.Generator(device=device)
    g.manual_seed(0x5EED + seed)
    if kind == "zero":
        keys = torch.zeros(n, dtype=torch.int32, device=device)
    elif as_rank_of > 1:
        keys = torch.randint(0, 2**32 // as_rank_of, (n,), dtype=torch.int64, device=device, generator=g)
        keys = torch.where(keys >= 2**31, keys - 2**32, keys).to(torch.int32)
    else:
        keys = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device=device, generator=g)
    vals = (torch.arange(n, dtype=torch.int64, device=device) + index_base).to(torch.int32)
    return keys, vals


def verify_sorted(torch, keys0, out_k, out_v, vals_are_local_iota):
    """GPU-side checks: keys ascending as unsigned; (with vals = local iota) out_k[i] == keys0[out_v[i]] and
    equal keys keep ascending values (stability)."""
    flipped = out_k ^ (-2**31)  # unsigned order == signed order of key ^ 0x80000000
    ok = bool((flipped[1:] >= flipped[:-1]).all())
    if vals_are_local_iota:
        idx = out_v.to(torch.int64) & 0xFFFFFFFF
        ok = ok and bool((keys0[idx] == out_k).all())
        eq = out_k[1:] == out_k[:-1]
        ok = ok and bool((idx[1:][eq] > idx[:-1][eq]).all())
    return ok


def pair_fingerprint(torch, keys, vals):
    """Two order-independent 64-bit sums over hashes of the (key, value) pairs (int64 arithmetic that wraps): equal for two
    arrays of pairs exactly when -- up to a 2^-64-ish chance -- they hold the same multiset of pairs.  Sums, not xor: every
    torch.distributed backend all-reduces sums."""
    def s64(x):  # a 64-bit constant as the int64 with the same bits
        return x - (1 << 64) if x >= 1 << 63 else x
    k = keys.to(torch.int64) & 0xFFFFFFFF
    v = vals.to(torch.int64) & 0xFFFFFFFF
    h = k * s64(0x9E3779B97F4A7C15) + v * s64(0xC2B2AE3D27D4EB4F) + s64(0x165667B19E3779F9)
    h = (h ^ ((h >> 29) & 0x7FFFFFFFF)) * s64(0xBF58476D1CE4E5B9)
    g = (h ^ ((h >> 32) & 0xFFFFFFFF)) * s64(0x94D049BB133111EB)
    return torch.stack([h.sum(), g.sum()])


def cpu_baseline(sample_log2, keys_host=None, vals_host=None):
    """std::sort of (key, val) structs by key on the host cores (oracle/cpu_sort_baseline.cpp).  With keys_host /
    vals_host (the GPU workload's own input array copied to the host) it sorts exactly that
    array, once single-threaded and once per thread count; without them a generated 2^sample_log2 sample."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    lib_path = os.path.join(ROOT, "oracle", "libglu_cpu_baseline.so")
    if not os.path.exists(lib_path):
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    L = ctypes.CDLL(lib_path)
    L.glu_cpu_sort_pairs.restype = ctypes.c_double
    L.glu_cpu_sort_pairs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int]
    L.glu_cpu_hardware_threads.restype = ctypes.c_uint
    cores = int(L.glu_cpu_hardware_threads()) or (os.cpu_count() or 1)
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    if keys_host is not None:
        keys, vals = keys_host, vals_host
        n = int(keys.shape[0])
        what = "the same array as the GPU workload (copied to the host), %d pairs = 2^%.2f" % (n, np.log2(n))
    else:
        n = 1 << sample_log2
        rng = np.random.default_rng(0x5EED)
        keys = rng.integers(0, 2**32, n, dtype=np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        what = "a generated 2^%d sample (fallback: not the GPU workload's array)" % sample_log2
    k1, v1 = keys.copy(), vals.copy()
    t1 = L.glu_cpu_sort_pairs(k1.ctypes.data, v1.ctypes.data, n, 1)
    assert (k1[1:] >= k1[:-1]).all()
    # __gnu_parallel::sort does not scale to every hardware thread of a big host: try a few widths, keep the best
    tried = {}
    for threads in sorted({cores, max(cores // 2, 1), max(cores // 4, 1), max(cores // 8, 1)}, reverse=True):
        if threads < 2:
            continue
        kp, vp = keys.copy(), vals.copy()
        tried[threads] = L.glu_cpu_sort_pairs(kp.ctypes.data, vp.ctypes.data, n, threads)
        assert (kp == k1).all()
    best_threads = min(tried, key=tried.get) if tried else 1
    tp = tried.get(best_threads, t1)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {
        "value": round(n / tp / 1e6, 2), "unit": "Mkeys/s", "cores": best_threads, "kind": "port",
        "sample": "%s; std::sort of uint32 key+val structs by key, one run single-threaded and one run per thread count "
                  "(__gnu_parallel::sort, best of %s threads reported; the host has %d hardware threads)"
                  % (what, "/".join(str(t) for t in sorted(tried)), cores),
        "host_hardware_threads": cores,
        "all_thread_counts_Mkeys_s": {str(t): round(n / tried[t] / 1e6, 2) for t in sorted(tried)},
        "single_thread_value": round(n / t1 / 1e6, 2), "cpu_model": model,
        "reference_published": "53.4 Mkeys/s (RTX 2060 SUPER, all-zero keys, reference README.md:133)",
    }


def one_gpu_figures(torch, G, device, log2n, K, W, kind, digit_bits, barrier, max_over_ranks):
    """N > 1 runs: what ONE GPU of this node does on the N = 1 workload, measured in this very run and the same way as the
    line's own figures -- K sorts of pristine copies between barrier + synchronize pairs, max over ranks (every rank sorts
    on its own GPU at the same time) -- once one sort at a time and once with two independent sorts in flight (two sorter
    objects on two streams: one GPU has no exchange to hide, so that figure says what the pipelining alone is worth)."""
    n = 1 << log2n
    keys0, vals0 = make_input(torch, n, kind, 0, device)
    sets = [(keys0.clone(), vals0.clone()) for _ in range(K + W)]
    out = {"pairs": n, "steps": K, "warmup": W}
    for depth in (1, 2):
        streams = [torch.cuda.Stream(device=device) for _ in range(depth)]
        sorters = [G.RadixSort(digit_bits=digit_bits) for _ in range(depth)]
        for srt in sorters:
            srt.prepare_internal_buffers(n)
        if depth > 1:
            for k, v in sets:
                k.copy_(keys0)
                v.copy_(vals0)

        def step(i):
            k, v = sets[i]
            sorters[i % depth].run_ptr(k.data_ptr(), v.data_ptr(), n, 0, streams[i % depth].cuda_stream)

        barrier()
        for i in range(W):
            step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(W, W + K):
            step(i)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        out["value_depth%d" % depth] = round(n * K / elapsed / 1e6, 1)
        out["ms_per_step_depth%d" % depth] = round(elapsed / K * 1e3, 4)
        del sorters
    out["verified"] = verify_sorted(torch, keys0, sets[-1][0], sets[-1][1], True)
    del sets
    torch.cuda.empty_cache()
    return out


def load_traffic(workload_key):
    """HBM bytes per scatter launch from committed rocprofv3 PMC passes (profiles/traffic_*.json), if they were
    collected for this exact workload; otherwise null."""
    try:
        best = None
        pdir = os.path.join(ROOT, "profiles")
        for f in sorted(os.listdir(pdir)):
            if f.startswith("traffic_") and f.endswith(".json"):
                d = json.load(open(os.path.join(pdir, f)))
                if d.get("workload_key") == workload_key:
                    best = d
                    best["_file"] = f
        return best
    except Exception:
        return None


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run ... bench.py <same arguments>` as a
    child process (this parent never touches a GPU and nothing is exec'ed over a process that did), forward rank 0's JSON
    line to stdout and return the child's exit code."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("[bench] launching %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "8")  # (torchrun would set 1 and say so; the CPU side of a rank is the launch loop)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for raw in child.stdout:
        text = raw.decode(errors="replace")
        try:
            if "metric" in json.loads(text):
                line = text.strip()
                continue
        except ValueError:
            pass
        sys.stderr.write(text)  # anything else a rank wrote to the real stdout
    rc = child.wait()
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    elif rc == 0:
        sys.stderr.write("[bench] the ranks exited with 0 but printed no JSON line\n")
        rc = 1
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    # stdout must carry exactly ONE line (the JSON): RCCL prints a version banner to the C-level stdout at exit, so
    # keep a private copy of the real stdout for the JSON and point fd 1 (and Python's sys.stdout) at stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libglu_hip has no CPU fallback")

    if args.rehearse_one_gpu:
        local_rank = 0
        if not os.environ.get("GLU_HIP_RCCL_LIB"):
            raise SystemExit("--rehearse-one-gpu needs GLU_HIP_RCCL_LIB=tests/cpp/bin/libmock_rccl.so (RCCL refuses two ranks on one GPU)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import glu_hip as G

    G.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29577")
        if args.rehearse_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    log2n = args.log2_keys if args.log2_keys is not None else (28 if world == 1 else 27)
    n = 1 << log2n
    K, W = args.steps, args.warmup
    # everything runs on one explicit torch stream: the handle of torch's default stream is 0, which the C ABI
    # reads as "use the library's own queue"
    work_stream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(work_stream)
    stream = work_stream.cuda_stream
    assert stream != 0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    result = {}

    def make_line(res, elapsed, units, workload, parallelism):
        line = {
            "metric": "Mkeys/s sorting 2^28 uint32 key+val; % HBM roofline; 1/2/4/8 GPU",
            "value": round(units / elapsed / 1e6, 1),
            "unit": "Mkeys/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": workload, "pairs_per_gpu": n, "key_distribution": args.keys,
                       "parallelism": parallelism, "device": G.device_info()},
        }
        line.update(res)
        return line

    if world == 1 and not args.force_dist:
        sorter = G.RadixSort(digit_bits=args.digit_bits)
        t_prep = time.perf_counter()
        sorter.prepare_internal_buffers(n)
        # (prepare places the two scratch arrays by measurement: glu_radix_sort_scratch_placement, include/glu_hip.h)
        result["scratch_placement"] = dict(sorter.scratch_placement(), prepare_s=round(time.perf_counter() - t_prep, 3))
        keys0, vals0 = make_input(torch, n, args.keys, 0, device)
        # one pristine copy per step so that nothing but the sort runs inside the timed region
        free_bytes = torch.cuda.mem_get_info()[0]
        copies = K + W
        restore_in_region = False
        if copies * n * 8 > free_bytes * 0.8:
            copies = max(1, int(free_bytes * 0.8) // (n * 8))
            restore_in_region = True
        sets = [(keys0.clone(), vals0.clone()) for _ in range(copies)]

        def step(i):
            k, v = sets[i % copies]
            if restore_in_region and i >= copies:
                k.copy_(keys0)
                v.copy_(vals0)
            sorter.run_ptr(k.data_ptr(), v.data_ptr(), n, 0, stream)

        for i in range(W):
            step(i)
        barrier()
        # LIGHT per-kernel events in the timed region: only around the kernels that move the data (the scatter of every pass
        # that is expected to run, the in-LDS pass) -- an event between two kernels costs the queue microseconds, and all 28 of
        # a sort that ends in LDS were 4 % of its time; count / scan kernel times come from three more sorts after the region
        sorter.set_profiling(False if args.no_kernel_events else (True if args.full_kernel_events else "light"))
        step_events = []
        t0 = time.perf_counter()
        for i in range(W, W + K):
            if not args.no_kernel_events:  # per-step device time (every step sorts a differently placed copy)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(work_stream)
                step(i)
                e1.record(work_stream)
                step_events.append((e0, e1))
            else:
                step(i)
        barrier()
        elapsed = time.perf_counter() - t0
        step_ms = sorted(a.elapsed_time(b) for a, b in step_events)
        prof = sorter.read_profile()
        sorter.set_profiling(False)
        detail = {"count_ms": 0.0, "scan_ms": 0.0, "passes": 0}
        if not args.no_kernel_events:  # (outside the timed region: every kernel boundary of three more sorts)
            sorter.set_profiling(True)
            for i in range(W + K, W + K + 3):
                k, v = sets[i % copies]
                k.copy_(keys0)
                v.copy_(vals0)
                sorter.run_ptr(k.data_ptr(), v.data_ptr(), n, 0, stream)
            torch.cuda.synchronize()
            detail = sorter.read_profile()
            sorter.set_profiling(False)
        bits = sorter.digit_bits
        units = n * K
        verified = None
        if not args.no_verify:
            k, v = sets[(W + K - 1) % copies]
            verified = verify_sorted(torch, keys0, k, v, True)
        # roofline of the dominant kernel: the scatter pass reads key+val and writes key+val of every pair
        passes = max(int(prof["passes"]), 1)
        scatter_ms = prof["scatter_ms"] / passes
        alg_bytes = n * 2 * (KEY_BYTES + VAL_BYTES)
        achieved = alg_bytes / (scatter_ms * 1e-3) / 1e9 if scatter_ms > 0 else 0.0
        workload_key = "radix_sort_u32_pairs_2^%d_%s_bits%d" % (log2n, args.keys, bits)
        traffic = load_traffic(workload_key)
        passes_per_sort = passes // K
        # bytes the sort really moves per pair: every pass reads and writes key + val in its scatter; a pass reads the keys
        # once more in its count kernel unless it is the second pass of a pair that took its count table from the first
        # pass's two-digit histogram (then the pair moves that table instead: 256 x workgroups x 512 B written and read)
        # (skipped[p]: 1 = an identity pass found by its count kernel, 2 = known before counting: no key read at all)
        # A sort that ended in LDS (include/glu_hip.h, glu_radix_sort_read_finish): two counting passes on the top 16 key bits
        # (one read of the keys for both tables) and one pass that orders every run of equal top bits inside LDS, reading and
        # writing each pair once; the profile above then holds those two passes, and the in-LDS pass on its own.
        fin = sorter.read_finish()
        ended_in_lds = bool(fin["accepted"])
        finish_ms = prof["finish_ms"] / max(int(prof["finish_passes"]), 1)
        if ended_in_lds:
            key_reads, scatters = 1, 2
            pair_table_bytes = 2 * 256 * 256 * 512 + 2 * 65536 * 4
            bytes_per_pair_moved = round(scatters * 2 * (KEY_BYTES + VAL_BYTES) + key_reads * KEY_BYTES + pair_table_bytes / n +
                                         2 * (KEY_BYTES + VAL_BYTES), 2)
        else:
            npl = 4 if fin["attempted"] else passes_per_sort  # (the plan describes the ordinary passes)
            skipped, alone, roles = sorter.read_plan(npl, roles=True)
            from_table = sum(1 for p in range(npl) if roles[p] == 2 and not alone[p] and skipped[p] != 2)
            key_reads = sum(1 for p in range(npl) if skipped[p] != 2) - from_table + (1 if fin["attempted"] else 0)
            scatters = sum(1 for p in range(npl) if not skipped[p])
            pair_table_bytes = sum(1 for p in range(npl) if roles[p] == 1 and skipped[p] != 2) * 2 * 256 * 256 * 512
            bytes_per_pair_moved = round(scatters * 2 * (KEY_BYTES + VAL_BYTES) + key_reads * KEY_BYTES + pair_table_bytes / n, 2)
        result.update({
            "roofline": {
                "bound": "hbm", "kernel": "radix_scatter_lines_kernel<u32,%d>" % bits,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
                # not a measurement of this run: the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this
                # same command) cannot run under the driver; the committed summary of the same workload is quoted
                "traffic_source": ("profiles/" + traffic["_file"]) if traffic else None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": round(scatter_ms, 4),
                "launches_timed": passes,
                "count_kernel_avg_ms": round(detail["count_ms"] / max(int(detail["passes"]), 1), 4),
                "scan_kernel_avg_ms": round(detail["scan_ms"] / max(int(detail["passes"]), 1), 4),
                "timing": "HIP events around every scatter launch and every in-LDS pass of the timed steps (on the sort's stream); "
                          "count / scan kernels: three more sorts after the timed region with events at every kernel boundary",
            },
            "whole_sort": {
                "passes": passes_per_sort, "digit_bits": bits,
                "count_kernels_reading_keys": key_reads, "scatter_passes_run": scatters,
                "ended_in_lds": ended_in_lds,
                "in_lds_pass": ({
                    "kernel": "radix_finish_bucket_kernel", "avg_launch_ms": round(finish_ms, 4),
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "achieved_GBps": round(alg_bytes / (finish_ms * 1e-3) / 1e9, 1) if finish_ms > 0 else None,
                    "frac_of_peak": round(alg_bytes / (finish_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if finish_ms > 0 else None,
                    "longest_run": fin["longest_run"], "capacity": fin["capacity"]} if ended_in_lds else None),
                "bytes_per_pair_moved": bytes_per_pair_moved,
                "achieved_GBps_own_bytes": round(units * bytes_per_pair_moved / elapsed / 1e9, 1),
                "frac_of_peak_own_bytes": round(units * bytes_per_pair_moved / elapsed / 1e9 / HBM_PEAK_GBPS, 4),
                # (what the reference's 8-pass structure would move, 160 B/pair, is priced in `reference_pass_structure` below,
                # for the sort that has that structure -- not here, where it would be bytes this sort does not move)
            },
            "verified": verified,
            "restore_copies_in_timed_region": restore_in_region,
        })
        if step_ms:
            # the spread is the physical placement of each copy's arrays in HBM, not noise
            result["step_device_ms"] = {"min": round(step_ms[0], 4), "median": round(step_ms[len(step_ms) // 2], 4),
                                        "max": round(step_ms[-1], 4)}
        # the same sort by the four ordinary passes (the attempt to end in LDS switched off), measured like the legs below:
        # warm-ups, then steps on restored inputs, device time per sort.  Reported next to the headline, not part of `value`.
        if ended_in_lds and not args.no_alt:
            os.environ["GLU_HIP_SORT_LDS_FINISH"] = "0"
            try:
                four = G.RadixSort(digit_bits=args.digit_bits)
            finally:
                del os.environ["GLU_HIP_SORT_LDS_FINISH"]
            four.prepare_internal_buffers(n)
            f_warm, f_steps = max(W, 3), max(K, 10)
            f_events = []
            for i in range(f_warm + f_steps):
                sets[i % copies][0].copy_(keys0)
                sets[i % copies][1].copy_(vals0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(work_stream)
                four.run_ptr(sets[i % copies][0].data_ptr(), sets[i % copies][1].data_ptr(), n, 0, stream)
                e1.record(work_stream)
                if i >= f_warm:
                    f_events.append((e0, e1))
            barrier()
            f_ms = sorted(a.elapsed_time(b) for a, b in f_events)
            f_verified = None
            if not args.no_verify:
                k, v = sets[(f_warm + f_steps - 1) % copies]
                f_verified = verify_sorted(torch, keys0, k, v, True)
            result["four_pass_sort"] = {
                "what": "the same sort with GLU_HIP_SORT_LDS_FINISH=0: four 8-bit counting passes, 72.5 B/pair",
                "steps": f_steps, "warmup": f_warm, "ms_per_step": round(f_ms[len(f_ms) // 2], 4), "ms_per_step_min": round(f_ms[0], 4),
                "timing": "device time per sort (HIP events), median and min",
                "value": round(n / (f_ms[len(f_ms) // 2] * 1e-3) / 1e6, 1), "unit": "Mkeys/s",
                "scratch_placement": four.scratch_placement(), "verified": f_verified,
            }
            four.destroy()
            del four
        # the same sort with the reference's pass structure (8 x 4-bit digits, 160 B/pair), measured like the headline:
        # warm-ups, then K timed steps on restored inputs (restores outside the per-step device timing), median + min.
        # Reported next to the headline, not part of `value`.
        if bits != 4 and not args.no_alt:
            alt = G.RadixSort(digit_bits=4)
            alt.prepare_internal_buffers(n)
            alt_warm, alt_steps = max(W, 3), max(K, 10)

            def restore(i):
                sets[i % copies][0].copy_(keys0)
                sets[i % copies][1].copy_(vals0)

            for i in range(alt_warm):
                restore(i)
                alt.run_ptr(sets[i % copies][0].data_ptr(), sets[i % copies][1].data_ptr(), n, 0, stream)
            barrier()
            alt.set_profiling(True)
            alt_events = []
            for i in range(alt_steps):
                restore(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(work_stream)
                alt.run_ptr(sets[i % copies][0].data_ptr(), sets[i % copies][1].data_ptr(), n, 0, stream)
                e1.record(work_stream)
                alt_events.append((e0, e1))
            barrier()
            alt_ms = sorted(a.elapsed_time(b) for a, b in alt_events)
            ap = alt.read_profile()
            alt.set_profiling(False)
            alt_verified = None
            if not args.no_verify:
                k, v = sets[(alt_steps - 1) % copies]
                alt_verified = verify_sorted(torch, keys0, k, v, True)
            a_scatter_ms = ap["scatter_ms"] / max(int(ap["passes"]), 1)
            a_med, a_min = alt_ms[len(alt_ms) // 2], alt_ms[0]
            # bytes this sort really moved per pair (second passes of pairs that took their table from the first pass's
            # two-digit histogram did not read the keys again; a 4-bit pair's tables are 4 MiB + 256 KiB, written and read)
            a_passes = int(ap["passes"]) // alt_steps
            a_skipped, a_alone, a_roles = alt.read_plan(a_passes, roles=True)
            a_from_table = sum(1 for p in range(a_passes) if a_roles[p] == 2 and not a_alone[p] and a_skipped[p] != 2)
            a_key_reads = sum(1 for p in range(a_passes) if a_skipped[p] != 2) - a_from_table
            a_scatters = sum(1 for p in range(a_passes) if not a_skipped[p])
            a_tables = sum(1 for p in range(a_passes) if a_roles[p] == 1 and a_skipped[p] != 2) * 2 * (256 * 16 * 1024 + 16 * 256 * 16 * 4)
            a_moved = round(a_scatters * 2 * (KEY_BYTES + VAL_BYTES) + a_key_reads * KEY_BYTES + a_tables / n, 2)
            result["reference_pass_structure"] = {
                "digit_bits": 4, "passes": int(ap["passes"]) // alt_steps, "steps": alt_steps, "warmup": alt_warm,
                "ms_per_step": round(a_med, 4), "ms_per_step_min": round(a_min, 4), "timing": "device time per sort (HIP events), median and min",
                "value": round(n / (a_med * 1e-3) / 1e6, 1), "unit": "Mkeys/s",
                "achieved_GBps_at_160B_per_pair": round(n * 160 / (a_med * 1e-3) / 1e9, 1),
                "frac_of_peak_at_160B_per_pair": round(n * 160 / (a_med * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "frac_of_peak_at_160B_per_pair_best": round(n * 160 / (a_min * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "count_kernels_reading_keys": a_key_reads,
                "bytes_per_pair_moved": a_moved,
                "frac_of_peak_own_bytes": round(n * a_moved / (a_med * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "scatter_kernel": "radix_scatter_lines_kernel<u32,4>",
                "scatter_kernel_avg_ms": round(a_scatter_ms, 4),
                "scatter_kernel_frac_of_peak": round(alg_bytes / (a_scatter_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if a_scatter_ms > 0 else None,
                "count_kernel_avg_ms": round(ap["count_ms"] / max(int(ap["passes"]), 1), 4),
                "verified": alt_verified,
            }
        workload = "2^%d uint32 key + uint32 val pairs, %s keys, vals=iota, in-place stable radix sort, 1x MI355X" % (
            log2n, "uniform-random full-range" if args.keys == "uniform" else "all-zero")
        parallelism = "single"
    else:
        from glu_hip import dist as D

        keys0, vals0 = make_input(torch, n, args.keys, rank, device, index_base=rank * n, as_rank_of=args.as_rank_of if args.force_dist else 1)

        def sharded(native):
            """The whole N > 1 measurement over one transport (native = glu_dist_* inside libglu_hip.so with its own RCCL
            communicator; otherwise torch.distributed collectives around the same C-ABI device work).  Collective."""
            res = {}
            hang = os.environ.get("GLU_BENCH_TEST_NATIVE_HANG")  # tests of the watchdog below: this rank never arrives
            if native and hang is not None and int(hang) == rank:
                time.sleep(1e6)
            def run_depth(depth, rounds=None):
                """W warm-up sorts, then K timed sorts with `depth` sorts in flight (depth 1 = strictly one after the other, the
                same regime as the N = 1 line; depth 2 = consecutive independent sorts on two streams / buffer sets /
                communicators, so that the exchange of sort i+1 can run under the local sort of sort i).  rounds: rounds of
                the exchange (None = the library's choice: 3 at depth 1 for shards of 2^24 pairs and more, 1 at depth 2)."""
                dsort = D.DistributedRadixSort(slots=depth, profile=not args.no_kernel_events,
                                               native=native, rounds=rounds)
                if args.digit_bits is not None:
                    for srt in dsort.local_sorters():
                        srt.set_digit_bits(args.digit_bits)
                if dsort.native and depth > 1 and args.reserved_cus:
                    for slot in dsort._slots:  # leave CUs to the RCCL kernels of the other sort in flight
                        slot["native"].set_reserved_cus(args.reserved_cus)
                for i in range(W):
                    dsort.sort_async(keys0, vals0)
                barrier()
                dsort.phase_times()  # drop the warm-up stamps
                sorters = dsort.local_sorters()
                for srt in sorters:
                    srt.set_profiling("light" if not args.no_kernel_events else False)  # (events around the scatter / in-LDS kernels only)
                t0 = time.perf_counter()
                handle = None
                for i in range(K):
                    handle = dsort.sort_async(keys0, vals0)
                barrier()
                dt = time.perf_counter() - t0
                profs = [srt.read_profile() for srt in sorters]
                for srt in sorters:
                    srt.set_profiling(False)
                return {"dsort": dsort, "elapsed": dt, "handle": handle, "profs": profs, "sorters": sorters,
                        "phases": dsort.phase_times()}

            def max_over_ranks(x):
                tt = torch.tensor([x], dtype=torch.float64, device=device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                return float(tt.item())

            # Two timed regions of K sorts each.  depth 1: one sort at a time on one stream (partition -> exchange -> local sort,
            # each waiting for the one before) -- the regime of the N = 1 line, and what `value` / `ms_per_step` report for every N.
            # depth 2: the K sorts alternate between two glu_dist objects (own stream, buffers and communicator), so that the
            # exchange of sort i + 1 runs under the local sort of sort i -- how a caller with a stream of independent batches
            # uses the API; reported beside the headline as `value_depth2`.  The per-kernel roofline numbers come from depth 1.
            r1 = run_depth(1)
            elapsed1 = max_over_ranks(r1["elapsed"])
            res["value_depth1"] = round(n * world * K / elapsed1 / 1e6, 1)
            res["ms_per_step_depth1"] = round(elapsed1 / K * 1e3, 4)
            if r1["dsort"].native:
                # how the exchange was posted: in rounds (groups of buckets travel while the groups that have arrived are sorted:
                # the library's default for one sort at a time on more than one rank) -- and, beside it, the same K sorts with
                # ONE grouped exchange, then the local sort (what rounds 2 and 3 measured)
                res["exchange_rounds"] = r1["dsort"]._slots[0]["native"].last_rounds()
                if res["exchange_rounds"] > 1:
                    r1b = run_depth(1, rounds=1)
                    e1b = max_over_ranks(r1b["elapsed"])
                    res["value_depth1_one_round"] = round(n * world * K / e1b / 1e6, 1)
                    res["ms_per_step_depth1_one_round"] = round(e1b / K * 1e3, 4)
                    res["phases_ms_rank0_one_round"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r1b["phases"].items()}
                    # `value` is the BETTER of the two ways to post the exchange (both stay on the line): whether rounds pay over a
                    # fabric has only ever been costed on paper -- no run of this repository has had more than one GPU
                    one_round_elapsed = r1b["elapsed"] if e1b < elapsed1 else None
                    res["value_is"] = ("one grouped exchange (value_depth1_one_round)" if e1b < elapsed1
                                       else "the exchange in %d rounds (value_depth1)" % res["exchange_rounds"])
                    del r1b
            depth = max(1, args.pipeline_depth)
            elapsed = r1["elapsed"]
            if res.get("value_is", "").startswith("one grouped"):
                elapsed = one_round_elapsed
            if depth > 1:
                r2 = run_depth(depth)
                res["value_depth%d" % depth] = round(n * world * K / max_over_ranks(r2["elapsed"]) / 1e6, 1)
                res["ms_per_step_depth%d" % depth] = round(max_over_ranks(r2["elapsed"]) / K * 1e3, 4)
                res["phases_ms_rank0_depth%d" % depth] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r2["phases"].items()}
                del r2
            res["pipeline_depth"] = depth
            dsort, handle, profs, sorters = r1["dsort"], r1["handle"], r1["profs"], r1["sorters"]
            res["native_c_abi"] = bool(dsort.native)
            res["local_sort"] = (dsort._slots[0]["native"].last_local_sort() if dsort.native else dsort.last_local_sort)
            # did rank 0's segmented local sort end in LDS (one counting pass + one in-LDS pass instead of three passes)?
            res["local_sort_in_lds_rank0"] = sorters[0].read_seg_finish()
            if args.rehearse_one_gpu:
                res["rehearsal"] = "NOT A MEASUREMENT: %d ranks share one GPU and exchange through files (tests/cpp/mock_rccl.cpp)" % world
            rk, rv, cnt = handle.synchronize()
            # rank 0's view: the scatter kernel (1 partition launch over n pairs + 4 sort launches over its shard per sort)
            # and the device time of every phase of a sort
            launches = sum(int(pf["passes"]) for pf in profs)
            scatter_ms = sum(pf["scatter_ms"] for pf in profs) / max(launches, 1)
            per_sort = max(launches // max(K, 1), 1)
            alg_bytes = 2 * (KEY_BYTES + VAL_BYTES) * (n + (per_sort - 1) * int(cnt)) // per_sort
            if launches and scatter_ms > 0:
                achieved = alg_bytes / (scatter_ms * 1e-3) / 1e9
                res["roofline"] = {
                    "bound": "hbm", "kernel": "radix_scatter_lines_kernel<u32,%d> (rank 0)" % sorters[0].digit_bits,
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                    "traffic": None, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(scatter_ms, 4),
                    "launches_timed": launches,
                    "count_kernel_avg_ms": round(sum(pf["count_ms"] for pf in profs) / launches, 4),
                }
            phases = r1["phases"]
            res["phases_ms_rank0"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in phases.items()}
            units = n * world * K
            verified = None
            if not args.no_verify:
                flipped = rk ^ (-2**31)
                ok = bool((flipped[1:] >= flipped[:-1]).all())
                # rank boundaries: my last key <= next rank's first key; counts add up
                first = flipped[:1].to(torch.int64) if cnt > 0 else torch.full((1,), 2**40, device=device)
                last = flipped[-1:].to(torch.int64) if cnt > 0 else torch.full((1,), -2**40, device=device)
                edges = torch.stack([first, last]).reshape(1, 2)
                gathered = [torch.zeros_like(edges) for _ in range(world)]
                dist.all_gather(gathered, edges)
                total = torch.tensor([cnt], dtype=torch.int64, device=device)
                dist.all_reduce(total)
                if rank == 0:
                    prev_last = None
                    for e in gathered:
                        f, l = int(e[0, 0]), int(e[0, 1])
                        if f > l:
                            continue  # empty shard
                        if prev_last is not None and f < prev_last:
                            ok = False
                        prev_last = l
                ok = ok and int(total.item()) == n * world
                # the VALUES: (1) the output pairs of all ranks are the input pairs of all ranks (two all-reduced sums over pair
                # hashes: every value is still with its key, nothing lost, nothing doubled); (2) equal keys keep ascending
                # values -- the values are global input indices, so that IS the stable order -- inside a rank and across the
                # boundaries between ranks
                fp = pair_fingerprint(torch, rk[:cnt], rv[:cnt]) - pair_fingerprint(torch, keys0, vals0)
                dist.all_reduce(fp)
                ok = ok and bool((fp == 0).all())
                uv = rv[:cnt].to(torch.int64) & 0xFFFFFFFF
                eq = rk[1:cnt] == rk[:cnt - 1]
                ok = ok and bool((uv[1:][eq] > uv[:-1][eq]).all())
                ends = torch.zeros(1, 4, dtype=torch.int64, device=device)  # first key, first value, last key, last value
                if cnt > 0:
                    ends[0, 0], ends[0, 1], ends[0, 2], ends[0, 3] = flipped[0], uv[0], flipped[cnt - 1], uv[cnt - 1]
                ends_all = [torch.zeros_like(ends) for _ in range(world)]
                dist.all_gather(ends_all, ends)
                counts_all = [torch.zeros_like(total) for _ in range(world)]
                dist.all_gather(counts_all, torch.tensor([cnt], dtype=torch.int64, device=device))
                prev = None
                for e, c in zip(ends_all, counts_all):
                    if int(c.item()) == 0:
                        continue
                    if prev is not None and int(e[0, 0]) == int(prev[0, 2]) and not int(e[0, 1]) > int(prev[0, 3]):
                        ok = False  # the same key on both sides of a rank boundary, values not ascending
                    prev = e
                flag = torch.tensor([1 if ok else 0], device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                verified = bool(flag.item())
            res["verified"] = verified
            res["verified_what"] = ("keys ascending inside and across ranks; counts add up; the output pairs of all ranks are the input "
                                    "pairs of all ranks (all-reduced sums of pair hashes); equal keys keep ascending values (= input "
                                    "order) inside a rank and across rank boundaries")
            res["shard_pairs_rank0"] = int(cnt)
            workload = ("2^%d uint32 key+val pairs per GPU (%d GPUs, 2^%.2f total), uniform-random keys; top-8-bit bucket "
                        "partition + one RCCL all-to-all over xGMI + local sort") % (log2n, world, log2n + __import__("math").log2(world))
            rounds_used = res.get("exchange_rounds", 1)
            parallelism = "bucket-sharded x%d (%s per sort), one sort at a time" % (
                world, "1 grouped RCCL exchange" if rounds_used <= 1 else "the exchange in %d rounds of grouped RCCL sends / receives, "
                "each group of buckets sorted behind its own round" % rounds_used)
            if depth > 1:
                parallelism += " (value_depth%d: %d independent sorts in flight)" % (depth, depth)

            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            del r1, dsort, handle, profs, sorters, rk, rv
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            return res, float(t.item()), units, workload, parallelism

        # The native transport has only ever met more than one rank through a test double (no multi-GPU box reaches the
        # development loop), so the line must not depend on it: the torch.distributed transport is measured first and kept as
        # the fallback line; a native run that raises, or does not finish by the deadline, makes rank 0 print that line
        # (with `native_error`) and every rank leave.
        want_native = args.rehearse_one_gpu or args.transport == "native"
        fallback = None
        bail_lock = threading.Lock()
        state = {"armed": False}

        def bail(reason):
            with bail_lock:
                if not state["armed"]:
                    return
                state["armed"] = False
                if rank == 0 and fallback is not None:
                    fallback["native_error"] = reason
                    os.write(json_fd, (json.dumps(fallback) + "\n").encode())
                sys.stderr.write("[bench rank %d] native transport abandoned: %s\n" % (rank, reason))
                sys.stderr.flush()
                try:  # where every thread of this rank stands (a hang is diagnosed from the launcher's log)
                    import faulthandler

                    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                except Exception:
                    pass
                os._exit(0)

        watchdog = None
        if want_native and not args.no_transport_fallback:
            t0 = time.perf_counter()
            fres, felapsed, funits, fworkload, fpar = sharded(False)
            took = time.perf_counter() - t0
            if rank == 0:
                fallback = make_line(fres, felapsed, funits, fworkload, fpar)
            deadline = float(os.environ.get("GLU_BENCH_NATIVE_DEADLINE_S", "0")) or (120.0 + 5.0 * took)
            state["armed"] = True
            watchdog = threading.Timer(deadline, bail, args=("did not finish within %.0f s" % deadline,))
            watchdog.daemon = True
            watchdog.start()
        try:
            result, elapsed, units, workload, parallelism = sharded(want_native)
        except BaseException as e:  # noqa: a rank that fails alone must not take the launcher down before rank 0 has printed
            if watchdog is None:
                raise
            bail("%s: %s" % (type(e).__name__, e))  # (waits for a bail the watchdog has begun: that one ends the process)
            raise
        with bail_lock:
            state["armed"] = False
        if watchdog is not None:
            watchdog.cancel()
        if not args.no_one_gpu:
            def max_over(x):
                tt = torch.tensor([x], dtype=torch.float64, device=device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                return float(tt.item())

            one_log2 = args.one_gpu_log2 if args.one_gpu_log2 is not None else (min(28, log2n + 1) if args.rehearse_one_gpu else 28)
            one = one_gpu_figures(torch, G, device, one_log2, K, W, args.keys, args.digit_bits, barrier, max_over)
            result["one_gpu"] = one
            for d in (1, 2):
                if "value_depth%d" % d in result:
                    result["speedup_vs_1gpu_depth%d" % d] = round(result["value_depth%d" % d] / one["value_depth%d" % d], 3)
        if fallback is not None:
            result["torch_transport"] = {k: fallback[k] for k in ("value", "ms_per_step", "value_depth1", "ms_per_step_depth1", "verified",
                                                                  "local_sort", "phases_ms_rank0") if k in fallback}

    if rank == 0:
        line = make_line(result, elapsed, units, workload, parallelism)
        if world == 1 and not args.force_dist and not args.no_cpu_baseline:
            if args.cpu_sample_log2 is None:
                # the same input array as the GPU sorted, whole, one run per configuration
                kh = keys0.cpu().numpy().view("uint32")
                vh = vals0.cpu().numpy().view("uint32")
                line["cpu_baseline"] = cpu_baseline(None, kh, vh)
            else:
                line["cpu_baseline"] = cpu_baseline(args.cpu_sample_log2)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
