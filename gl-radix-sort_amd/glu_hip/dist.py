"""Multi-GPU radix sort: one process per GPU, one all-to-all over xGMI on the top key bits, then local sorts.

The reference (loryruta/gl-radix-sort) is single-device; this is the sharded form BASELINE.json config 4 asks
for.  Rank r holds slice r of the global array.  Steps:

  1. each rank: stable partition of its slice by the top-8-bit bucket (one counting pass of the local sort
     kernels, glu_radix_sort_partition_ptr) + the 256-bin bucket histogram;
  2. all-gather of the R x 256 histogram -> every rank derives the same contiguous bucket -> rank assignment;
  3. ONE grouped exchange of keys and values (RCCL: ncclSend / ncclRecv to and from every peer inside one
     ncclGroupStart / ncclGroupEnd), receive segments ordered by source rank;
  4. each rank: local stable sort of what it received -- the segmented sort of the low 24 bits per bucket (one segmented
     pass + an in-LDS pass where the runs fit, else three segmented passes)
     (glu_radix_sort_run_segments_ptr: the shard arrives as one message per source rank, each grouped by bucket, and the
     first pass regroups it), or the ordinary sort of all 32 bits for small / fragmented shards.

Two transports.  `native` = the whole sort inside libglu_hip.so (glu_dist_* of include/glu_hip.h: the library makes its
own RCCL communicator from a unique id that is broadcast through the torch process group; torch only owns the tensors):
what bench.py --gpus N runs, opted into with native=True or GLU_HIP_DIST_NATIVE=1 -- it has only ever met one-rank RCCL
communicators and a file-based test double, never more than one real GPU.  Default = the torch.distributed transport below
(all_gather + all_to_all_single under "nccl", point-to-point under gloo) around the same C-ABI device work; it is also what
the CPU tests drive with an oracle-backed stand-in for the device work.

The concatenation of the rank outputs in rank order equals the single-device stable sort: a bucket is never
split across ranks, equal keys share a bucket, the local partition and the local sort are stable and receive
segments keep (source rank, source index) order.

The planning functions are pure numpy so the rank logic is testable on CPU with simulated ranks; the device
work goes through `local_ops` (HipLocalOps: libglu_hip.so on torch device memory; there is no CPU version in
the product -- the CPU tests inject an oracle-backed stand-in).
"""
import numpy as np

TOP_BITS = 8
NUM_BUCKETS = 1 << TOP_BITS


def plan_bucket_to_rank(bucket_totals, world_size):
    """Contiguous, monotone bucket -> rank map balancing the element counts.

    bucket_totals: [NUM_BUCKETS] global counts.  Rank r gets buckets [cut[r], cut[r+1]) where cut[r] is the
    bucket boundary whose prefix count is nearest to r * N / R.  A bucket is never split (a single hot bucket
    bounds the balance).  Returns int64 array [NUM_BUCKETS] of rank ids.
    """
    totals = np.asarray(bucket_totals, dtype=np.int64)
    assert totals.shape == (NUM_BUCKETS,)
    n = int(totals.sum())
    prefix = np.concatenate([[0], np.cumsum(totals)])  # prefix[b] = elements in buckets < b
    cuts = [0]
    for r in range(1, world_size):
        target = (n * r) // world_size
        b = int(np.searchsorted(prefix, target, side="left"))
        # choose the nearer boundary of b-1 / b
        if b > 0 and abs(int(prefix[b - 1]) - target) <= abs(int(prefix[min(b, NUM_BUCKETS)]) - target):
            b -= 1
        b = max(b, cuts[-1])
        cuts.append(min(b, NUM_BUCKETS))
    cuts.append(NUM_BUCKETS)
    out = np.empty(NUM_BUCKETS, dtype=np.int64)
    for r in range(world_size):
        out[cuts[r]:cuts[r + 1]] = r
    return out


def split_counts(all_hist, bucket_to_rank, rank):
    """all_hist: [R, NUM_BUCKETS] per-rank bucket histograms.  Returns (send_counts[R], recv_counts[R]) for
    `rank`: send_counts[d] = my elements whose bucket belongs to rank d; recv_counts[s] = rank s's elements
    whose bucket belongs to me."""
    all_hist = np.asarray(all_hist, dtype=np.int64)
    world = all_hist.shape[0]
    owner = np.asarray(bucket_to_rank)
    send = np.array([int(all_hist[rank][owner == d].sum()) for d in range(world)], dtype=np.int64)
    recv = np.array([int(all_hist[s][owner == rank].sum()) for s in range(world)], dtype=np.int64)
    return send, recv


def shard_pieces(all_hist, bucket_to_rank, rank):
    """The shard of `rank` as it arrives: one message per source rank, each holding the rank's buckets in ascending
    order.  Returns (begin, length, segment) per piece in (source, bucket) order and the number of segments (= buckets
    the rank owns): a segment's pieces are in source order, the stable order of its elements."""
    all_hist = np.asarray(all_hist, dtype=np.int64)
    mine = np.nonzero(np.asarray(bucket_to_rank) == rank)[0]
    if mine.size == 0:
        z = np.zeros(0, dtype=np.uint64)
        return z, z, np.zeros(0, dtype=np.uint32), 0
    g0, g1 = int(mine[0]), int(mine[-1]) + 1
    lens = all_hist[:, g0:g1].reshape(-1)
    begin = np.concatenate([[0], np.cumsum(lens)[:-1]])
    seg = np.tile(np.arange(g1 - g0, dtype=np.uint32), all_hist.shape[0])
    keep = lens > 0
    return begin[keep].astype(np.uint64), lens[keep].astype(np.uint64), seg[keep], g1 - g0


class _DeviceMemory:
    """`count` int32 words of device memory that somebody else owns, for torch.as_tensor (no copy)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<i4", "data": (ptr, False), "version": 2}


def _device_array(torch, ptr, count, device):
    if count == 0 or not ptr:
        return torch.empty(0, dtype=torch.int32, device=device)
    return torch.as_tensor(_DeviceMemory(ptr, count), device=device)


class HipLocalOps:
    """Device work of one rank on torch tensors (int32 views of the uint32 data) through libglu_hip.so."""

    def __init__(self, digit_bits=None):
        import torch
        from . import RadixSort

        if not torch.cuda.is_available():
            raise RuntimeError("HipLocalOps needs an MI355X: there is no CPU fallback")
        self.torch = torch
        self.sorter = RadixSort(digit_bits=digit_bits)

    def _stream(self):
        # NOTE: the handle of torch's default stream is 0, which the C ABI reads as "use the library queue";
        # DistributedRadixSort therefore always runs inside its own (non-default) torch stream.
        h = self.torch.cuda.current_stream().cuda_stream
        if h == 0:
            raise RuntimeError("HipLocalOps must run under a non-default torch stream (torch.cuda.stream(...))")
        return h

    def prepare(self, count):
        self.sorter.prepare_internal_buffers(count)

    def partition(self, keys, vals, out_keys, out_vals, hist):
        """Stable partition by (key >> 24); hist: int32[256] device tensor."""
        n = keys.numel()
        if n == 0:  # an empty local slice: torch hands out data_ptr() == 0, nothing to partition
            hist.zero_()
            return
        self.sorter.partition_ptr(keys.data_ptr(), vals.data_ptr(), out_keys.data_ptr(), out_vals.data_ptr(), n,
                                  32 - TOP_BITS, TOP_BITS, hist.data_ptr(), self._stream())

    def sort(self, keys, vals, count):
        if count == 0:  # a rank that owns no bucket with elements (skewed keys) receives nothing
            return
        self.sorter.run_ptr(keys.data_ptr(), vals.data_ptr(), count, 0, self._stream())

    def sort_segments(self, in_keys, in_vals, out_keys, out_vals, count, begin, length, seg, nseg, key_bits):
        """The shard's pieces (source-major, grouped by bucket) -> buckets in order, each sorted by its low key_bits bits."""
        self.sorter.run_segments_ptr(in_keys.data_ptr(), in_vals.data_ptr(), out_keys.data_ptr(), out_vals.data_ptr(), count,
                                     begin, length, seg, nseg, key_bits, self._stream())


class SortHandle:
    """Result of DistributedRadixSort.sort_async: this rank's shard (views into the slot's receive buffers, valid until
    the slot is used again, i.e. for `slots` further sort_async calls).  wait() makes the caller's current stream wait
    for the sort; synchronize() blocks the host."""

    def __init__(self, owner, slot, keys, vals, count, event):
        self._owner, self.slot, self.keys, self.vals, self.count, self._event = owner, slot, keys, vals, count, event

    def wait(self):
        if self._event is not None:
            self._owner.torch.cuda.current_stream(self.keys.device).wait_event(self._event)
        return self.keys, self.vals, self.count

    def synchronize(self):
        if self._event is not None:
            self._event.synchronize()
        return self.keys, self.vals, self.count


class DistributedRadixSort:
    """sort(keys, vals) -> (sorted_keys, sorted_vals, count): this rank's shard of the globally sorted array
    (shard sizes vary with the data).  keys / vals: 1-D int32 torch tensors holding uint32 bit patterns.

    slots > 1 lets consecutive sorts overlap: sort_async() runs each sort on its own stream with its own buffers and
    device-op object, so the all-to-all of sort i+1 (RCCL, few CUs) can run under the local sort of sort i.  The order
    of collectives is the call order on every rank, so ranks must issue the same sequence of sorts."""

    def __init__(self, group=None, local_ops=None, capacity_factor=1.25, slots=1, local_ops_factory=None, profile=False,
                 native=None, rounds=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        # native: the whole sort inside the C library (glu_dist_*).  Opt-in (argument or GLU_HIP_DIST_NATIVE=1): it has never
        # run over more than one real GPU; bench.py opts in and verifies its output.  (native=True works under any backend:
        # the rehearsal of bench.py, where the ranks share one GPU, the torch group is gloo and GLU_HIP_RCCL_LIB names the
        # test transport.)
        import os

        if native is None:
            native = os.environ.get("GLU_HIP_DIST_NATIVE", "0") not in ("", "0")
        self.native = bool(native) and local_ops is None and local_ops_factory is None and torch.cuda.is_available()
        self.capacity_factor = capacity_factor
        self.segmented_min = 1 << 24  # torch transport: shards from this size up take the segmented local sort
        self.last_local_sort = None
        if self.native:
            made = []
            for _ in range(max(1, slots)):
                nd = self._make_native(profile)  # collective: every rank gets an object or every rank gets None
                if nd is None:
                    break
                made.append(nd)
            if len(made) == max(1, slots):
                self._slots = [{"ops": None, "native": nd, "bufs": None, "stream": None} for nd in made]
                # The exchange in rounds (groups of buckets travel while the groups that have arrived are sorted) shortens ONE
                # sort and costs compute; with several sorts in flight the exchange of one hides under the local sort of
                # another anyway, so those take one round.  (The same on every rank: part of the message sequence.)
                if rounds is None and slots > 1:
                    rounds = 1
                if rounds is not None:
                    for nd in made:
                        nd.set_rounds(rounds)
            else:
                import warnings

                for nd in made:
                    nd.destroy()
                warnings.warn("glu_dist is not available on every rank (%s): using the torch.distributed transport"
                              % (self.native_error,))
                self.native = False
        if not self.native:
            if local_ops_factory is None:
                local_ops_factory = (lambda: local_ops) if local_ops is not None else HipLocalOps
            self._slots = [{"ops": local_ops_factory(), "bufs": None, "stream": None} for _ in range(max(1, slots))]
        self._next_slot = 0
        self.last_plan = None
        # profile=True: device-time stamps around the phases of every sort (events on the slot's stream), summed by
        # phase_times(); they are what tells a slow exchange from a slow local sort in a multi-GPU run
        self.profile = profile
        self._stamps = []

    def _all_agree(self, ok):
        """True iff `ok` holds on every rank (one all_gather_object: works under gloo and nccl alike)."""
        flags = [None] * self.world
        self.dist.all_gather_object(flags, bool(ok), group=self.group)
        return all(flags)

    def _make_native(self, profile):
        """One glu_dist per slot: rank 0 draws the RCCL unique id, the torch process group carries it to the others.
        Every step's outcome is agreed between the ranks before the next one, so that either every rank has an object or
        every rank falls back to the torch transport (nobody is left inside ncclCommInitRank or a broadcast)."""
        from . import Dist, dist_available, dist_unique_id

        self.native_error = None
        uid, err = None, None
        if self.rank == 0:
            try:
                uid = dist_unique_id()  # dlopens RCCL
            except Exception as e:
                err = str(e)
        ids = [(uid, err)]
        src = self.dist.get_global_rank(self.group, 0) if self.group is not None else 0
        self.dist.broadcast_object_list(ids, src=src, group=self.group)  # rank 0 always broadcasts, success or not
        uid, err = ids[0]
        mine = err
        if mine is None and self.rank != 0:
            try:
                dist_available()  # can this rank load RCCL at all?  (dlopen + symbols only)
            except Exception as e:
                mine = str(e)
        if not self._all_agree(mine is None):
            self.native_error = mine or "another rank cannot load RCCL"
            return None
        d = None
        try:
            d = Dist(uid, self.world, self.rank)  # ncclCommInitRank: collective; every rank has RCCL by now
            d.set_profiling(profile)
        except Exception as e:
            mine = str(e)
        if not self._all_agree(d is not None):
            self.native_error = mine or "glu_dist_create failed on another rank"
            if d is not None:
                d.destroy()
            return None
        return d

    def local_sorters(self):
        """The local RadixSort object of every slot (kernel-level profiling in bench.py)."""
        return [s["native"].sorter if self.native else s["ops"].sorter for s in self._slots]

    @property
    def ops(self):
        return self._slots[0]["ops"]

    def _buffers(self, slot, n_local, device, need_recv=0):
        cap = max(int(n_local * self.capacity_factor) + 4096, need_recv)
        b = slot["bufs"]
        if b is None or b["n_local"] < n_local or b["cap"] < cap or b["device"] != device:
            t = self.torch
            b = {
                "n_local": n_local, "cap": cap, "device": device,
                "part_k": t.empty(max(n_local, 1), dtype=t.int32, device=device),
                "part_v": t.empty(max(n_local, 1), dtype=t.int32, device=device),
                "recv_k": t.empty(cap, dtype=t.int32, device=device),
                "recv_v": t.empty(cap, dtype=t.int32, device=device),
                "out_k": None, "out_v": None,  # output arrays of the segmented local sort (allocated on first use)
                "hist": t.zeros(NUM_BUCKETS, dtype=t.int32, device=device),
                "all_hist": t.zeros(self.world * NUM_BUCKETS, dtype=t.int32, device=device),
            }
            slot["bufs"] = b
            if hasattr(slot["ops"], "prepare"):
                slot["ops"].prepare(cap)
        return b

    def _all_to_all(self, out, inp, recv_counts, send_counts):
        dist = self.dist
        backend = dist.get_backend(self.group)
        if backend == "gloo":
            # gloo has no all_to_all_single: same exchange as pairwise send/recv (CPU tests; the one-GPU rehearsal of
            # bench.py).  Device tensors are staged through host copies: gloo's point-to-point ops hand the tensor's data
            # pointer to the socket layer without any stream ordering.
            on_device = inp.is_cuda
            if on_device:
                self.torch.cuda.current_stream(inp.device).synchronize()
            src = inp.cpu() if on_device else inp
            dst = self.torch.empty(out.shape, dtype=out.dtype, device="cpu") if on_device else out
            reqs = []
            so = np.concatenate([[0], np.cumsum(send_counts)])
            ro = np.concatenate([[0], np.cumsum(recv_counts)])
            for peer in range(self.world):
                if peer == self.rank:
                    dst[ro[peer]:ro[peer + 1]].copy_(src[so[peer]:so[peer + 1]])
                    continue
                if send_counts[peer]:
                    reqs.append(dist.isend(src[so[peer]:so[peer + 1]].contiguous(), peer, group=self.group))
                if recv_counts[peer]:
                    reqs.append(dist.irecv(dst[ro[peer]:ro[peer + 1]], peer, group=self.group))
            for r in reqs:
                r.wait()
            if on_device:
                out.copy_(dst)
        else:
            dist.all_to_all_single(out, inp, [int(c) for c in recv_counts], [int(c) for c in send_counts],
                                   group=self.group)

    def sort_async(self, keys, vals):
        """Enqueues one distributed sort on the next slot's private stream (ordered after the caller's current stream)
        and returns a SortHandle.  The host blocks only for the bucket-histogram exchange."""
        slot = self._slots[self._next_slot]
        index = self._next_slot
        self._next_slot = (self._next_slot + 1) % len(self._slots)
        if not keys.is_cuda:
            k, v, c = self._sort(slot, keys, vals)
            return SortHandle(self, index, k, v, c, None)
        t = self.torch
        if slot["stream"] is None:
            slot["stream"] = t.cuda.Stream(device=keys.device)
        slot["stream"].wait_stream(t.cuda.current_stream(keys.device))
        with t.cuda.stream(slot["stream"]):
            k, v, c = self._sort(slot, keys, vals)
            event = t.cuda.Event()
            event.record(slot["stream"])
        return SortHandle(self, index, k, v, c, event)

    def sort(self, keys, vals):
        """One sort, complete (stream-wise) when it returns: the caller's current stream waits for it."""
        return self.sort_async(keys, vals).wait()

    def _stamp(self, marks, device):
        if self.profile and device.type == "cuda":
            e = self.torch.cuda.Event(enable_timing=True)
            e.record(self.torch.cuda.current_stream(device))
            marks.append(e)

    def phase_times(self, reset=True):
        """Average device milliseconds per sort of each phase (profile=True; call after synchronising)."""
        names = ("partition", "histogram_exchange_and_plan", "all_to_all", "local_sort")
        if self.native:
            per = [s["native"].phase_times() for s in self._slots]
            count = sum(p["sorts"] for p in per)
            return {"sorts": count, **{n: (sum(p[n] * p["sorts"] for p in per) / count if count else 0.0) for n in names}}
        sums, count = [0.0] * len(names), 0
        for marks in self._stamps:
            if len(marks) != len(names) + 1:
                continue
            for i in range(len(names)):
                sums[i] += marks[i].elapsed_time(marks[i + 1])
            count += 1
        if reset:
            self._stamps = []
        return {"sorts": count, **{n: (sums[i] / count if count else 0.0) for i, n in enumerate(names)}}

    def _sort_native(self, slot, keys, vals):
        """The sort inside libglu_hip.so (glu_dist_sort_ptr): partition + histogram exchange + plan, the exchange (in rounds
        for large shards) and the local sort, into the receive arrays of the glu_dist object -- which glu_dist_prepare has
        PLACED by measurement together with the send-side arrays and the sorter's scratch (where large arrays lie to each other
        decides between discrete speeds of every pass on this memory system; arrays from torch's allocator cannot be chosen).
        The returned tensors alias those arrays (valid until the slot is used again)."""
        t = self.torch
        nd = slot["native"]
        n_local = keys.numel()
        stream = t.cuda.current_stream(keys.device).cuda_stream
        b = slot["bufs"]
        if b is None or b["n_local"] < n_local:
            try:
                nd.prepare(n_local, int(n_local * self.capacity_factor) + 4096)
            except Exception:
                pass  # (a rank that cannot allocate says so inside the sort, to every rank: the failure is collective there)
            slot["bufs"] = {"n_local": n_local}
        k_ptr, v_ptr, n_recv = nd.sort_ptr(keys.data_ptr() if n_local else 0, vals.data_ptr() if n_local else 0, n_local, stream)
        return (_device_array(t, k_ptr, n_recv, keys.device), _device_array(t, v_ptr, n_recv, keys.device), n_recv)

    def _sort(self, slot, keys, vals):
        if self.native and keys.is_cuda:
            return self._sort_native(slot, keys, vals)
        t, dist = self.torch, self.dist
        ops = slot["ops"]
        n_local = keys.numel()
        b = self._buffers(slot, n_local, keys.device)
        marks = []
        self._stamp(marks, keys.device)

        # 1. local stable partition by top-8-bit bucket + histogram
        ops.partition(keys, vals, b["part_k"], b["part_v"], b["hist"])
        self._stamp(marks, keys.device)

        # 2. everyone learns every rank's histogram (R x 256 int32: latency-bound, tiny)
        dist.all_gather_into_tensor(b["all_hist"], b["hist"], group=self.group)
        all_hist = b["all_hist"].cpu().numpy().reshape(self.world, NUM_BUCKETS).astype(np.int64)

        # 3. identical plan on every rank
        owner = plan_bucket_to_rank(all_hist.sum(axis=0), self.world)
        send_counts, recv_counts = split_counts(all_hist, owner, self.rank)
        n_recv = int(recv_counts.sum())
        if n_recv > b["cap"]:
            b = self._grow_recv(slot, b, n_recv)
        self.last_plan = {"owner": owner, "send": send_counts, "recv": recv_counts}
        self._stamp(marks, keys.device)

        # 4. keys and values to every peer (this transport: gloo point-to-point or two torch all-to-alls; the native path
        #    posts one grouped RCCL exchange); receive segments ordered by source rank
        recv_k = b["recv_k"][:n_recv]
        recv_v = b["recv_v"][:n_recv]
        self._all_to_all(recv_k, b["part_k"][:n_local], recv_counts, send_counts)
        self._all_to_all(recv_v, b["part_v"][:n_local], recv_counts, send_counts)
        self._stamp(marks, keys.device)

        # 5. local stable sort of the received pairs: segmented (three passes over the low 24 bits per bucket, the first one
        #    regrouping the source-major shard) when the device ops offer it and the shard is large and not fragmented
        self.last_local_sort = "ordinary"
        if hasattr(ops, "sort_segments") and n_recv >= self.segmented_min:
            begin, length, seg, nseg = shard_pieces(all_hist, owner, self.rank)
            if 0 < begin.size <= 1024:
                if b["out_k"] is None or b["out_k"].numel() < b["cap"]:
                    b["out_k"] = t.empty(b["cap"], dtype=t.int32, device=b["device"])
                    b["out_v"] = t.empty(b["cap"], dtype=t.int32, device=b["device"])
                out_k, out_v = b["out_k"][:n_recv], b["out_v"][:n_recv]
                ops.sort_segments(recv_k, recv_v, out_k, out_v, n_recv, begin, length, seg, nseg, 32 - TOP_BITS)
                self.last_local_sort = "segmented"
                recv_k, recv_v = out_k, out_v
        if self.last_local_sort == "ordinary":
            ops.sort(recv_k, recv_v, n_recv)
        self._stamp(marks, keys.device)
        if marks:
            self._stamps.append(marks)
        return recv_k, recv_v, n_recv

    def _grow_recv(self, slot, b, n_recv):
        t = self.torch
        cap = int(n_recv * 1.1) + 4096
        b["recv_k"] = t.empty(cap, dtype=t.int32, device=b["device"])
        b["recv_v"] = t.empty(cap, dtype=t.int32, device=b["device"])
        b["out_k"] = b["out_v"] = None
        b["cap"] = cap
        if hasattr(slot["ops"], "prepare"):
            slot["ops"].prepare(cap)
        return b
