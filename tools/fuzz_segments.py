"""Randomized parity run of the segmented sort (glu_radix_sort_run_segments_ptr) against numpy: random piece layouts (source-major
shards, pieces in any address order, empty and tiny pieces, one hot segment), key distributions, key_bits and sizes around
the switch points (2^16: gather + per-segment sorts below, segmented passes from there up).  Time-boxed, not part of the suite.
usage (GPU box): python tools/fuzz_segments.py [seconds] [seed]"""
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gl-radix-sort_amd"))
import torch
import glu_hip as G

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
print("seed", seed, flush=True)
sorter = G.RadixSort()
stream = torch.cuda.Stream()
t0, cases, failures = time.time(), 0, 0
while time.time() - t0 < budget:
    r = rng.random()
    if r < 0.25:
        n = int(rng.integers(0, 70000))
    elif r < 0.5:
        n = int(rng.choice([65535, 65536, 65537, 10240 * 7, 256 * 10240 + 5])) + int(rng.integers(0, 3))
    else:
        n = int(2 ** rng.uniform(16, 22.5))
    nseg = int(rng.choice([1, 2, 7, 32, 64, 256]))
    sources = int(rng.choice([1, 2, 3, 8]))
    kind = int(rng.integers(0, 5))
    weights = np.ones(nseg)
    if kind == 1:
        weights[rng.integers(0, nseg)] = 50 * nseg  # one hot segment
    if kind == 2:
        weights[rng.random(nseg) < 0.5] = 0          # empty segments
        if not weights.any():
            weights[0] = 1
    if kind == 3:
        weights = rng.random(nseg) ** 6 + 1e-9       # many tiny segments
    weights = weights / weights.sum()
    per_source = np.diff(np.linspace(0, n, sources + 1).astype(np.int64))
    lens = np.concatenate([rng.multinomial(int(p), weights) for p in per_source]) if n else np.zeros(sources * nseg, np.int64)
    begin = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    seg = np.tile(np.arange(nseg), sources)
    if kind == 4:  # pieces listed in another order than their addresses
        perm = rng.permutation(lens.size)
        begin, lens, seg = begin[perm], lens[perm], seg[perm]
    kk = int(rng.integers(0, 4))
    keys = rng.integers(0, 2 ** 32, n, dtype=np.uint32)
    if kk == 1:
        keys &= np.uint32(0x00FF00FF)
    if kk == 2:
        keys = rng.integers(0, 4, n, dtype=np.uint32) * np.uint32(0x00010101)
    if kk == 3:
        keys = np.sort(keys)[::-1].copy()
    vals = np.arange(n, dtype=np.uint32)
    key_bits = int(rng.choice([0, 8, 16, 24, 24, 24, 32]))
    kin = torch.from_numpy(keys.view(np.int32).copy()).cuda()
    vin = torch.from_numpy(vals.view(np.int32).copy()).cuda()
    kout = torch.full((max(n, 1),), -1, dtype=torch.int32, device="cuda")
    vout = torch.full((max(n, 1),), -1, dtype=torch.int32, device="cuda")
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        sorter.run_segments_ptr(kin.data_ptr(), vin.data_ptr(), kout.data_ptr(), vout.data_ptr(), n, begin, lens, seg, nseg, key_bits,
                                stream.cuda_stream)
        gk, gv = kout[:n].cpu().numpy().view(np.uint32), vout[:n].cpu().numpy().view(np.uint32)
    ek, ev = [], []
    mask = np.uint32((1 << key_bits) - 1) if key_bits < 32 else np.uint32(0xFFFFFFFF)
    for g in range(nseg):
        idx = np.nonzero(seg == g)[0]
        if idx.size == 0:
            continue
        k = np.concatenate([keys[begin[i]:begin[i] + lens[i]] for i in idx])
        v = np.concatenate([vals[begin[i]:begin[i] + lens[i]] for i in idx])
        if key_bits:
            o = np.argsort(k & mask, kind="stable")
            k, v = k[o], v[o]
        ek.append(k)
        ev.append(v)
    ek = np.concatenate(ek) if ek else np.zeros(0, np.uint32)
    ev = np.concatenate(ev) if ev else np.zeros(0, np.uint32)
    ok = (gk == ek).all() and (gv == ev).all()
    cases += 1
    if not ok:
        failures += 1
        print("FAIL n=%d nseg=%d sources=%d kind=%d kk=%d key_bits=%d" % (n, nseg, sources, kind, kk, key_bits), flush=True)
print("cases %d, failures %d" % (cases, failures))
sys.exit(1 if failures else 0)
