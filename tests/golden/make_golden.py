"""Regenerates the golden fixtures in this directory.

  reference_vectors.json   known-answer vectors held by the reference's own tests, transcribed as data
                           (inputs + expected outputs; file:line of each in the "source" fields), plus the
                           std::minstd_rand conformance values the reference's seeded inputs depend on.
  oracle_checksums.json    FNV-1a checksums of the oracle's output (sorted keys, stably permuted iota values)
                           for every input size of the reference's RadixSort tests, and the scanned block-count
                           tables of one small case -- produced by oracle/glu_oracle.c; they pin the oracle
                           against regressions and let a GPU test compare without re-running it.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402


def fnv1a64(a):
    """FNV-1a over the little-endian bytes of a uint32 array (vectorised per byte lane is not possible for
    FNV, so this runs in Python over bytes of small arrays only)."""
    h = 0xCBF29CE484222325
    for b in np.ascontiguousarray(a, dtype=np.uint32).tobytes():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


REFERENCE_VECTORS = {
    "minstd_rand": {
        "source": "test/util/Random.hpp:15-39 (std::minstd_rand; seed 0 -> default seed 1)",
        "seed": 1, "first": [48271, 182605794, 1291394886, 1914720637], "value_10000": 399268537,
    },
    "blelloch_scan_simple": {
        "source": "test/blelloch_scan_tests.cpp:12-26",
        "data_type": 3, "input": [1, 2, 3, 4, 5, 6, 7, 8], "expected": [0, 1, 3, 6, 10, 15, 21, 28],
    },
    "reduce_simple_uint": {
        "source": "test/reduce_tests.cpp:14-52",
        "data_type": 3,
        "input": [32, 35, 1, 3, 95, 10, 22, 24, 44, 37, 7, 80, 33, 54, 46, 23, 14, 84, 11, 67,
                  4, 58, 70, 61, 16, 36, 83, 9, 56, 99, 28, 98, 69, 21, 51, 34, 48, 91, 62, 19,
                  59, 79, 39, 92, 97, 78, 52, 40, 66, 47, 89, 88, 74, 49, 31, 20, 45, 13, 26, 72,
                  43, 30, 65, 94, 63, 8, 60, 15, 93, 86, 41, 75, 12, 73, 55, 90, 64, 96, 53, 1,
                  57, 71, 50, 42, 29, 2, 77, 25, 82, 18, 81, 85, 27, 5, 6, 68, 17, 38, 87, 76],
        "cases": [
            {"op": 0, "count": 100, "expected": 4951},
            {"op": 1, "count": 5, "expected": 319200},
            {"op": 2, "count": 100, "expected": 1},
            {"op": 3, "count": 100, "expected": 99},
        ],
    },
    "reduce_all": {
        "source": "test/reduce_tests.cpp:54-145 (op = sum; float tolerances are the reference's WithinAbs 0.1)",
        "cases": [
            {"data_type": 3, "input": [1, 11, 80, 73, 48, 40, 89, 36, 70, 57], "expected": [505], "abs_tol": 0},
            {"data_type": 0, "input": [42.138, 18.228, -19.127, 86.564, 11.904, 48.538, 30.606, 11.338, -32.699, -29.587],
             "expected": [167.9], "abs_tol": 0.1},
            {"data_type": 1, "input": [-6.20, -56.02, 49.42, 52.38, -23.81, -29.72, 95.46, 77.37, -85.00, 81.74],
             "expected": [155.6], "abs_tol": 0.1},
            {"data_type": 4, "input": [-77.08, 19.54, 98.89, -16.09, 10.53, 91.17, 43.06, -94.18, -19.18, 0.86,
                                       -49.99, -92.53, -4.68, 42.34, 2.79, -4.26, -17.49, 43.99, 79.45, -14.58],
             "expected": [66.29, -23.75], "abs_tol": 0.1},
            {"data_type": 5, "input": [-17.04, 1.79, 82.67, 39.72, 52.66, 24.75, -19.05, 91.92, 19.15, 44.93, -52.13, 18.85,
                                       -84.25, 69.53, -11.43, 33.17, 19.46, -14.30, -15.20, -63.83, -20.51, -56.75, -2.70, 82.66,
                                       3.86, 55.48, -12.37, -11.02, -30.62, -67.54, -29.89, -77.30, -21.55, 50.46, 39.34, 81.08,
                                       -56.40, 84.61, 90.26, 13.35],
             "expected": [-135.24, 192.97, 69.49, 208.59], "abs_tol": 0.1},
            {"data_type": 10, "input": [-38, -88, 57, -34, 61, 60, -90, 73, -23, -17, 34, -79, -80, 53, 24, -23, -88, 69, -83, -67],
             "expected": [-226, -53], "abs_tol": 0},
            {"data_type": 11, "input": [-95, 99, -30, 2, -69, 33, 78, 20, 33, -43, -38, -26, 69, -67, -17, -57, 18, -23, -2, -53,
                                        88, -96, 40, -48, -93, -47, -91, 59, -89, 82, 10, 94, -15, 7, 41, 14, 63, 53, -40, 53],
             "expected": [-90, -2, -49, 58], "abs_tol": 0},
        ],
    },
    "radix_sort_tests": {
        "source": "test/radix_sort_tests.cpp:88-158 (seed 1; assertions: keys sorted + permutation of the input)",
        "cases": [{"n": n, "min": 0, "max": 4294967295} for n in (128, 256, 512, 1024)]
                 + [{"n": 2048, "min": 0, "max": 10}]
                 + [{"n": n, "min": 0, "max": 4294967295}
                    for n in (10993, 14978, 16243, 18985, 23857, 27865, 33363, 41298, 45821, 47487)],
    },
    "blelloch_scan_tests": {
        "source": "test/blelloch_scan_tests.cpp:28-82 (seed 123, values in [0,100); assertion: == std::exclusive_scan)",
        "sizes": [1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576],
        "partitions": {"count": 1024, "num_partitions": [1, 32, 100, 1000]},
    },
    "reduce_size_tests": {
        "source": "test/reduce_tests.cpp:147-183 (seed 1, values in [0,100); assertion: == std::accumulate mod 2^32)",
        "fitting": [32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072],
        "non_fitting": [1, 31, 93, 201, 693, 2087, 7358, 88289, 345897, 6094798, 5238082, 10043898],
    },
}


def write_cpp_header(path):
    """The same known-answer vectors as C++ data for tests/cpp (generated: edit REFERENCE_VECTORS, not the header)."""
    ctype = {0: "float", 1: "double", 2: "int32_t", 3: "uint32_t", 4: "float", 5: "float", 10: "int32_t", 11: "int32_t"}
    comps = {0: 1, 1: 1, 2: 1, 3: 1, 4: 2, 5: 4, 10: 2, 11: 4}

    def lit(x, t):
        if t == "float":
            return repr(float(x)) + "f"
        if t == "double":
            return repr(float(x))
        return str(int(x)) + ("u" if t == "uint32_t" else "")

    out = ["// GENERATED by tests/golden/make_golden.py from the known-answer vectors of the reference's tests",
           "// (tests/golden/reference_vectors.json) -- data only.", "#pragma once", "#include <cstdint>", "#include <vector>", "",
           "namespace golden", "{"]
    g = REFERENCE_VECTORS["reduce_simple_uint"]
    out.append("    inline const std::vector<uint32_t> k_reduce_simple_input{%s};" % ", ".join(str(v) + "u" for v in g["input"]))
    out.append("    struct SimpleCase { int op; size_t count; uint32_t expected; };")
    out.append("    inline const std::vector<SimpleCase> k_reduce_simple_cases{%s};" % ", ".join(
        "{%d, %d, %du}" % (c["op"], c["count"], c["expected"]) for c in g["cases"]))
    out.append("    struct TypedCase { int data_type; int components; std::vector<double> input; std::vector<double> expected; double abs_tol; };")
    rows = []
    for c in REFERENCE_VECTORS["reduce_all"]["cases"]:
        rows.append("        {%d, %d, {%s}, {%s}, %r}" % (c["data_type"], comps[c["data_type"]], ", ".join(repr(float(v)) for v in c["input"]),
                                                           ", ".join(repr(float(v)) for v in c["expected"]), float(c["abs_tol"])))
    out.append("    inline const std::vector<TypedCase> k_reduce_all_cases{\n%s};" % ",\n".join(rows))
    s = REFERENCE_VECTORS["blelloch_scan_simple"]
    out.append("    inline const std::vector<uint32_t> k_scan_simple_input{%s};" % ", ".join(str(v) + "u" for v in s["input"]))
    out.append("    inline const std::vector<uint32_t> k_scan_simple_expected{%s};" % ", ".join(str(v) + "u" for v in s["expected"]))
    r = REFERENCE_VECTORS["reduce_size_tests"]
    out.append("    inline const std::vector<size_t> k_reduce_fitting_sizes{%s};" % ", ".join(map(str, r["fitting"])))
    out.append("    inline const std::vector<size_t> k_reduce_non_fitting_sizes{%s};" % ", ".join(map(str, r["non_fitting"])))
    b = REFERENCE_VECTORS["blelloch_scan_tests"]
    out.append("    inline const std::vector<size_t> k_scan_sizes{%s};" % ", ".join(map(str, b["sizes"])))
    out.append("    inline const std::vector<size_t> k_scan_partition_counts{%s};" % ", ".join(map(str, b["partitions"]["num_partitions"])))
    rs = REFERENCE_VECTORS["radix_sort_tests"]["cases"]
    out.append("    struct SortCase { size_t n; uint32_t min; uint32_t max; };")
    out.append("    inline const std::vector<SortCase> k_radix_sort_cases{%s};" % ", ".join("{%d, %du, %du}" % (c["n"], c["min"], c["max"]) for c in rs))
    out.append("} // namespace golden")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


def main():
    with open(os.path.join(HERE, "reference_vectors.json"), "w") as f:
        json.dump(REFERENCE_VECTORS, f, indent=1)
    write_cpp_header(os.path.join(ROOT, "tests", "cpp", "util", "golden_vectors.hpp"))

    sums = {"source": "oracle/glu_oracle.c glu_oracle_radix_sort_reference on the reference's test inputs "
                      "(keys = minstd_rand seed 1, vals = iota)", "cases": []}
    for case in REFERENCE_VECTORS["radix_sort_tests"]["cases"]:
        n = case["n"]
        keys = O.minstd_sample(1, n, case["min"], case["max"])
        vals = np.arange(n, dtype=np.uint32)
        res = O.radix_sort_reference(keys, vals)
        sums["cases"].append({"n": n, "min": case["min"], "max": case["max"],
                              "sorted_keys_fnv1a64": fnv1a64(res["result_keys"]),
                              "sorted_vals_fnv1a64": fnv1a64(res["result_vals"])})
    # per-pass scanned tables of one small case (N = 3001 -> 3 blocks, nbp2 = 4)
    n = 3001
    keys = O.minstd_sample(1, n, 0, 0xFFFFFFFF)
    res = O.radix_sort_reference(keys, np.arange(n, dtype=np.uint32), trace=True)
    sums["trace_n3001"] = {"tables": res["tables"].tolist(),
                           "sorted_keys_fnv1a64": fnv1a64(res["result_keys"]),
                           "sorted_vals_fnv1a64": fnv1a64(res["result_vals"])}
    # num_steps quirk (RadixSort.hpp:286-287,331-332): odd num_steps leaves the result in scratch
    res = O.radix_sort_reference(keys, np.arange(n, dtype=np.uint32), num_steps=3)
    sums["num_steps_3_n3001"] = {"result_in_scratch": res["result_in_scratch"], "passes": res["passes"],
                                 "result_keys_fnv1a64": fnv1a64(res["result_keys"]),
                                 "result_vals_fnv1a64": fnv1a64(res["result_vals"]),
                                 "user_keys_fnv1a64": fnv1a64(res["user_keys"])}
    with open(os.path.join(HERE, "oracle_checksums.json"), "w") as f:
        json.dump(sums, f, indent=1)
    print("wrote reference_vectors.json, oracle_checksums.json")


if __name__ == "__main__":
    main()
