// C++ API tests of glu::RadixSort -- the reference's cases (reference test/radix_sort_tests.cpp:88-158: same
// sizes, seed and key ranges, same sorted + permutation assertions) plus what the reference leaves unpinned:
// value order (stability), key bit 31, num_steps, tiny counts.
#include <algorithm>
#include <cinttypes>
#include <numeric>
#include <random>
#include <thread>
#include <unordered_map>
#include <vector>

#include "glu/RadixSort.hpp"
#include "util/golden_vectors.hpp"
#include "util/minstd_inputs.hpp"
#include "util/mini_test.hpp"

using namespace glu;

namespace
{
    template<typename T>
    std::unordered_map<T, size_t> value_histogram(const std::vector<T>& v)
    {
        std::unordered_map<T, size_t> h;
        for (const T& x : v) ++h[x];
        return h;
    }

    template<typename T>
    void check_permutation(const std::vector<T>& a, const std::vector<T>& b)
    {
        CHECK(a.size() == b.size());
        CHECK(value_histogram(a) == value_histogram(b));
    }

    template<typename T>
    void check_sorted(const std::vector<T>& v)
    {
        CHECK(std::is_sorted(v.begin(), v.end()));
    }

    /// keys sorted + permutation, as the reference asserts
    void run_reference_case(size_t n, GLuint min, GLuint max)
    {
        std::vector<GLuint> keys = test_inputs::minstd_vector<GLuint>(1, n, min, max);
        std::vector<GLuint> vals(n);

        ShaderStorageBuffer key_buffer(keys);
        ShaderStorageBuffer val_buffer(vals);

        RadixSort radix_sort;
        radix_sort(key_buffer.handle(), val_buffer.handle(), keys.size());

        std::vector<GLuint> sorted_keys = key_buffer.get_data<GLuint>();
        check_permutation(keys, sorted_keys);
        check_sorted(sorted_keys);
    }

    /// full contract: identical to std::stable_sort of the pairs by (masked) key
    void run_stability_case(std::vector<GLuint> keys, size_t num_steps, uint32_t digit_bits)
    {
        const size_t n = keys.size();
        std::vector<GLuint> vals(n);
        std::iota(vals.begin(), vals.end(), 0u);

        std::vector<GLuint> order(vals);
        const uint32_t mask = (num_steps == 0 || num_steps >= 8) ? 0xFFFFFFFFu : ((1u << (4 * num_steps)) - 1u);
        std::stable_sort(order.begin(), order.end(), [&](GLuint a, GLuint b) { return (keys[a] & mask) < (keys[b] & mask); });

        ShaderStorageBuffer key_buffer(n ? n * sizeof(GLuint) : 4), val_buffer(n ? n * sizeof(GLuint) : 4);
        if (n)
        {
            key_buffer.write_data(keys.data(), n * sizeof(GLuint));
            val_buffer.write_data(vals.data(), n * sizeof(GLuint));
        }
        RadixSort radix_sort;
        radix_sort.set_digit_bits(digit_bits);
        radix_sort(key_buffer.handle(), val_buffer.handle(), n, num_steps);

        std::vector<GLuint> out_keys = key_buffer.get_data<GLuint>();
        std::vector<GLuint> out_vals = val_buffer.get_data<GLuint>();
        bool same = true;
        for (size_t i = 0; i < n; i++) same = same && out_vals[i] == order[i] && out_keys[i] == keys[order[i]];
        CHECK(same);
    }
} // namespace

TEST_CASE("integer-helpers-like-the-reference")
{
    // glu/gl_utils.hpp:279-302 (values the reference's own arithmetic gives, including its edge cases at 0)
    CHECK(glu::next_power_of_2<uint32_t>(0) == 0u);
    CHECK(glu::next_power_of_2<uint32_t>(1) == 1u);
    CHECK(glu::next_power_of_2<uint32_t>(3) == 4u);
    CHECK(glu::next_power_of_2<uint32_t>(1024) == 1024u);
    CHECK(glu::next_power_of_2<uint32_t>(1025) == 2048u);
    CHECK(glu::is_power_of_2<uint32_t>(0));
    CHECK(glu::is_power_of_2<uint32_t>(64) && !glu::is_power_of_2<uint32_t>(65));
    CHECK(glu::div_ceil<size_t>(1025, 1024) == 2 && glu::div_ceil<size_t>(1024, 1024) == 1 && glu::div_ceil<size_t>(0, 1024) == 0);
}

TEST_CASE("RadixSort-reference-cases")
{
    // RadixSort-128-256-512-1024, RadixSort-2048 and RadixSort-multiple-sizes of the reference
    for (const golden::SortCase& c : golden::k_radix_sort_cases) run_reference_case(c.n, c.min, c.max);
}

TEST_CASE("RadixSort-stable-full-32-bit")
{
    std::mt19937 gen(0x5EED);
    for (uint32_t bits : {4u, 8u})
        for (size_t n : {0, 1, 2, 3, 1023, 1024, 1025, 4096, 4097, 65537, 1 << 20})
        {
            std::vector<GLuint> keys(n);
            for (auto& k : keys) k = gen(); // sets bit 31
            run_stability_case(keys, 0, bits);
        }
}

TEST_CASE("RadixSort-stable-duplicates")
{
    for (uint32_t bits : {4u, 8u})
    {
        run_stability_case(test_inputs::minstd_vector<GLuint>(7, 50000, 0, 10), 0, bits);
        run_stability_case(std::vector<GLuint>(30000, 0u), 0, bits);           // the reference's benchmark input
        run_stability_case(std::vector<GLuint>(30000, 0xFFFFFFFFu), 0, bits);
        std::vector<GLuint> asc(20000), desc(20000);
        std::iota(asc.begin(), asc.end(), 0u);
        for (size_t i = 0; i < desc.size(); i++) desc[i] = GLuint(desc.size() - i) * 77777u;
        run_stability_case(asc, 0, bits);
        run_stability_case(desc, 0, bits);
    }
}

TEST_CASE("RadixSort-num-steps")
{
    std::mt19937 gen(99);
    std::vector<GLuint> keys(30011);
    for (auto& k : keys) k = gen();
    for (uint32_t bits : {4u, 8u})
        for (size_t steps = 1; steps <= 9; steps++) run_stability_case(keys, steps, bits);
}

TEST_CASE("RadixSort-reuse-and-prepare")
{
    RadixSort radix_sort;
    radix_sort.prepare_internal_buffers(100000);
    std::mt19937 gen(5);
    for (size_t n : {100000, 5000, 77777})
    {
        std::vector<GLuint> keys(n), vals(n);
        for (auto& k : keys) k = gen();
        std::iota(vals.begin(), vals.end(), 0u);
        ShaderStorageBuffer kb(keys), vb(vals);
        radix_sort(kb.handle(), vb.handle(), n);
        std::vector<GLuint> sk = kb.get_data<GLuint>(), sv = vb.get_data<GLuint>();
        check_sorted(sk);
        bool paired = true;
        for (size_t i = 0; i < n; i++) paired = paired && keys[sv[i]] == sk[i];
        CHECK(paired);
    }
}

TEST_CASE("RadixSort-options-on-the-object")
{
    // switches are set on the object (glu_radix_sort_set_option), not through the process environment: the same sort by the small
    // geometry, without the one-workgroup path, with 4-bit digits -- the results are the same, an unknown name is an error status
    std::mt19937 gen(6);
    const size_t n = 30000;
    std::vector<GLuint> keys(n), vals(n);
    for (auto& k : keys) k = gen();
    std::iota(vals.begin(), vals.end(), 0u);
    std::vector<GLuint> want_k, want_v;
    for (int variant = 0; variant < 3; variant++)
    {
        RadixSort radix_sort;
        if (variant == 1) radix_sort.set_option("SORT_SMALL", 1), radix_sort.set_option("GLU_HIP_SORT_NO_SINGLE_BLOCK", 1);
        if (variant == 2) radix_sort.set_option("digit_bits", 4);
        ShaderStorageBuffer kb(keys), vb(vals);
        radix_sort(kb.handle(), vb.handle(), n);
        std::vector<GLuint> sk = kb.get_data<GLuint>(), sv = vb.get_data<GLuint>();
        check_sorted(sk);
        if (variant == 0) want_k = sk, want_v = sv;
        CHECK(sk == want_k);
        CHECK(sv == want_v);
        if (variant == 2) CHECK(radix_sort.digit_bits() == 4u);
    }
    glu_radix_sort raw = nullptr;
    CHECK(glu_radix_sort_create(&raw) == GLU_OK);
    CHECK(glu_radix_sort_set_option(raw, "NO_SUCH_SWITCH", 1) == GLU_ERROR_INVALID_ARGUMENT);
    CHECK(glu_radix_sort_destroy(raw) == GLU_OK);
}

TEST_CASE("RadixSort-u64-keys")
{
    // 64-bit keys + 32-bit values (BASELINE config 5; not in the reference): same stable contract
    std::mt19937_64 gen(0xC0FFEE);
    for (size_t n : {2, 5000, 70001, 3 * (1 << 20) + 17})
    {
        std::vector<uint64_t> keys(n);
        for (auto& k : keys) k = gen() >> (gen() % 3 == 0 ? 40 : 0); // mix of wide and narrow keys, duplicates in the narrow ones
        std::vector<GLuint> vals(n);
        std::iota(vals.begin(), vals.end(), 0u);
        std::vector<GLuint> order(vals);
        std::stable_sort(order.begin(), order.end(), [&](GLuint a, GLuint b) { return keys[a] < keys[b]; });

        ShaderStorageBuffer key_buffer(keys), val_buffer(vals);
        RadixSort radix_sort;
        radix_sort.prepare_internal_buffers_u64(n);
        radix_sort.sort_u64(key_buffer.handle(), val_buffer.handle(), n);
        std::vector<uint64_t> out_keys = key_buffer.get_data<uint64_t>();
        std::vector<GLuint> out_vals = val_buffer.get_data<GLuint>();
        bool same = true;
        for (size_t i = 0; i < n; i++) same = same && out_vals[i] == order[i] && out_keys[i] == keys[order[i]];
        CHECK(same);
    }
}

TEST_CASE("RadixSort-keys-only")
{
    // the reference's README snippet sorts a single buffer; its header needs a dummy value buffer -- here both work
    for (size_t n : {100, 20000, 4 * (1 << 20) + 1})
    {
        std::vector<GLuint> keys = test_inputs::minstd_vector<GLuint>(5, n, 0, UINT32_MAX);
        for (size_t i = 0; i < n; i += 3) keys[i] ^= 0x80000000u;
        ShaderStorageBuffer key_buffer(keys);
        RadixSort radix_sort;
        radix_sort.sort_keys(key_buffer.handle(), n);
        std::vector<GLuint> expected(keys);
        std::sort(expected.begin(), expected.end());
        CHECK(key_buffer.get_data<GLuint>() == expected);
    }
}

TEST_CASE("RadixSort-typed-keys")
{
    std::mt19937 gen(77);
    const size_t n = 100001;
    std::vector<float> keys(n);
    for (auto& k : keys) k = float(int(gen() % 20001) - 10000) * 0.25f;
    std::vector<GLuint> vals(n);
    std::iota(vals.begin(), vals.end(), 0u);
    std::vector<GLuint> order(vals);
    std::stable_sort(order.begin(), order.end(), [&](GLuint a, GLuint b) { return keys[a] < keys[b]; });
    ShaderStorageBuffer kb(keys), vb(vals);
    RadixSort radix_sort;
    radix_sort.sort_typed(static_cast<float*>(kb.device_ptr()), static_cast<uint32_t*>(vb.device_ptr()), n);
    std::vector<float> out_keys = kb.get_data<float>();
    std::vector<GLuint> out_vals = vb.get_data<GLuint>();
    bool same = true;
    for (size_t i = 0; i < n; i++) same = same && out_vals[i] == order[i] && out_keys[i] == keys[order[i]];
    CHECK(same);

    std::vector<int64_t> ikeys(n);
    for (auto& k : ikeys) k = int64_t(gen()) * int64_t(gen() % 2 ? -1 : 1) * 65537;
    std::vector<int64_t> expected(ikeys);
    std::sort(expected.begin(), expected.end());
    ShaderStorageBuffer ib(ikeys);
    radix_sort.sort_typed(static_cast<int64_t*>(ib.device_ptr()), static_cast<uint32_t*>(nullptr), n);
    CHECK(ib.get_data<int64_t>() == expected);
}

TEST_CASE("RadixSort-bit-range")
{
    // keys that fit 24 bits sort in three 8-bit passes; a sort by the middle bits is stable in everything else
    std::mt19937 gen(5);
    const size_t n = 150001;
    std::vector<GLuint> keys(n), vals(n);
    for (auto& k : keys) k = gen();
    std::iota(vals.begin(), vals.end(), 0u);
    for (auto range : {std::pair<uint32_t, uint32_t>{0, 24}, {8, 20}, {24, 32}})
    {
        const uint32_t width = range.second - range.first;
        auto field = [&](GLuint k) { return (k >> range.first) & ((width == 32 ? 0u : (1u << width)) - 1u); };
        std::vector<GLuint> order(vals);
        std::stable_sort(order.begin(), order.end(), [&](GLuint a, GLuint b) { return field(keys[a]) < field(keys[b]); });
        ShaderStorageBuffer kb(keys), vb(vals);
        RadixSort radix_sort;
        radix_sort.sort_bit_range(static_cast<uint32_t*>(kb.device_ptr()), static_cast<uint32_t*>(vb.device_ptr()), n, range.first,
                                  range.second);
        std::vector<GLuint> out_keys = kb.get_data<GLuint>(), out_vals = vb.get_data<GLuint>();
        bool same = true;
        for (size_t i = 0; i < n; i++) same = same && out_vals[i] == order[i] && out_keys[i] == keys[order[i]];
        CHECK(same);
    }
}

TEST_CASE("RadixSort-raw-pointer-overload")
{
    // native callers: raw device pointers (here taken from ShaderStorageBuffer) on the library queue
    std::mt19937 gen(31);
    const size_t n = 200003;
    std::vector<GLuint> keys(n), vals(n);
    for (auto& k : keys) k = gen();
    std::iota(vals.begin(), vals.end(), 0u);
    ShaderStorageBuffer kb(keys), vb(vals);
    RadixSort radix_sort;
    radix_sort(static_cast<uint32_t*>(kb.device_ptr()), static_cast<uint32_t*>(vb.device_ptr()), n, 0, nullptr);
    std::vector<GLuint> sk = kb.get_data<GLuint>(), sv = vb.get_data<GLuint>();
    check_sorted(sk);
    bool paired = true;
    for (size_t i = 0; i < n; i++) paired = paired && keys[sv[i]] == sk[i];
    CHECK(paired);
}

TEST_CASE("RadixSort-segmented-sort")
{
    // the local sort of the sharded sort: pieces (source rank, bucket) -> buckets in order, each stably sorted by 24 bits
    std::mt19937 gen(77);
    const size_t n = 700001;
    const uint32_t sources = 3, segments = 8;
    std::vector<GLuint> keys(n), vals(n);
    for (auto& k : keys) k = gen() & 0x00FF0FFFu; // duplicate-heavy low 24 bits
    std::iota(vals.begin(), vals.end(), 0u);
    std::vector<RadixSort::Piece> pieces;
    uint64_t at = 0;
    for (uint32_t s = 0; s < sources; s++)
        for (uint32_t g = 0; g < segments; g++)
        {
            uint64_t len = (s == sources - 1 && g == segments - 1) ? n - at : (gen() % (2 * n / (sources * segments)));
            if (at + len > n) len = n - at;
            pieces.push_back({at, len, g});
            at += len;
        }
    CHECK(at == n);
    ShaderStorageBuffer kin(keys), vin(vals), kout(n * sizeof(GLuint)), vout(n * sizeof(GLuint));
    RadixSort radix_sort;
    radix_sort.sort_segments(static_cast<uint32_t*>(kin.device_ptr()), static_cast<uint32_t*>(vin.device_ptr()),
                             static_cast<uint32_t*>(kout.device_ptr()), static_cast<uint32_t*>(vout.device_ptr()), n, pieces, segments, 24);
    std::vector<GLuint> gk = kout.get_data<GLuint>(), gv = vout.get_data<GLuint>();
    // expected: per segment, its pieces laid end to end, std::stable_sort by the low 24 bits
    std::vector<GLuint> ek, ev;
    for (uint32_t g = 0; g < segments; g++)
    {
        std::vector<std::pair<GLuint, GLuint>> seg;
        for (const auto& pc : pieces)
            if (pc.segment == g)
                for (uint64_t i = pc.begin; i < pc.begin + pc.length; i++) seg.push_back({keys[i], vals[i]});
        std::stable_sort(seg.begin(), seg.end(), [](const auto& a, const auto& b) { return (a.first & 0xFFFFFFu) < (b.first & 0xFFFFFFu); });
        for (const auto& e : seg) ek.push_back(e.first), ev.push_back(e.second);
    }
    CHECK(gk == ek);
    CHECK(gv == ev);
}

TEST_CASE("RadixSort-two-host-threads")
{
    // glu_hip.h: distinct handles may be used from distinct host threads.  Two threads, each with its own sorter and
    // buffers, start together (the first launch of a kernel instantiation does its one-time setup under both) and
    // sort inputs large enough for the large-tile kernels and small enough for the single-workgroup kernel.
    struct Job
    {
        size_t n;
        uint32_t seed;
        bool ok = false;
    };
    std::vector<Job> jobs = {{4200003, 1}, {4200003, 2}, {9001, 3}, {9001, 4}};
    auto work = [](Job* job) {
        std::mt19937 gen(job->seed);
        std::vector<GLuint> keys(job->n), vals(job->n);
        for (auto& k : keys) k = gen();
        std::iota(vals.begin(), vals.end(), 0u);
        std::vector<GLuint> order(vals);
        std::stable_sort(order.begin(), order.end(), [&](GLuint a, GLuint b) { return keys[a] < keys[b]; });
        ShaderStorageBuffer kb(keys), vb(vals);
        RadixSort radix_sort;
        bool same = true;
        for (int round = 0; round < 3; round++) // the same objects again: scratch reuse from this thread
        {
            kb.write_data(keys.data(), job->n * sizeof(GLuint));
            vb.write_data(vals.data(), job->n * sizeof(GLuint));
            radix_sort(kb.handle(), vb.handle(), job->n);
            std::vector<GLuint> out_keys = kb.get_data<GLuint>(), out_vals = vb.get_data<GLuint>();
            for (size_t i = 0; i < job->n; i++) same = same && out_vals[i] == order[i] && out_keys[i] == keys[order[i]];
        }
        job->ok = same;
    };
    for (size_t pair = 0; pair < jobs.size(); pair += 2)
    {
        std::thread a(work, &jobs[pair]), b(work, &jobs[pair + 1]);
        a.join();
        b.join();
        CHECK(jobs[pair].ok);
        CHECK(jobs[pair + 1].ok);
    }
}

int main(int argc, char** argv) { return mini_test::run(argc, argv); }
