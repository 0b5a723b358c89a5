// C++ API tests of glu::DistributedRadixSort (the sharded sort of BASELINE.json configs[3]; the reference has no
// multi-device path).
//   * the plan (bucket -> rank map, send / receive counts) on simulated ranks: host only;
//   * a whole sharded sort with simulated ranks: the plan + the library's partition pass + std::stable_sort stand in for
//     the exchange and the local sorts, checked against std::stable_sort of the whole array;
//   * a real one-rank RCCL communicator on the GPU: the full glu_dist code path (partition, ncclAllGather, plan, exchange
//     = the device copy of the rank's own part, local sort) against std::stable_sort.
#include <algorithm>
#include <numeric>
#include <random>
#include <vector>

#include "glu/DistributedRadixSort.hpp"
#include "glu/RadixSort.hpp"
#include "util/mini_test.hpp"

using namespace glu;

namespace
{
    std::vector<uint32_t> histograms(const std::vector<std::vector<GLuint>>& slices)
    {
        std::vector<uint32_t> h(slices.size() * 256, 0);
        for (size_t r = 0; r < slices.size(); r++)
            for (GLuint k : slices[r]) h[r * 256 + (k >> 24)]++;
        return h;
    }

    void check_plan(const std::vector<uint32_t>& all_hist, int world)
    {
        std::vector<int> owner = DistributedRadixSort::plan_buckets(all_hist, world);
        // monotone, contiguous, every rank id valid
        bool monotone = owner[0] >= 0;
        for (int b = 1; b < 256; b++) monotone = monotone && owner[b] >= owner[b - 1] && owner[b] < world;
        CHECK(monotone);
        // every cut sits on the bucket boundary nearest to r * N / R (unless pushed up by the cut before it)
        std::vector<uint64_t> prefix(257, 0);
        for (int b = 0; b < 256; b++)
        {
            uint64_t t = 0;
            for (int r = 0; r < world; r++) t += all_hist[size_t(r) * 256 + b];
            prefix[b + 1] = prefix[b] + t;
        }
        const uint64_t n = prefix[256];
        int prev_cut = 0;
        for (int r = 1; r < world; r++)
        {
            int cut = int(std::lower_bound(owner.begin(), owner.end(), r) - owner.begin());
            const uint64_t target = n * uint64_t(r) / uint64_t(world);
            auto dist_to = [&](int b) { return prefix[b] > target ? prefix[b] - target : target - prefix[b]; };
            bool nearest = true;
            for (int b = prev_cut; b <= 256; b++) nearest = nearest && dist_to(cut) <= dist_to(b);
            CHECK(nearest);
            prev_cut = cut;
        }
        // counts: what rank s sends to d is what d receives from s; everything is sent exactly once
        std::vector<std::vector<uint64_t>> send(world), recv(world);
        for (int r = 0; r < world; r++) DistributedRadixSort::plan_counts(all_hist, world, r, owner, send[r], recv[r]);
        uint64_t total = 0;
        bool consistent = true;
        for (int s = 0; s < world; s++)
            for (int d = 0; d < world; d++)
            {
                consistent = consistent && send[s][d] == recv[d][s];
                total += send[s][d];
            }
        CHECK(consistent);
        CHECK(total == n);
    }
} // namespace

TEST_CASE("DistributedRadixSort-plan-simulated-ranks")
{
    std::mt19937 gen(7);
    for (int world : {1, 2, 3, 4, 8, 16})
        for (int kind = 0; kind < 6; kind++)
        {
            std::vector<uint32_t> h(size_t(world) * 256, 0);
            for (auto& c : h) c = kind == 1 ? 0u : uint32_t(gen() % 5000);
            if (kind == 2)
                for (int r = 0; r < world; r++) h[size_t(r) * 256 + 17] += 4000000u; // one hot bucket
            if (kind == 3)
                for (int r = 0; r < world; r++)
                    for (int b = 1; b < 256; b++) h[size_t(r) * 256 + b] = 0; // everything in bucket 0 (keys < 2^24)
            if (kind == 4)
                for (int r = 0; r < world; r++)
                    for (int b = 0; b < 255; b++) h[size_t(r) * 256 + b] = 0; // everything in the last bucket
            if (kind == 5)
                for (int r = 1; r < world; r++)
                    for (int b = 0; b < 256; b++) h[size_t(r) * 256 + b] = 0; // only rank 0 has data
            check_plan(h, world);
        }
}

TEST_CASE("DistributedRadixSort-simulated-ranks-equal-single-device-sort")
{
    // R simulated ranks in one process: partition every slice on the GPU (the pass the sharded sort starts with), route
    // the bucket groups by the plan on the host, stable-sort every shard, concatenate: must equal the stable sort of the
    // whole array, values included (duplicate-heavy keys).
    std::mt19937 gen(11);
    for (int world : {2, 3, 8})
    {
        const size_t per_rank = 40000 + 1234 * world;
        std::vector<std::vector<GLuint>> keys(world), vals(world);
        std::vector<GLuint> all_keys, all_vals;
        for (int r = 0; r < world; r++)
        {
            const size_t n = r == 1 ? per_rank / 3 : per_rank; // ragged slices
            keys[r].resize(n);
            vals[r].resize(n);
            for (size_t i = 0; i < n; i++)
            {
                keys[r][i] = (gen() % 4 == 0) ? (GLuint(gen() % 7) << 29) | 5u : GLuint(gen());
                vals[r][i] = GLuint(all_keys.size());
                all_keys.push_back(keys[r][i]);
                all_vals.push_back(vals[r][i]);
            }
        }
        std::vector<uint32_t> all_hist = histograms(keys);
        std::vector<int> owner = DistributedRadixSort::plan_buckets(all_hist, world);
        // device partition of every slice by the top 8 bits (stable), through the C ABI
        std::vector<std::vector<GLuint>> part_k(world), part_v(world);
        for (int r = 0; r < world; r++)
        {
            const size_t n = keys[r].size();
            ShaderStorageBuffer kb(keys[r]), vb(vals[r]), ok(n * sizeof(GLuint)), ov(n * sizeof(GLuint)), hist(256 * sizeof(GLuint));
            glu_radix_sort sorter = nullptr;
            GLU_CHECK_STATUS(glu_radix_sort_create(&sorter));
            GLU_CHECK_STATUS(glu_radix_sort_partition_ptr(sorter, static_cast<const uint32_t*>(kb.device_ptr()),
                                                          static_cast<const uint32_t*>(vb.device_ptr()), static_cast<uint32_t*>(ok.device_ptr()),
                                                          static_cast<uint32_t*>(ov.device_ptr()), n, 24, 8,
                                                          static_cast<uint32_t*>(hist.device_ptr()), nullptr));
            part_k[r] = ok.get_data<GLuint>();
            part_v[r] = ov.get_data<GLuint>();
            std::vector<GLuint> h = hist.get_data<GLuint>();
            bool same_hist = true;
            for (int b = 0; b < 256; b++) same_hist = same_hist && h[b] == all_hist[size_t(r) * 256 + b];
            CHECK(same_hist);
            glu_radix_sort_destroy(sorter);
        }
        // exchange on the host: destination d receives, in source-rank order, the contiguous bucket group of every source
        std::vector<GLuint> out_keys, out_vals;
        for (int d = 0; d < world; d++)
        {
            std::vector<uint64_t> send, recv;
            std::vector<GLuint> sk, sv;
            for (int s = 0; s < world; s++)
            {
                DistributedRadixSort::plan_counts(all_hist, world, s, owner, send, recv);
                uint64_t off = 0;
                for (int x = 0; x < d; x++) off += send[x];
                sk.insert(sk.end(), part_k[s].begin() + off, part_k[s].begin() + off + send[d]);
                sv.insert(sv.end(), part_v[s].begin() + off, part_v[s].begin() + off + send[d]);
            }
            std::vector<size_t> order(sk.size());
            std::iota(order.begin(), order.end(), size_t(0));
            std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return sk[a] < sk[b]; });
            for (size_t i : order)
            {
                out_keys.push_back(sk[i]);
                out_vals.push_back(sv[i]);
            }
        }
        std::vector<size_t> order(all_keys.size());
        std::iota(order.begin(), order.end(), size_t(0));
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return all_keys[a] < all_keys[b]; });
        REQUIRE(out_keys.size() == all_keys.size());
        bool same = true;
        for (size_t i = 0; i < order.size(); i++) same = same && out_keys[i] == all_keys[order[i]] && out_vals[i] == all_vals[order[i]];
        CHECK(same);
    }
}

TEST_CASE("DistributedRadixSort-one-rank-rccl")
{
    // the whole native path with a real (one-rank) RCCL communicator
    DistributedRadixSort::UniqueId id = DistributedRadixSort::unique_id();
    DistributedRadixSort dsort(id, 1, 0);
    CHECK(dsort.world_size() == 1);
    CHECK(dsort.rank() == 0);
    std::mt19937 gen(3);
    for (size_t n : {size_t(0), size_t(1), size_t(1000), size_t(300001), size_t(4 * 1024 * 1024 + 77)})
    {
        std::vector<GLuint> keys(n), vals(n);
        for (auto& k : keys) k = (gen() % 8 == 0) ? 0xABCD0000u : GLuint(gen());
        std::iota(vals.begin(), vals.end(), 0u);
        ShaderStorageBuffer kb(n ? n * sizeof(GLuint) : 4), vb(n ? n * sizeof(GLuint) : 4);
        if (n)
        {
            kb.write_data(keys.data(), n * sizeof(GLuint));
            vb.write_data(vals.data(), n * sizeof(GLuint));
        }
        DistributedRadixSort::Shard shard = dsort(kb.handle(), vb.handle(), n);
        REQUIRE(shard.count == n);
        CHECK(!dsort.last_local_sort_was_segmented()); // shards below 2^24 pairs take the ordinary local sort
        std::vector<GLuint> out_k(n), out_v(n);
        GLU_CHECK_STATUS(glu_device_synchronize());
        if (n)
        {
            GLuint kh = 0, vh = 0;
            GLU_CHECK_STATUS(glu_buffer_wrap(shard.keys, n * sizeof(GLuint), &kh));
            GLU_CHECK_STATUS(glu_buffer_wrap(shard.vals, n * sizeof(GLuint), &vh));
            GLU_CHECK_STATUS(glu_buffer_read(kh, out_k.data(), n * sizeof(GLuint), 0));
            GLU_CHECK_STATUS(glu_buffer_read(vh, out_v.data(), n * sizeof(GLuint), 0));
            glu_buffer_destroy(kh);
            glu_buffer_destroy(vh);
        }
        std::vector<GLuint> order(vals);
        std::stable_sort(order.begin(), order.end(), [&](GLuint a, GLuint b) { return keys[a] < keys[b]; });
        bool same = true;
        for (size_t i = 0; i < n; i++) same = same && out_v[i] == order[i] && out_k[i] == keys[order[i]];
        CHECK(same);
        // the caller's slice is untouched
        if (n)
        {
            std::vector<GLuint> in_k = kb.get_data<GLuint>();
            CHECK(in_k == keys);
        }
    }
}

TEST_CASE("DistributedRadixSort-one-rank-segmented-local-sort")
{
    // a shard of 2^24 pairs: the exchange lands in the sorter's scratch and three segmented passes produce the shard
    DistributedRadixSort::UniqueId id = DistributedRadixSort::unique_id();
    DistributedRadixSort dsort(id, 1, 0);
    const size_t n = (size_t(1) << 24) + 12345;
    std::mt19937 gen(11);
    std::vector<GLuint> keys(n), vals(n);
    for (auto& k : keys) k = (gen() % 16 == 0) ? 0x7F00FF00u : GLuint(gen());
    std::iota(vals.begin(), vals.end(), 0u);
    ShaderStorageBuffer kb(keys), vb(vals);
    dsort.prepare_internal_buffers(n, n);
    DistributedRadixSort::Shard shard = dsort(kb.handle(), vb.handle(), n);
    REQUIRE(shard.count == n);
    CHECK(dsort.last_local_sort_was_segmented());
    GLU_CHECK_STATUS(glu_device_synchronize());
    std::vector<GLuint> out_k(n), out_v(n);
    GLuint kh = 0, vh = 0;
    GLU_CHECK_STATUS(glu_buffer_wrap(shard.keys, n * sizeof(GLuint), &kh));
    GLU_CHECK_STATUS(glu_buffer_wrap(shard.vals, n * sizeof(GLuint), &vh));
    GLU_CHECK_STATUS(glu_buffer_read(kh, out_k.data(), n * sizeof(GLuint), 0));
    GLU_CHECK_STATUS(glu_buffer_read(vh, out_v.data(), n * sizeof(GLuint), 0));
    glu_buffer_destroy(kh);
    glu_buffer_destroy(vh);
    std::vector<GLuint> order(vals);
    std::stable_sort(order.begin(), order.end(), [&](GLuint a, GLuint b) { return keys[a] < keys[b]; });
    bool same = true;
    for (size_t i = 0; i < n; i++) same = same && out_v[i] == order[i] && out_k[i] == keys[order[i]];
    CHECK(same);
}

int main(int argc, char** argv) { return mini_test::run(argc, argv); }
