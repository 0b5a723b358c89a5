// glu/DistributedRadixSort.hpp -- glu::RadixSort sharded over the GPUs of one node (one process per GPU).
//
// The reference (loryruta/gl-radix-sort) is single-device; this is the C++ face of the glu_dist_* entry points of
// glu_hip.h: rank r holds slice r of the array, the sort partitions by the top 8 key bits, exchanges once over RCCL /
// xGMI, and sorts locally.  The ranks' shards concatenated in rank order equal glu::RadixSort of the whole array
// (stable); shard sizes follow the data.  Same failure convention as the other operators (print + exit(1)); the C
// entry points underneath agree on a rank's failure before anything is exchanged, so when one rank's call fails EVERY
// rank prints and exits -- none is left waiting in a collective.
#ifndef GLU_DISTRIBUTEDRADIXSORT_HPP
#define GLU_DISTRIBUTEDRADIXSORT_HPP

#include <array>
#include <cstdint>
#include <vector>

#include "hip_utils.hpp"

namespace glu
{
    class DistributedRadixSort
    {
    public:
        using UniqueId = std::array<unsigned char, GLU_DIST_UNIQUE_ID_BYTES>;

        /// Rank 0 draws the id; the application carries it to the other ranks (MPI_Bcast, a file, a socket ...).
        static UniqueId unique_id()
        {
            UniqueId id{};
            GLU_CHECK_STATUS(glu_dist_unique_id(id.data(), id.size()));
            return id;
        }

        /// Collective over all ranks (creates the RCCL communicator on the library's device, glu_set_device).
        DistributedRadixSort(const UniqueId& id, int world_size, int rank)
        {
            GLU_CHECK_STATUS(glu_dist_create(id.data(), id.size(), world_size, rank, &m_impl));
        }
        DistributedRadixSort(const DistributedRadixSort&) = delete;
        DistributedRadixSort& operator=(const DistributedRadixSort&) = delete;
        ~DistributedRadixSort() { glu_dist_destroy(m_impl); }

        /// Grow-only buffers for slices of `local_count` pairs and shards of `recv_capacity` pairs
        /// (the analogue of RadixSort::prepare_internal_buffers).
        void prepare_internal_buffers(size_t local_count, size_t recv_capacity)
        {
            GLU_CHECK_STATUS(glu_dist_prepare(m_impl, local_count, recv_capacity));
        }

        /// This rank's shard of the sorted array: device pointers owned by the object, valid until its next sort.
        struct Shard
        {
            uint32_t* keys = nullptr;
            uint32_t* vals = nullptr;
            size_t count = 0;
        };

        /// Sorts the global array whose slice on this rank is (keys, vals)[0, local_count); collective.  Enqueues on
        /// `stream` (a hipStream_t, nullptr = the library queue); the host waits only for the 8 KiB histogram exchange.
        Shard operator()(const uint32_t* keys, const uint32_t* vals, size_t local_count, void* stream = nullptr)
        {
            Shard s;
            GLU_CHECK_STATUS(glu_dist_sort_ptr(m_impl, keys, vals, local_count, stream, &s.keys, &s.vals, &s.count));
            return s;
        }
        /// Buffer-handle form (the reference's calling convention): the slice lives in two ShaderStorageBuffers.
        Shard operator()(GLuint key_buffer, GLuint val_buffer, size_t local_count)
        {
            GLU_CHECK_ARGUMENT(key_buffer, "Invalid key buffer");
            GLU_CHECK_ARGUMENT(val_buffer, "Invalid value buffer");
            void *k = nullptr, *v = nullptr;
            GLU_CHECK_STATUS(glu_buffer_device_ptr(key_buffer, &k));
            GLU_CHECK_STATUS(glu_buffer_device_ptr(val_buffer, &v));
            return (*this)(static_cast<const uint32_t*>(k), static_cast<const uint32_t*>(v), local_count, nullptr);
        }

        /// true if the local sort of the last sort was the segmented sort of the low 24 bits per bucket -- one segmented pass + an
        /// in-LDS pass where the runs fit an LDS tile, else three segmented passes -- (the shard arrives as one
        /// message per source rank, each grouped by bucket: RadixSort::sort_segments), false for the ordinary sort of all 32
        /// bits (small or very fragmented shards, a partition on a lower key byte).
        [[nodiscard]] bool last_local_sort_was_segmented() const
        {
            uint32_t seg = 0;
            GLU_CHECK_STATUS(glu_dist_last_local_sort(m_impl, &seg));
            return seg != 0;
        }

        /// Rounds of the exchange (glu_dist_set_rounds): groups of every rank's buckets travel one after the other while the
        /// groups that have arrived are sorted -- the default (3) for one sort at a time; 1 for callers that keep several sorts
        /// in flight on several objects.  The same value on every rank.
        void set_exchange_rounds(int rounds) { GLU_CHECK_STATUS(glu_dist_set_rounds(m_impl, rounds)); }
        [[nodiscard]] int last_exchange_rounds() const
        {
            uint32_t r = 0;
            GLU_CHECK_STATUS(glu_dist_last_rounds(m_impl, &r));
            return (int) r;
        }

        [[nodiscard]] int world_size() const
        {
            int w = 0;
            GLU_CHECK_STATUS(glu_dist_world(m_impl, &w, nullptr));
            return w;
        }
        [[nodiscard]] int rank() const
        {
            int r = 0;
            GLU_CHECK_STATUS(glu_dist_world(m_impl, nullptr, &r));
            return r;
        }

        /// The plan every rank derives from the gathered bucket histograms (host only: usable without a GPU).
        /// all_hist: [world_size][256] counts; returns the owner rank of every bucket.
        static std::vector<int> plan_buckets(const std::vector<uint32_t>& all_hist, int world_size)
        {
            GLU_CHECK_ARGUMENT(all_hist.size() == size_t(world_size) * 256, "all_hist must hold world_size x 256 counts");
            std::vector<int> owner(256);
            GLU_CHECK_STATUS(glu_dist_plan_buckets(all_hist.data(), world_size, owner.data()));
            return owner;
        }
        static void plan_counts(const std::vector<uint32_t>& all_hist, int world_size, int rank, const std::vector<int>& owner,
                                std::vector<uint64_t>& send_counts, std::vector<uint64_t>& recv_counts)
        {
            send_counts.assign(world_size, 0);
            recv_counts.assign(world_size, 0);
            GLU_CHECK_STATUS(glu_dist_plan_counts(all_hist.data(), world_size, rank, owner.data(), send_counts.data(), recv_counts.data()));
        }

    private:
        glu_dist m_impl = nullptr;
    };
} // namespace glu

#endif // GLU_DISTRIBUTEDRADIXSORT_HPP
