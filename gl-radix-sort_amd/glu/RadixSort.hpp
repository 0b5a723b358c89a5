// glu/RadixSort.hpp -- glu::RadixSort on MI355X (drop-in for reference glu/RadixSort.hpp:186-354).
#ifndef GLU_RADIXSORT_HPP
#define GLU_RADIXSORT_HPP

#include <cstdint>
#include <type_traits>
#include <vector>

#include "BlellochScan.hpp"
#include "hip_utils.hpp"

namespace glu
{
    /// Stable LSD radix sort of (uint32 key, uint32 value) pairs, ascending by key, in place in the caller's
    /// two buffers.  `radix_sort(keys, vals, count)` only enqueues GPU work (reference semantics); reading a
    /// buffer back waits for it.
    class RadixSort
    {
    public:
        explicit RadixSort() { GLU_CHECK_STATUS(glu_radix_sort_create(&m_impl)); }

        RadixSort(const RadixSort&) = delete;
        RadixSort& operator=(const RadixSort&) = delete;

        ~RadixSort() { glu_radix_sort_destroy(m_impl); }

        /// Allocates (grow-only) the internal scratch for `count` pairs, so that a following sort of at most
        /// `count` pairs allocates nothing.
        void prepare_internal_buffers(size_t count) { GLU_CHECK_STATUS(glu_radix_sort_prepare(m_impl, count)); }

        /// @param num_steps number of 4-bit digits to sort by, starting from the least significant;
        ///        0 (default) or more than 8 = all 32 bits.
        /// Unlike the reference, the result is in key_buffer / val_buffer for every num_steps (the reference
        /// leaves it in a private scratch buffer when num_steps is odd).
        void operator()(GLuint key_buffer, GLuint val_buffer, size_t count, size_t num_steps = 0)
        {
            GLU_CHECK_ARGUMENT(key_buffer, "Invalid key buffer");
            GLU_CHECK_ARGUMENT(val_buffer, "Invalid value buffer");
            if (count <= 1) return;
            GLU_CHECK_STATUS(glu_radix_sort_run(m_impl, key_buffer, val_buffer, count, num_steps));
        }

        /// Native form: raw device pointers + hipStream_t (nullptr = the library queue).
        void operator()(uint32_t* device_keys, uint32_t* device_vals, size_t count, size_t num_steps, void* stream)
        {
            GLU_CHECK_STATUS(glu_radix_sort_run_ptr(m_impl, device_keys, device_vals, count, num_steps, stream));
        }

        /// Keys only (not in the reference, which needs a dummy value buffer): sorts `count` uint32 keys in place.
        void sort_keys(GLuint key_buffer, size_t count, size_t num_steps = 0)
        {
            GLU_CHECK_ARGUMENT(key_buffer, "Invalid key buffer");
            if (count <= 1) return;
            GLU_CHECK_STATUS(glu_radix_sort_run_keys(m_impl, key_buffer, count, num_steps));
        }

        /// Typed keys on raw device pointers (not in the reference): KeyT in {uint32_t, int32_t, float, uint64_t,
        /// int64_t, double}; natural order (floats as a total order), stable; device_vals may be nullptr.
        template<typename KeyT>
        void sort_typed(KeyT* device_keys, uint32_t* device_vals, size_t count, void* stream = nullptr)
        {
            GLU_CHECK_STATUS(glu_radix_sort_run_typed_ptr(m_impl, device_keys, device_vals, count, key_type_of<KeyT>(), stream));
        }

        /// Stable sort by the key bits [begin_bit, end_bit) only (not in the reference): KeyT in {uint32_t, uint64_t},
        /// device_vals may be nullptr.
        template<typename KeyT>
        void sort_bit_range(KeyT* device_keys, uint32_t* device_vals, size_t count, uint32_t begin_bit, uint32_t end_bit,
                            void* stream = nullptr)
        {
            static_assert(std::is_same<KeyT, uint32_t>::value || std::is_same<KeyT, uint64_t>::value, "unsigned keys");
            GLU_CHECK_STATUS(glu_radix_sort_run_bit_range_ptr(m_impl, device_keys, device_vals, count, sizeof(KeyT) * 8,
                                                              begin_bit, end_bit, stream));
        }

        /// One piece of the input of sort_segments: elements [begin, begin + length) of the input arrays, part of `segment`.
        struct Piece
        {
            uint64_t begin, length;
            uint32_t segment;
        };
        /// Segmented stable sort on raw device pointers (not in the reference; the local sort of the sharded sort): the
        /// pieces -- together exactly `count` elements -- are grouped into `num_segments` segments (a segment = its pieces
        /// laid end to end in the order they are listed); the output arrays receive the segments in ascending order, each
        /// stably sorted by its low `key_bits` key bits (0, 8, 16, 24 or 32).  in != out; the input arrays are clobbered.
        void sort_segments(uint32_t* in_keys, uint32_t* in_vals, uint32_t* out_keys, uint32_t* out_vals, size_t count,
                           const std::vector<Piece>& pieces, uint32_t num_segments, uint32_t key_bits, void* stream = nullptr)
        {
            std::vector<uint64_t> begin(pieces.size()), length(pieces.size());
            std::vector<uint32_t> segment(pieces.size());
            for (size_t i = 0; i < pieces.size(); i++) begin[i] = pieces[i].begin, length[i] = pieces[i].length, segment[i] = pieces[i].segment;
            GLU_CHECK_STATUS(glu_radix_sort_run_segments_ptr(m_impl, in_keys, in_vals, out_keys, out_vals, count, begin.data(), length.data(),
                                                             segment.data(), pieces.size(), num_segments, key_bits, stream));
        }

        /// 64-bit keys with 32-bit values (not in the reference); num_steps counts 4-bit digits, 0 = all 64 bits.
        void sort_u64(GLuint key_buffer, GLuint val_buffer, size_t count, size_t num_steps = 0)
        {
            GLU_CHECK_ARGUMENT(key_buffer, "Invalid key buffer");
            GLU_CHECK_ARGUMENT(val_buffer, "Invalid value buffer");
            if (count <= 1) return;
            GLU_CHECK_STATUS(glu_radix_sort_run_u64(m_impl, key_buffer, val_buffer, count, num_steps));
        }
        void prepare_internal_buffers_u64(size_t count) { GLU_CHECK_STATUS(glu_radix_sort_prepare_u64(m_impl, count)); }
        /// Scratch for any entry point: key_bytes 4 or 8, with_vals false for the keys-only sorts.
        void prepare_internal_buffers(size_t count, size_t key_bytes, bool with_vals)
        {
            GLU_CHECK_STATUS(glu_radix_sort_prepare_ex(m_impl, count, key_bytes, with_vals ? 1 : 0));
        }

        /// Bits per counting pass: 4 = the reference's pass structure (8 passes), 8 = 4 passes; same output.
        void set_digit_bits(uint32_t bits) { GLU_CHECK_STATUS(glu_radix_sort_set_digit_bits(m_impl, bits)); }
        /// A switch of this object for tests / tuning (glu_radix_sort_set_option in glu_hip.h: "SORT_LDS_FINISH", "SORT_FORK", ...);
        /// the process environment only supplies defaults when an object is created.
        void set_option(const char* name, long long value) { GLU_CHECK_STATUS(glu_radix_sort_set_option(m_impl, name, value)); }
        [[nodiscard]] uint32_t digit_bits() const
        {
            uint32_t bits = 0;
            GLU_CHECK_STATUS(glu_radix_sort_get_digit_bits(m_impl, &bits));
            return bits;
        }

        /// What the last sort did about ending in LDS (glu_radix_sort_read_finish in glu_hip.h; synchronise its stream first):
        /// large whole-key sorts first try two counting passes on 16 top key bits and one pass that orders every run of
        /// equal top bits inside LDS; the device refuses if a run outgrows an LDS tile, and the ordinary passes run.
        struct FinishReport
        {
            bool attempted = false, accepted = false;
            uint32_t longest_run = 0, capacity = 0, top_bit = 0;
        };
        [[nodiscard]] FinishReport last_finish() const
        {
            uint32_t a = 0, b = 0;
            FinishReport r;
            GLU_CHECK_STATUS(glu_radix_sort_read_finish(m_impl, &a, &b, &r.longest_run, &r.capacity, &r.top_bit));
            r.attempted = a != 0;
            r.accepted = b != 0;
            return r;
        }

    private:
        template<typename KeyT>
        static constexpr glu_key_type key_type_of()
        {
            static_assert(std::is_same_v<KeyT, uint32_t> || std::is_same_v<KeyT, int32_t> || std::is_same_v<KeyT, float> ||
                              std::is_same_v<KeyT, uint64_t> || std::is_same_v<KeyT, int64_t> || std::is_same_v<KeyT, double>,
                          "unsupported key type");
            if constexpr (std::is_same_v<KeyT, uint32_t>) return GLU_KEY_UINT32;
            else if constexpr (std::is_same_v<KeyT, int32_t>) return GLU_KEY_INT32;
            else if constexpr (std::is_same_v<KeyT, float>) return GLU_KEY_FLOAT32;
            else if constexpr (std::is_same_v<KeyT, uint64_t>) return GLU_KEY_UINT64;
            else if constexpr (std::is_same_v<KeyT, int64_t>) return GLU_KEY_INT64;
            else return GLU_KEY_FLOAT64;
        }

        glu_radix_sort m_impl = nullptr;
    };
} // namespace glu

#endif // GLU_RADIXSORT_HPP
