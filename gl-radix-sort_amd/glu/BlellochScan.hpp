// glu/BlellochScan.hpp -- glu::BlellochScan on MI355X (drop-in for reference glu/BlellochScan.hpp:80-191).
#ifndef GLU_BLELLOCHSCAN_HPP
#define GLU_BLELLOCHSCAN_HPP

#include "data_types.hpp"
#include "hip_utils.hpp"

namespace glu
{
    /// Exclusive prefix sum (`+`, identity 0), in place, over `num_partitions` adjacent partitions of `count`
    /// elements each.  Keeps the reference's name although the device algorithm is a chunked
    /// reduce-then-scan rather than Blelloch's up/down sweep; results are identical for integer types.
    class BlellochScan
    {
    public:
        explicit BlellochScan(DataType data_type) :
            m_data_type(data_type)
        {
            GLU_CHECK_STATUS(glu_scan_create(static_cast<glu_data_type>(data_type), &m_impl));
        }

        BlellochScan(const BlellochScan&) = delete;
        BlellochScan& operator=(const BlellochScan&) = delete;

        ~BlellochScan() { glu_scan_destroy(m_impl); }

        /// @param buffer the buffer to scan
        /// @param count elements per partition (must be a power of 2, as in the reference)
        /// @param num_partitions number of adjacent partitions
        void operator()(GLuint buffer, size_t count, size_t num_partitions = 1)
        {
            GLU_CHECK_ARGUMENT(buffer, "Invalid buffer");
            GLU_CHECK_ARGUMENT(count > 0, "Count must be greater than zero");
            GLU_CHECK_ARGUMENT(is_power_of_2(count), "Count must be a power of 2");
            GLU_CHECK_ARGUMENT(num_partitions >= 1, "Num of partitions must be >= 1");
            GLU_CHECK_STATUS(glu_scan_run(m_impl, buffer, count, num_partitions));
        }

        /// Native form: raw device pointer + hipStream_t; any count > 0 is accepted.
        void operator()(void* device_data, size_t count, size_t num_partitions, void* stream)
        {
            GLU_CHECK_STATUS(glu_scan_run_ptr(m_impl, device_data, count, num_partitions, stream));
        }

        [[nodiscard]] DataType data_type() const { return m_data_type; }

    private:
        const DataType m_data_type;
        glu_scan m_impl = nullptr;
    };
} // namespace glu

#endif // GLU_BLELLOCHSCAN_HPP
