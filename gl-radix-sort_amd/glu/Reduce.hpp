// glu/Reduce.hpp -- glu::Reduce on MI355X (drop-in for reference glu/Reduce.hpp:42-136).
#ifndef GLU_REDUCE_HPP
#define GLU_REDUCE_HPP

#include "data_types.hpp"
#include "hip_utils.hpp"

namespace glu
{
    /// The operators that can be used for the reduction (reference glu/Reduce.hpp:42-48).
    enum ReduceOperator
    {
        ReduceOperator_Sum = GLU_REDUCE_SUM,
        ReduceOperator_Mul = GLU_REDUCE_MUL,
        ReduceOperator_Min = GLU_REDUCE_MIN,
        ReduceOperator_Max = GLU_REDUCE_MAX
    };

    /// In-place reduction: after `reduce(buffer, count)` element 0 of the buffer holds the (component-wise)
    /// sum / product / min / max of elements [0, count).  The work is enqueued, not waited for.
    class Reduce
    {
    public:
        explicit Reduce(DataType data_type, ReduceOperator operator_) :
            m_data_type(data_type),
            m_operator(operator_)
        {
            GLU_CHECK_STATUS(glu_reduce_create(static_cast<glu_data_type>(data_type),
                                               static_cast<glu_reduce_operator>(operator_), &m_impl));
        }

        Reduce(const Reduce&) = delete;
        Reduce& operator=(const Reduce&) = delete;

        ~Reduce() { glu_reduce_destroy(m_impl); }

        void operator()(GLuint buffer, size_t count)
        {
            GLU_CHECK_ARGUMENT(buffer, "Invalid buffer");
            GLU_CHECK_ARGUMENT(count > 0, "Count must be greater than zero");
            GLU_CHECK_STATUS(glu_reduce_run(m_impl, buffer, count));
        }

        /// Native form: raw device pointer + hipStream_t (nullptr = the library queue).
        void operator()(void* device_data, size_t count, void* stream)
        {
            GLU_CHECK_STATUS(glu_reduce_run_ptr(m_impl, device_data, count, stream));
        }

        [[nodiscard]] DataType data_type() const { return m_data_type; }
        [[nodiscard]] ReduceOperator reduce_operator() const { return m_operator; }

    private:
        const DataType m_data_type;
        const ReduceOperator m_operator;
        glu_reduce m_impl = nullptr;
    };
} // namespace glu

#endif // GLU_REDUCE_HPP
