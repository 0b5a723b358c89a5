// glu/hip_utils.hpp -- buffer RAII, device timer and integer helpers of the reference's glu/gl_utils.hpp,
// re-created on the C ABI of libglu_hip.so instead of OpenGL.
//
//   ShaderStorageBuffer        reference glu/gl_utils.hpp:146-246  (same members, move-only)
//   copy_buffer                :13-22
//   measure_gl_elapsed_time    :249-265  (also available as measure_elapsed_time)
//   div_ceil / is_power_of_2 / next_power_of_2   :279-302
//   print_stl_container / print_buffer / print_buffer_hex   :304-329
// Shader / Program (:25-143) have no counterpart: the kernels are compiled ahead of time for gfx950.
#ifndef GLU_HIP_UTILS_HPP
#define GLU_HIP_UTILS_HPP

#include <algorithm>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "errors.hpp"

/// Buffer names keep the reference's type: a 32-bit handle, 0 = no buffer.  (Re-declaring the typedef next to
/// a real GL header is harmless: both are `unsigned int`.)
typedef unsigned int GLuint;

namespace glu
{
    inline void
    copy_buffer(GLuint src_buffer, GLuint dst_buffer, size_t size, size_t src_offset = 0, size_t dst_offset = 0)
    {
        GLU_CHECK_STATUS(glu_buffer_copy(src_buffer, dst_buffer, size, src_offset, dst_offset));
    }

    /// RAII owner of one device buffer (HBM allocation) addressed by a GLuint-style handle.
    class ShaderStorageBuffer
    {
    public:
        explicit ShaderStorageBuffer(size_t initial_size = 0)
        {
            if (initial_size > 0) resize(initial_size, false);
        }

        explicit ShaderStorageBuffer(const void* data, size_t size)
        {
            GLU_CHECK_ARGUMENT(data, "ShaderStorageBuffer: data is null");
            GLU_CHECK_ARGUMENT(size > 0, "ShaderStorageBuffer: size is zero");
            GLU_CHECK_STATUS(glu_buffer_create_with_data(data, size, &m_handle));
            m_size = size;
        }

        template<typename T>
        explicit ShaderStorageBuffer(const std::vector<T>& data) :
            ShaderStorageBuffer(data.data(), data.size() * sizeof(T))
        {
        }

        ShaderStorageBuffer(const ShaderStorageBuffer&) = delete;
        ShaderStorageBuffer& operator=(const ShaderStorageBuffer&) = delete;

        ShaderStorageBuffer(ShaderStorageBuffer&& other) noexcept :
            m_handle(other.m_handle),
            m_size(other.m_size)
        {
            other.m_handle = 0;
            other.m_size = 0;
        }

        ~ShaderStorageBuffer()
        {
            if (m_handle) glu_buffer_destroy(m_handle);
        }

        [[nodiscard]] GLuint handle() const { return m_handle; }
        [[nodiscard]] size_t size() const { return m_size; }

        /// Raw device address, for callers that mix in their own HIP kernels.
        [[nodiscard]] void* device_ptr() const
        {
            void* p = nullptr;
            if (m_handle) GLU_CHECK_STATUS(glu_buffer_device_ptr(m_handle, &p));
            return p;
        }

        /// Grows or shrinks the buffer (a new allocation; the handle changes).  keep_data copies the common prefix.
        void resize(size_t size, bool keep_data = false)
        {
            if (size == m_size) return;
            glu_buffer fresh = 0;
            GLU_CHECK_STATUS(glu_buffer_create(size, &fresh));
            if (keep_data && m_handle) copy_buffer(m_handle, fresh, std::min(m_size, size));
            if (m_handle) GLU_CHECK_STATUS(glu_buffer_destroy(m_handle));
            m_handle = fresh;
            m_size = size;
        }

        /// Fills the whole buffer with a repeated 32-bit value.
        void clear(GLuint value) { GLU_CHECK_STATUS(glu_buffer_fill_u32(m_handle, value)); }

        void write_data(const void* data, size_t size)
        {
            GLU_CHECK_ARGUMENT(size <= m_size, "write_data: %zu bytes do not fit a buffer of %zu", size, m_size);
            GLU_CHECK_STATUS(glu_buffer_write(m_handle, data, size, 0));
        }

        /// Reads back the whole buffer (waits for the queued work that produces it).
        template<typename T>
        std::vector<T> get_data() const
        {
            GLU_CHECK_ARGUMENT(m_size % sizeof(T) == 0, "Size %zu isn't a multiple of %zu", m_size, sizeof(T));
            std::vector<T> result(m_size / sizeof(T));
            if (m_size > 0) GLU_CHECK_STATUS(glu_buffer_read(m_handle, result.data(), m_size, 0));
            return result;
        }

    private:
        GLuint m_handle = 0;
        size_t m_size = 0;
    };

    /// Device time, in nanoseconds, of the work `callback` enqueues (hipEvent pair on the library queue; the
    /// reference uses a GL_TIME_ELAPSED query).  Blocks until that work has finished.
    inline uint64_t measure_elapsed_time(const std::function<void()>& callback)
    {
        glu_timer timer = nullptr;
        GLU_CHECK_STATUS(glu_timer_begin(&timer));
        callback();
        uint64_t elapsed_ns = 0;
        GLU_CHECK_STATUS(glu_timer_end(timer, &elapsed_ns));
        return elapsed_ns;
    }

    /// Source-compatible name.
    inline uint64_t measure_gl_elapsed_time(const std::function<void()>& callback)
    {
        return measure_elapsed_time(callback);
    }

    template<typename IntegerT>
    IntegerT div_ceil(IntegerT n, IntegerT d)
    {
        return (n + d - 1) / d; // exact for every representable n (the reference detours through double)
    }

    template<typename T>
    bool is_power_of_2(T n)
    {
        return (n & (n - 1)) == 0; // true for 0, like the reference
    }

    template<typename IntegerT>
    IntegerT next_power_of_2(IntegerT n)
    {
        if (n == 0) return 0; // the reference's bit-smearing form wraps to 0 here (gl_utils.hpp:291-302)
        IntegerT p = 1;
        while (p < n) p <<= 1;
        return p;
    }

    template<typename Iterator>
    void print_stl_container(Iterator begin, Iterator end)
    {
        size_t i = 0;
        for (; begin != end; ++begin, ++i) std::printf("(%zu) %s, ", i, std::to_string(*begin).c_str());
        std::printf("\n");
    }

    template<typename T>
    void print_buffer(const ShaderStorageBuffer& buffer)
    {
        std::vector<T> data = buffer.get_data<T>();
        print_stl_container(data.begin(), data.end());
    }

    inline void print_buffer_hex(const ShaderStorageBuffer& buffer)
    {
        std::vector<GLuint> data = buffer.get_data<GLuint>();
        for (size_t i = 0; i < data.size(); i++) std::printf("(%zu) %08x, ", i, data[i]);
        std::printf("\n");
    }
} // namespace glu

#endif // GLU_HIP_UTILS_HPP
