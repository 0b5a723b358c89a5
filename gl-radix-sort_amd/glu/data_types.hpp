// glu/data_types.hpp -- element types accepted by glu::Reduce / glu::BlellochScan
// (reference glu/data_types.hpp:8-44; same enumerator names and values, std430 strides).
#ifndef GLU_DATA_TYPES_HPP
#define GLU_DATA_TYPES_HPP

#include <cstddef>

#include "errors.hpp"

namespace glu
{
    enum DataType
    {
        DataType_Float = GLU_DATA_TYPE_FLOAT,
        DataType_Double = GLU_DATA_TYPE_DOUBLE,
        DataType_Int = GLU_DATA_TYPE_INT,
        DataType_Uint = GLU_DATA_TYPE_UINT,
        DataType_Vec2 = GLU_DATA_TYPE_VEC2,
        DataType_Vec4 = GLU_DATA_TYPE_VEC4,
        DataType_DVec2 = GLU_DATA_TYPE_DVEC2,
        DataType_DVec4 = GLU_DATA_TYPE_DVEC4,
        DataType_UVec2 = GLU_DATA_TYPE_UVEC2,
        DataType_UVec4 = GLU_DATA_TYPE_UVEC4,
        DataType_IVec2 = GLU_DATA_TYPE_IVEC2,
        DataType_IVec4 = GLU_DATA_TYPE_IVEC4
    };

    namespace detail
    {
        struct DataTypeInfo
        {
            const char* glsl_name; ///< the name the reference pastes into its shaders
            size_t size;           ///< bytes per element in the buffer (std430 array stride)
            int components;
        };

        inline const DataTypeInfo& data_type_info(DataType data_type)
        {
            static const DataTypeInfo k_table[] = {
                {"float", 4, 1},  {"double", 8, 1}, {"int", 4, 1},    {"uint", 4, 1},
                {"vec2", 8, 2},   {"vec4", 16, 4},  {"dvec2", 16, 2}, {"dvec4", 32, 4},
                {"uvec2", 8, 2},  {"uvec4", 16, 4}, {"ivec2", 8, 2},  {"ivec4", 16, 4},
            };
            const int index = static_cast<int>(data_type);
            GLU_CHECK_ARGUMENT(index >= 0 && index < int(sizeof(k_table) / sizeof(k_table[0])), "Invalid data type: %d",
                               index);
            return k_table[index];
        }
    } // namespace detail

    /// Kept for source compatibility (reference glu/data_types.hpp:24-44).
    inline const char* to_glsl_type_str(DataType data_type) { return detail::data_type_info(data_type).glsl_name; }

    /// Bytes per element of `data_type` inside a buffer.
    inline size_t data_type_size(DataType data_type) { return detail::data_type_info(data_type).size; }
} // namespace glu

#endif // GLU_DATA_TYPES_HPP
