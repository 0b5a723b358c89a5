// device_utils.hpp -- small device helpers shared by the sort and the scan / reduce kernels (gfx950).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace glu_hip
{
// Non-temporal load of an object whose size is a multiple of 16 bytes (HIP vector types, packs of elements), 16 bytes at
// a time: for streams that are read once.  Keeping them out of the Infinity Cache matters twice on MI355X: the stream
// does not evict the dirty lines an earlier kernel left there (the count pass behind a scatter: 0.255 -> 0.22 ms at 2^28
// keys), and the read itself is faster (0.19 -> 0.17 ms alone).  Objects of other sizes take a plain load.
template<typename T>
__device__ __forceinline__ T load_streaming(const T* p)
{
    if constexpr (sizeof(T) % 16 == 0)
    {
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        T out;
        const u32x4_t* src = reinterpret_cast<const u32x4_t*>(p);
#pragma unroll
        for (unsigned i = 0; i < sizeof(T) / 16; i++)
        {
            const u32x4_t r = __builtin_nontemporal_load(src + i);
            __builtin_memcpy(reinterpret_cast<char*>(&out) + 16 * i, &r, 16);
        }
        return out;
    }
    else
        return *p;
}
// Non-temporal store, 16 bytes at a time (objects whose size is a multiple of 16 bytes), plain store otherwise.
template<typename T>
__device__ __forceinline__ void store_streaming(T* p, const T& value)
{
    if constexpr (sizeof(T) % 16 == 0)
    {
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t* dst = reinterpret_cast<u32x4_t*>(p);
#pragma unroll
        for (unsigned i = 0; i < sizeof(T) / 16; i++)
        {
            u32x4_t r;
            __builtin_memcpy(&r, reinterpret_cast<const char*>(&value) + 16 * i, 16);
            __builtin_nontemporal_store(r, dst + i);
        }
    }
    else
        *p = value;
}
} // namespace glu_hip
