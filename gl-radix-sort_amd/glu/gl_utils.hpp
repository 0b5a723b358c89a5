// Source compatibility: the reference's utilities header is glu/gl_utils.hpp; the HIP build keeps the name
// as a forwarding header.
#ifndef GLU_GL_UTILS_HPP
#define GLU_GL_UTILS_HPP
#include "hip_utils.hpp"
#endif
