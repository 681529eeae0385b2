/* libxsmm_math.h -- the reference's math/RNG/conversion header; those helpers are declared in libxsmm_utils.h. */
#ifndef LIBXSMM_MATH_H_ALIAS
#define LIBXSMM_MATH_H_ALIAS
#include "libxsmm_utils.h"
#endif
