/* libxsmm_macros.h -- the reference's macro header; the macros its samples and tests use live in libxsmm_utils.h. */
#ifndef LIBXSMM_MACROS_H_ALIAS
#define LIBXSMM_MACROS_H_ALIAS
#include "libxsmm_utils.h"
#endif
