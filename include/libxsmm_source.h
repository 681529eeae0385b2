/* libxsmm_source.h -- the reference's header-only entry point (the whole library compiled into the including translation unit).
 * This back end is a shared library: the header maps to the same declarations and the program links libxsmm_amd.so. */
#ifndef LIBXSMM_SOURCE_H_ALIAS
#define LIBXSMM_SOURCE_H_ALIAS
#include "libxsmm_utils.h"
#endif
