/* dlaf_c/init.h -- runtime start/stop.
 * Drop-in for the reference's include/dlaf_c/init.h:27,35 (src/c_api/init.cpp:21-57).  The
 * reference starts pika and leaves it suspended; here initialization selects the GPU of this
 * process (LOCAL_RANK, one process per GPU), and prepares the HIP kernels.  Both calls are
 * idempotent like upstream.  The argc/argv pairs are accepted and ignored except for the
 * "--dlaf:print-config" flag. */
#pragma once
#include <dlaf_c/utils.h>

DLAF_EXTERN_C void dlaf_initialize(int argc_pika, const char** argv_pika, int argc_dlaf,
                                   const char** argv_dlaf) DLAF_NOEXCEPT;
DLAF_EXTERN_C void dlaf_finalize(void) DLAF_NOEXCEPT;
