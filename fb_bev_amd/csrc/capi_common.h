// capi_common.h -- helpers shared by the translation units of the C ABI (capi.hip, capi_train.hip)
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "rt.h"

#define FBBEV_CHECK_LAUNCH()                      \
    do {                                          \
        int e_ = fbbev_rt_last_error();           \
        if (e_ != 0) return e_;                   \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
