// monodetr_amd/csrc/mdetr_tune.h -- the ONE environment variable through which tests force a launch geometry.
//
//   MDETR_TUNE="tgemm_tile=128x64,tgemm_pf=1,msda_tile_w=16"
//
// The launchers choose tiles, chunk counts and wave counts by fixed rules (no run-time search).  Every template those rules can
// select has to be exercised by the tests, also at sizes where the rule would not pick it: a test sets the key, the launcher
// takes the value instead of its rule's.  Nothing in the product sets it; the keys that exist are the `tune_*("...")` call sites
// (DESIGN.md lists them).  Kernel FAMILIES are switched by monodetr_amd/kernel_families.py, not here.
#pragma once
#include <stdlib.h>
#include <string.h>

namespace mdetr {

// the value of `key` in MDETR_TUNE, copied into buf (<= n - 1 characters), or nullptr
inline const char *tune_str(const char *key, char *buf, size_t n)
{
    const char *env = getenv("MDETR_TUNE");
    if (!env || !*env || n == 0) return nullptr;
    const size_t klen = strlen(key);
    for (const char *p = env; *p;) {
        const char *end = strchr(p, ',');
        const size_t len = end ? static_cast<size_t>(end - p) : strlen(p);
        if (len > klen && strncmp(p, key, klen) == 0 && p[klen] == '=') {
            size_t m = len - klen - 1;
            if (m > n - 1) m = n - 1;
            memcpy(buf, p + klen + 1, m);
            buf[m] = 0;
            return buf;
        }
        if (!end) break;
        p = end + 1;
    }
    return nullptr;
}

inline int tune_int(const char *key, int dflt)
{
    char buf[32];
    const char *v = tune_str(key, buf, sizeof(buf));
    return v && *v ? atoi(v) : dflt;
}

}  // namespace mdetr
