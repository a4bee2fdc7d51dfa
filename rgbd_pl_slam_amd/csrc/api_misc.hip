// api_misc.hip -- small C-ABI utilities.
#include <string.h>
#include "plf_common.h"

extern "C" const char *plf_version(void) { return "plf 0.1 (gfx950)"; }

extern "C" const char *plf_status_string(int status)
{
    switch (status) {
        case PLF_OK: return "ok";
        case PLF_E_EMPTY: return "empty input";
        case PLF_E_BADARG: return "bad argument / unsupported size";
        case PLF_E_CAPACITY: return "output capacity too small";
        case PLF_E_HIP: return "HIP runtime error";
        case PLF_E_NOMEM: return "out of memory";
        case PLF_E_RECTS: return "more LSD rectangles than the line handle holds";
        default: return "unknown";
    }
}

extern "C" int plf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ORBmatcher::DescriptorDistance (include/ORBmatcher.h:44, so@0x79d20; Thirdparty/DBoW2/DBoW2/FORB.cpp:82-102):
// 256-bit Hamming distance of two descriptors in HOST memory (scalar utility used by host-side callers).
extern "C" int plf_hamming256(const uint8_t *a, const uint8_t *b)
{
    int d = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        memcpy(&x, a + 8 * i, 8);
        memcpy(&y, b + 8 * i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}
