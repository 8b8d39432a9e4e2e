// kz_engine_util.hpp — error plumbing of the engine translation unit (kz_engine.hip and the headers it is split into):
// the thread-local message behind kz_last_error(), HIP_TRY, the host-side f16 conversion.  Included ONCE, inside
// kz_engine.hip's anonymous namespace.
#pragma once

thread_local std::string g_err;

int fail(const std::string &msg) {
    g_err = msg;
    return 1;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(std::string(#expr) + " failed: " + hipGetErrorString(e_) + " (" __FILE__ ":" + \
                        std::to_string(__LINE__) + ")");                                               \
    } while (0)

using kz::Conv;
using kz::Linear;
using kz::Model;
using kz::round_up;

uint16_t f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f;  // round-to-nearest-even, same conversion the device uses
    uint16_t b;
    memcpy(&b, &h, 2);
    return b;
}
