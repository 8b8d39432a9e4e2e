// kz_engine_util.hpp — error plumbing of the engine translation unit (kz_engine.hip and the headers it is split into):
// the thread-local message behind kz_last_error(), HIP_TRY, the host-side f16 conversion.  Included ONCE, inside
// kz_engine.hip's anonymous namespace.
#pragma once

thread_local std::string g_err;

int fail(const std::string &msg) {
    g_err = msg;
    return 1;
}

// No C++ exception may cross the C ABI: the host on the other side is Rust (kzero_amd/rust/hip.rs), where an unwinding
// foreign exception is undefined behaviour / an abort.  Every `extern "C"` entry point of kz_engine.hip runs its body through
// this: std::bad_alloc / std::length_error from host-side packing or an absurd max_batch become a non-zero return with a
// message behind kz_last_error(), like every other failure (the shim turns non-zero into the panic cudnn.rs:29-43 raises).
template <class F>
int guarded(const char *what, F &&body, int on_exception = 1) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        try {
            g_err.assign(what);
            g_err.append(": out of host memory (std::bad_alloc)");
        } catch (...) {
        }
    } catch (const std::exception &e) {
        try {
            g_err.assign(what);
            g_err.append(": C++ exception: ");
            g_err.append(e.what());
        } catch (...) {
        }
    } catch (...) {
        try {
            g_err.assign(what);
            g_err.append(": unknown C++ exception");
        } catch (...) {
        }
    }
    return on_exception;
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
