// api_common.hpp -- shared by the amt_gpu*.hip translation units that implement the C ABI.
#pragma once

#include <algorithm>
#include <cstdint>
#include <exception>
#include <mutex>
#include <string>

#include "decisions.hpp"
#include "engine.hpp"
#include "logo_model.hpp"

struct AmtGpuLogo {
    amt::LogoPlanes planes;
};

// UTF-16 (a Windows wchar_t string, NUL-terminated: what the reference's exports take, LogoScan.hpp:1083-1086) -> UTF-8.  A lone
// surrogate becomes U+FFFD: the result is always valid UTF-8.
inline std::string amt_utf8_from_utf16(const uint16_t* s, size_t n)
{
    std::string o;
    auto put = [&](uint32_t c) {
        if (c < 0x80) o += (char)c;
        else if (c < 0x800) { o += (char)(0xC0 | (c >> 6)); o += (char)(0x80 | (c & 0x3F)); }
        else if (c < 0x10000) { o += (char)(0xE0 | (c >> 12)); o += (char)(0x80 | ((c >> 6) & 0x3F)); o += (char)(0x80 | (c & 0x3F)); }
        else { o += (char)(0xF0 | (c >> 18)); o += (char)(0x80 | ((c >> 12) & 0x3F)); o += (char)(0x80 | ((c >> 6) & 0x3F)); o += (char)(0x80 | (c & 0x3F)); }
    };
    for (size_t i = 0; i < n; ++i) {
        uint32_t c = s[i];
        if (c >= 0xD800 && c < 0xDC00 && i + 1 < n && s[i + 1] >= 0xDC00 && s[i + 1] < 0xE000) {
            c = 0x10000 + ((c - 0xD800) << 10) + (s[i + 1] - 0xDC00);
            ++i;
        } else if (c >= 0xD800 && c < 0xE000) {
            c = 0xFFFD;
        }
        put(c);
    }
    return o;
}
inline std::string amt_utf8_from_utf16z(const uint16_t* s)
{
    if (!s) return std::string();
    size_t n = 0;
    while (s[n]) ++n;
    return amt_utf8_from_utf16(s, n);
}

// run f(); on any exception keep the message on the context and return 0 (no exceptions cross the ABI)
template <typename F> inline int guard(AmtGpuContext* c, F&& f, const char* caller = __builtin_FUNCTION())
{
    AMT_TRACE_SCOPE(caller);
    (void)caller;
    // calls on one context are serialised: its stream, staging ring, error string and timing spans are shared state
    std::unique_lock<std::recursive_mutex> lk;
    if (c) lk = std::unique_lock<std::recursive_mutex>(c->mu);
    try {
        f();
        return 1;
    } catch (const std::exception& e) {
        if (c) c->err = e.what();
    } catch (...) {
        if (c) c->err = "unknown error";
    }
    return 0;
}
