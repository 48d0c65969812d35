// api_common.hpp -- shared by the amt_gpu*.hip translation units that implement the C ABI.
#pragma once

#include <algorithm>
#include <exception>
#include <mutex>
#include <string>

#include "decisions.hpp"
#include "engine.hpp"
#include "logo_model.hpp"

struct AmtGpuLogo {
    amt::LogoPlanes planes;
};

// run f(); on any exception keep the message on the context and return 0 (no exceptions cross the ABI)
template <typename F> inline int guard(AmtGpuContext* c, F&& f)
{
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
