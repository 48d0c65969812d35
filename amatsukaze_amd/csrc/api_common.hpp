// api_common.hpp -- shared by the amt_gpu*.hip translation units that implement the C ABI.
#pragma once

#include <algorithm>
#include <exception>
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
