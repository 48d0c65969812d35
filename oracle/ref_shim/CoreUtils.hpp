// oracle/ref_shim/CoreUtils.hpp -- TEST INFRASTRUCTURE ONLY: AMTLogo.hpp includes "CoreUtils.hpp";
// everything it needs lives in the shim TranscodeSetting.hpp.
#pragma once
#include "TranscodeSetting.hpp"
