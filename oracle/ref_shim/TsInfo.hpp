// oracle/ref_shim/TsInfo.hpp -- TEST INFRASTRUCTURE ONLY: LogoScan.hpp includes it but uses nothing from it.
#pragma once
