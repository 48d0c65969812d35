#!/bin/bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE ONLY.
# Compiles the REAL reference sources of the logo path where they lie under /root/reference
#   Amatsukaze/ComputeKernel.cpp   (CalcCorrelation5x5_AVX, IsAVXAvailable)
#   Amatsukaze/LogoScan.hpp        (LogoDataParam, LogoScan, LogoAnalyzer/ScanLogo, AMTAnalyzeLogo,
#                                   AMTEraseLogo, LogoFrame)
#   Amatsukaze/AMTLogo.hpp, include/logo.h
# through the stand-in host headers in oracle/ref_shim/ (mock AviSynth / FFmpeg / UtVideo / Win32
# plumbing, no arithmetic) into oracle/_ref/libamt_ref.so.  No reference source is copied: the
# stage directory holds symlinks only, and is removed after the build.  Outputs only under oracle/_ref/
# (git-ignored, shipped to the GPU box by gpurun like any built .so).
# The reference's own build system (MSVC .vcxproj) is not run.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${AMT_REFERENCE_DIR:-/root/reference}"
OUT="$HERE/_ref"
if [ ! -d "$REF/Amatsukaze" ]; then
  echo "build_ref.sh: $REF not present (GPU box?) -- keeping any prebuilt $OUT/libamt_ref.so" >&2
  exit 0
fi
mkdir -p "$OUT"
STAGE="$(mktemp -d "$OUT/stage.XXXXXX")"
trap 'rm -rf "$STAGE"' EXIT
ln -s "$REF/Amatsukaze/LogoScan.hpp"      "$STAGE/LogoScan.hpp"
ln -s "$REF/Amatsukaze/AMTLogo.hpp"       "$STAGE/AMTLogo.hpp"
ln -s "$REF/Amatsukaze/ComputeKernel.cpp" "$STAGE/ComputeKernel.cpp"
ln -s "$REF/include/logo.h"               "$STAGE/logo.h"
for f in TranscodeSetting.hpp CoreUtils.hpp TsInfo.hpp TextOut.h intrin.h ref_driver.cpp; do
  ln -s "$HERE/ref_shim/$f" "$STAGE/$f"
done
CXX="${CXX:-g++}"
# MSVC v140 /O2 /fp:precise: IEEE fp32, no contraction; /arch:AVX only on ComputeKernel.cpp
# (Amatsukaze.vcxproj:124-147,224-231)
COMMON="-O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mno-fma -w"
"$CXX" $COMMON -mavx -mxsave -I"$STAGE" -c "$STAGE/ComputeKernel.cpp" -o "$STAGE/ComputeKernel.o"
"$CXX" $COMMON -I"$STAGE" -c "$STAGE/ref_driver.cpp" -o "$STAGE/ref_driver.o"
"$CXX" -shared -o "$OUT/libamt_ref.so" "$STAGE/ComputeKernel.o" "$STAGE/ref_driver.o"
echo "built $OUT/libamt_ref.so"

# layout probe (oracle/ref_shim/layout_probe.cpp): the amts struct definitions, cut by name out of the reference headers into the stage
# directory (removed with it), compiled with the reference's 2-byte wchar_t
cut_type() { iconv -f CP932 -t UTF-8 "$1" | tr -d '\r' | awk -v n="$2" '$0 ~ "^(enum|struct) "n"( |\\{|$)" && !fin {on=1} on{print} on && /^\};/ {on=0; fin=1}'; }
{
  for n in DECODER_TYPE DecoderSetting CMType VIDEO_STREAM_FORMAT VideoFormat AUDIO_CHANNELS AudioFormat; do
    cut_type "$REF/Amatsukaze/StreamUtils.hpp" "$n"
  done
  for n in FilterSourceFrame FilterAudioFrame; do cut_type "$REF/Amatsukaze/StreamReform.hpp" "$n"; done
} > "$STAGE/ref_types.inc"
"$CXX" -O1 -std=c++17 -fshort-wchar -w -I"$STAGE" "$HERE/ref_shim/layout_probe.cpp" -o "$OUT/layout_probe"
echo "built $OUT/layout_probe"
