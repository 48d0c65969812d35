#!/bin/bash
# tests/golden/make_amts_ref.sh -- regenerates amts_ref_layout.json and amts_ref_sample.dat from the REFERENCE's own struct definitions
# (oracle/build_ref.sh compiles them into oracle/_ref/layout_probe; see oracle/ref_shim/layout_probe.cpp).  Needs /root/reference; the two
# outputs are committed so the pin also holds where the reference is absent (the GPU box).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
bash "$HERE/../../oracle/build_ref.sh"
"$HERE/../../oracle/_ref/layout_probe" layout > "$HERE/amts_ref_layout.json"
"$HERE/../../oracle/_ref/layout_probe" sample "$HERE/amts_ref_sample.dat"
echo "wrote $HERE/amts_ref_layout.json $HERE/amts_ref_sample.dat"
