#!/bin/bash
# usage: tools/build_variant.sh NAME "EXTRA FLAGS"  -> tools/dbglibs/NAME.so (objects under csrc/build_NAME); the product
# library is not touched.  Select a variant at run time with FU_LIB_PATH (floodplanet_code_amd/_lib.py, tools/ab_libs.sh).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/dbglibs
make -C floodplanet_code_amd/csrc -j6 BUILD=build_$1 LIB=../../tools/dbglibs/$1.so EXTRA="$2" 2>&1 | grep -E "error|warning: v|spill" || true
ls -la tools/dbglibs/$1.so
