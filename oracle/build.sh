#!/bin/bash
# Builds the oracle's plain-C pieces (test infrastructure).  Run from anywhere.
set -e
cd "$(dirname "$0")"
gcc -O2 -fno-fast-math -ffp-contract=off -shared -fPIC nms_ref.c -o libnms_ref.so
echo "built oracle/libnms_ref.so"
