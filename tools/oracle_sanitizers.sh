#!/bin/bash
# AddressSanitizer + UBSan over the CPU build (the oracle, which shares models_harness.inc,
# model_functors.h, slab_logic.inc and polarity.cuh with the device build): builds an instrumented
# liboracle_models.so, runs the oracle-only tests against it, restores the normal library.
# (GPU AddressSanitizer is not available on the MI355X pool.)
set -e
cd "$(dirname "$0")/.."
make -C oracle > /dev/null
cp oracle/_build/liboracle_models.so /tmp/oracle_normal.so
trap 'cp /tmp/oracle_normal.so oracle/_build/liboracle_models.so; touch oracle/_build/liboracle_models.so' EXIT
g++ -O1 -g -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -fsanitize=address,undefined \
    -fno-omit-frame-pointer -Ioracle -Iinclude -Iyalla_amd/csrc -shared -Wl,-Bsymbolic \
    -o oracle/_build/liboracle_models.so oracle/oracle_models.cpp
touch oracle/_build/liboracle_models.so
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
python -m pytest tests/test_oracle_kats.py tests/test_golden.py tests/test_growth.py \
    tests/test_model_functors_independent.py tests/test_heun_independent.py tests/test_slab.py -q -m "not gpu" -k "not gloo" -p no:cacheprovider
