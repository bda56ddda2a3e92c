#!/bin/bash
# Runs the CPU-side oracle tests against an AddressSanitizer + UBSan build of oracle/ (the checker must not itself read out of bounds).
# The sanitized library temporarily takes the place of oracle/liboracle.so and is put back afterwards.
set -u
cd "$(dirname "$0")/.."
OUT=${OUT:-/tmp/tb_oracle_san}; mkdir -p "$OUT"
FMA=$(grep -q -m1 ' fma ' /proc/cpuinfo && echo -mfma)
(cd oracle && g++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $FMA -pthread -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -shared -o "$OUT/liboracle_san.so" tb_oracle.cpp bvh_ref.cpp post_ref.cpp rt_ref.cpp) || exit 1
make -s -C oracle && cp oracle/liboracle.so "$OUT/liboracle_orig.so" && cp "$OUT/liboracle_san.so" oracle/liboracle.so
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_oracle_known_answers.py tests/test_material_branches.py \
    tests/test_oracle_furnace.py tests/test_post_process.py tests/test_realtime_chain.py tests/test_host_scene.py tests/test_two_level.py -m "not gpu" -x -q
rc=$?
cp "$OUT/liboracle_orig.so" oracle/liboracle.so
exit $rc
