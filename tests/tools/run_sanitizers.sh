#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the host code the product shares with the CPU twin (setup,
# assembler, refinement pattern builder, thread team, ADMM driver) -- CPU build only (no GPU sanitizers on this pool):
#   bash tests/tools/run_sanitizers.sh
set -e
cd "$(dirname "$0")/../.."
OUT=${TMPDIR:-/tmp}/score_asan
mkdir -p "$OUT"
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp -std=c++17 -shared -fPIC \
    -o "$OUT/libscore_cpu_asan.so" oracle/cpu_twin/score_cpu.cpp
SCORE_ASAN_LIB="$OUT/libscore_cpu_asan.so" LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 \
    OMP_NUM_THREADS=2 python tests/tools/sanitizer_workload.py
