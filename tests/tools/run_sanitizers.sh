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
# ThreadSanitizer over four concurrent model constructions + score_create calls (persistent host team, its
# fall-back threads, the parallel row builders); OMP_NUM_THREADS=1 keeps the uninstrumented libgomp out of the report
g++ -O1 -g -fsanitize=thread -fno-omit-frame-pointer -fopenmp -std=c++17 -shared -fPIC \
    -o "$OUT/libscore_cpu_tsan.so" oracle/cpu_twin/score_cpu.cpp
SCORE_TSAN_LIB="$OUT/libscore_cpu_tsan.so" LD_PRELOAD=$(g++ -print-file-name=libtsan.so) \
    TSAN_OPTIONS="report_signal_unsafe=0 history_size=4 halt_on_error=1" OMP_NUM_THREADS=1 python tests/tools/tsan_workload.py
