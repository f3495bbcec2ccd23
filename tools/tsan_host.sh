#!/bin/bash
# ThreadSanitizer over the host threads of the product: the chunk-parallel loaders (eight threads, pieces forced on the tiny
# fixture) and the file phase (cgx_assemble_files, six threads).  CPU only; builds cgx_host.c with -fsanitize=thread and links the
# harness tests/cpu_sim/host_threads.c to it and to the device object, as the product is linked.  usage: tools/tsan_host.sh
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; OUT="${CGX_TSAN_DIR:-/tmp/cgx_tsan}"; mkdir -p "$OUT/files"
make -s -C "$ROOT/cgx_amd/csrc" cgx_device.o
gcc -O1 -g -std=gnu11 -fPIC -fno-math-errno -fsanitize=thread -c "$ROOT/cgx_amd/csrc/cgx_host.c" -o "$OUT/cgx_host.o"
gcc -O1 -g -std=gnu11 -fsanitize=thread -I"$ROOT/include" -c "$ROOT/tests/cpu_sim/host_threads.c" -o "$OUT/host_threads.o"
hipcc --offload-arch=gfx950 -fsanitize=thread "$OUT/host_threads.o" "$OUT/cgx_host.o" "$ROOT/cgx_amd/csrc/cgx_device.o" -lm -lpthread -lz -o "$OUT/host_threads" 2>/dev/null
CGX_THREADS=8 CGX_LOAD_PIECE_MIN=1 TSAN_OPTIONS="halt_on_error=1" "$OUT/host_threads" "$ROOT/tests/golden/tiny" "$OUT/files"
