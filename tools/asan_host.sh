#!/bin/bash
# Runs a command against a sanitizer build of the HOST code of the product (cgx_host.c with ASan + UBSan, linked to the device
# object exactly as cgx_amd/csrc/Makefile links the product).  CPU only: the GPU pool offers no sanitizer runs.
#   tools/asan_host.sh python3 -m pytest tests/test_loaders.py tests/test_abi.py -q -m "not gpu"
#   tools/asan_host.sh python3 tools/fuzz_loaders.py --cases 400
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${CGX_ASAN_DIR:-/tmp/cgx_asan}"
mkdir -p "$OUT"
make -s -C "$ROOT/cgx_amd/csrc" cgx_device.o
gcc -O1 -g -std=gnu11 -fPIC -Wall -Wextra -fno-math-errno -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize=alignment \
    -c "$ROOT/cgx_amd/csrc/cgx_host.c" -o "$OUT/cgx_host.o"
hipcc --offload-arch=gfx950 -shared -fPIC "$ROOT/cgx_amd/csrc/cgx_device.o" "$OUT/cgx_host.o" -lm -lpthread -lz -o "$OUT/libcgx_hip.so"
export CGX_LIB="$OUT/libcgx_hip.so"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:${ASAN_OPTIONS}"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:${UBSAN_OPTIONS}"
exec "$@"
