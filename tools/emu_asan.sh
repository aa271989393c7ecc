#!/bin/bash
# Memory-safety audit of the kernels without a GPU: build the CPU device emulator with AddressSanitizer and run the emulator
# test files against it.  Tensor buffers come from torch's CPU allocator (malloc family -> red zones), so an out-of-bounds
# read or write of any kernel on the tested shapes is reported with the kernel source line.
#   bash tools/emu_asan.sh [pytest args]        (default: every tests/test_emu_*.py and the BEVDet chain)
#   FBBEV_SAN=ubsan bash tools/emu_asan.sh      same with -fsanitize=alignment,signed-integer-overflow,bounds (every vector
#                                               load / store on its natural alignment, no overflow in the index arithmetic)
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/fbbev_emu_asan
mkdir -p "$OUT"
if [ "$FBBEV_SAN" == "ubsan" ]; then
  SAN="-fsanitize=alignment,signed-integer-overflow,bounds -fno-sanitize-recover=all"; RT=libubsan.so
else
  SAN="-fsanitize=address -fno-omit-frame-pointer"; RT=libasan.so
fi
g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off $SAN -x c++ \
    -I "$REPO/tests/emu" "$REPO/fb_bev_amd/csrc/capi.hip" -o "$OUT/libfbbev_emu_asan.so"
export LD_PRELOAD=$(gcc -print-file-name=$RT)
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1
export FBBEV_EMU_LIB="$OUT/libfbbev_emu_asan.so"
cd "$REPO"
if [ $# -gt 0 ]; then exec python -m pytest "$@"; fi
exec python -m pytest tests/test_emu_kernels.py tests/test_emu_conv3d.py tests/test_bevdet_view_transformer.py -q -x -m "not gpu"
