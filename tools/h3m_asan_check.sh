#!/bin/bash
# Sanitizer pass over the hor3map column routines (CPU build only; GPU ASan is not available on this pool):
# the device code of blom_amd/csrc/hor3map_*.h compiled for the host with AddressSanitizer, run over the
# adversarial and the model-like column sets.  Every per-column array is allocated at its exact size.
set -e
cd "$(dirname "$0")/../tests/hostcheck"
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address \
      -shared-libasan -shared -o /tmp/libh3m_hostcheck_asan.so h3m_hostcheck.hip 2>&1 | grep -v "Woption-ignored\|ignoring" || true
RT=$(find /opt/rocm/lib/llvm -name 'libclang_rt.asan-x86_64.so' | head -1)
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD=$RT python3 ../../tools/h3m_asan_cases.py
