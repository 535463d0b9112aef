#!/bin/bash
# packed frame kernel: waves per SIMD asked of the register allocator (ab/lib_w4_16.so: four up to the 16-wide window, ab/lib_w4_24.so: up to 24)
for lib in "" ab/lib_w4_16.so ab/lib_w4_24.so ""; do
  echo "== lib: ${lib:-product}"
  if [ -n "$lib" ]; then export OPS_AMD_LIB=$PWD/$lib; else unset OPS_AMD_LIB; fi
  timeout 300 python scripts/frame_pack_check.py ab 3x3:65536 4x4:32768 9x4:16384 5x5:32768 6x6:16384 7x7:16384 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l, end=''); continue
    print(r['frame'], r['half_bandwidth'], 'pack_ms', r['pack_ms'], 'solves/s %.3e' % r['pack_solves_per_s'])
"
done
