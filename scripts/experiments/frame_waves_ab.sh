#!/bin/bash
# packed frame kernel A/B over variant libraries (scripts/build_frame_variant.sh): scripts/experiments/frame_waves_ab.sh ab/lib_a.so ab/lib_b.so ...
# (the product library first and last: box drift; SHAPES="10x10:16384 15x16:12288" selects the frames)
for lib in "" "$@" ""; do
  echo "== lib: ${lib:-product}"
  if [ -n "$lib" ]; then export OPS_AMD_LIB=$PWD/$lib; else unset OPS_AMD_LIB; fi
  timeout 300 python scripts/frame_pack_check.py ab ${SHAPES:-3x3:65536 4x4:32768 9x4:16384 5x5:32768 6x6:16384 7x7:16384 8x8:16384 9x9:16384} 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l, end=''); continue
    print(r['frame'], r['half_bandwidth'], 'pack_ms', r['pack_ms'], 'solves/s %.3e' % r['pack_solves_per_s'], '| frame_pack=0 (wave kernel) ms', r['wave_ms'])
"
done
