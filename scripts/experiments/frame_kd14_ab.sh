for lib in "" ab/lib_kd14.so; do
  echo "== lib: ${lib:-product}"
  if [ -n "$lib" ]; then export OPS_AMD_LIB=$PWD/$lib; fi
  timeout 300 python scripts/frame_pack_check.py ab 4x4:32768 9x4:16384 3x10:16384 4x10:16384 5x5:32768 2>&1 | grep -v amdgpu.ids
done
export OPS_AMD_LIB=$PWD/ab/lib_kd14.so
timeout 300 python scripts/frame_pack_check.py 2>&1 | grep " kd 14 \|worst"
