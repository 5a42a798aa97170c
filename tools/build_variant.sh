# Dev tool: build a variant of libm2d_hip.so into music2dance_amd/lib_<tag>/ with extra hipcc flags (M2D_LIB=... selects it)
#   bash tools/build_variant.sh tune -DM2D_TUNING        bash tools/build_variant.sh stamp -DM2D_STAMP -DM2D_TUNING
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/music2dance_amd/lib_$TAG
mkdir -p $OUT
pids=""
for f in m2d_runtime gemm_engine conv1d tcn conv1d_thin bn gru pointwise; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c $ROOT/music2dance_amd/csrc/$f.hip -o $OUT/$f.o &
  pids="$pids $!"
done
for p in $pids; do wait $p || exit 1; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libm2d_hip.so $OUT/*.o && echo built $OUT/libm2d_hip.so
