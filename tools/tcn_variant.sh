# Dev tool: build a variant of csrc/tcn.hip only and link it with the product's other objects into
# music2dance_amd/lib_<tag>/libm2d_hip.so (M2D_LIB=... selects it):  bash tools/tcn_variant.sh nosync -DM2D_STAMP -DTCN_X_NOSYNC
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/music2dance_amd/lib_$TAG
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c $ROOT/music2dance_amd/csrc/tcn.hip -o $OUT/tcn.o || exit 1
OBJS=$(ls $ROOT/music2dance_amd/lib/obj/*.o | grep -v "/tcn.o\|\.stamp\.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libm2d_hip.so $OBJS $OUT/tcn.o && rm -f $OUT/tcn.o && echo built $OUT/libm2d_hip.so
