# Dev tool: build a variant of ONE source of csrc/ and link it with the product's other objects into
# music2dance_amd/lib_<tag>/libm2d_hip.so (M2D_LIB=... selects it):  bash tools/file_variant.sh nostore conv1d_thin -DTHIN_X_NOSTORE
TAG=$1; FILE=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/music2dance_amd/lib_$TAG
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c $ROOT/music2dance_amd/csrc/$FILE.hip -o $OUT/$FILE.o || exit 1
OBJS=$(ls $ROOT/music2dance_amd/lib/obj/*.o | grep -v "/$FILE.o\|\.stamp\.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libm2d_hip.so $OBJS $OUT/$FILE.o && rm -f $OUT/$FILE.o && echo built $OUT/libm2d_hip.so
