# A/B of library builds with one ABI: tools/ab.sh <mesh_n> <spp> <variant> <variant> ...   (libraries tools/bin/libpt_<variant>.so)
mesh=$1; spp=$2; shift; shift
for rep in 1 2 3; do for v in "$@"; do echo "== $v"; PT_LIB_OVERRIDE=$PWD/tools/bin/libpt_$v.so timeout -k 10 200 python tools/sweep.py $mesh $spp "[{}]" 2>&1 | grep "^{" | cut -c1-60; done; done
