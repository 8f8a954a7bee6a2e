# Development ablations of mx_bank_kernel (GPU box or here to build): bash tools/abl_mx.sh build "1 2" | run "1 2"
# The ablated objects and libraries go to lib/abl/ (results of such builds are wrong by design); the shipped library is never touched.
set -e
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
P=template-speech-recognition_amd; mkdir -p $P/lib/abl
if [ "$1" = build ]; then
  python $P/tsr_hip/build.py > /dev/null
  for A in $2; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fvisibility=hidden -DMX_ABL=$A -Iinclude -c $P/csrc/detect_mx.hip -o $P/lib/abl/detect_mx_$A.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/abl/libmx_abl$A.so $(ls $P/lib/obj/*.o | grep -v detect_mx) $P/lib/abl/detect_mx_$A.o -ldl
  done
else
  for A in $2; do echo -n "MX_ABL=$A: "; TSR_HIP_LIB=$PWD/$P/lib/abl/libmx_abl$A.so timeout -k 5 120 python tools/time_fast.py 8 9 | tail -1; done
fi
