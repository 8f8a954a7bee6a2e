# Build ablation variants of the fast kernel (locally) or time them (on the GPU box).
#   bash tools/ablate.sh build "0 1 2 4 8 5 13 15"
#   bash tools/ablate.sh run   "0 1 2 4 8 5 13 15"
cd ${GRAFT_REPO_ROOT:-/root/repo}
PKG=template-speech-recognition_amd
if [ "$1" = build ]; then
  for n in $2; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -fvisibility=hidden -Wno-unused-parameter -DTSR_ABL=$n \
      -o $PKG/lib/libtsr_abl$n.so $PKG/csrc/*.hip -ldl &
  done; wait; ls $PKG/lib
else
  for n in $2; do
    echo -n "ABL=$n  "
    TSR_HIP_LIB=$PWD/$PKG/lib/libtsr_abl$n.so timeout -k 5 120 python tools/time_fast.py ${3:-}
  done
fi
