# Build ablation variants of the block-Toeplitz kernel (locally) or time them (on the GPU box).
#   bash tools/ablate_toep.sh build "0 1 2 4"     bash tools/ablate_toep.sh run "0 1 2 4"
cd ${GRAFT_REPO_ROOT:-/root/repo}
PKG=template-speech-recognition_amd
if [ "$1" = build ]; then
  for n in $2; do
    ( /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-parameter -DTSR_TABL=$n -DTSR_WABL=$n \
        -c $PKG/csrc/${SRC:-detect_ws}.hip -o /tmp/toep_abl$n.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/lib/libtsr_tabl$n.so \
        $(ls $PKG/lib/obj/*.o | grep -v ${SRC:-detect_ws}.hip.o) /tmp/toep_abl$n.o -ldl ) &
  done; wait; ls $PKG/lib
else
  for n in $2; do
    echo -n "TABL=$n  "
    TSR_HIP_LIB=$PWD/$PKG/lib/libtsr_tabl$n.so timeout -k 5 120 python tools/time_fast.py ${3:-}
  done
fi
