# Development ablations of fast_bank_kernel on config 3's D = 2050 bank (GPU box): bash tools/abl_bank.sh "0 1 2 4 8"
# The ablated objects and library go to lib/abl/ (results of such a build are wrong by design); the shipped
# lib/libtsr_hip.so and lib/obj/ are never touched, the timing tool is pointed at the ablated library through TSR_HIP_LIB.
# Compiler output is shown and a failed compile stops the run.
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..} && mkdir -p gpurun_out
C=template-speech-recognition_amd/csrc; O=template-speech-recognition_amd/lib/obj; A_DIR=template-speech-recognition_amd/lib/abl
mkdir -p $A_DIR
python template-speech-recognition_amd/tsr_hip/build.py > /dev/null   # the unablated objects the link takes the rest from
for A in ${1:-0 1 2 4 8}; do
  pids=""
  for f in detect_mfma_i0 detect_mfma_i1 detect_mfma_i2 detect_mfma_i3 detect_mfma_i4 detect_mfma_i5; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fvisibility=hidden -DTSR_ABL=$A -Iinclude -I$C -c $C/$f.hip -o $A_DIR/$f.hip.o &
    pids="$pids $!"
  done
  for p in $pids; do wait $p; done   # (set -e: a failed compile ends the script here)
  objs=$(for o in $O/*.o; do b=$(basename $o); if [ -f $A_DIR/$b ]; then echo $A_DIR/$b; else echo $o; fi; done)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $A_DIR/libtsr_abl.so $objs -ldl
  echo "ABL=$A: $(TSR_HIP_LIB=$PWD/$A_DIR/libtsr_abl.so python tools/time_config3.py 1000 bits 3 2>&1 | grep layout=bits | sed 's/.*D=2050 bank \([0-9.]*\) ms.*/bank \1 ms/')"
done
