# Development ablations of code_parts_mfma_kernel on config 3 (GPU box): bash tools/abl_coder.sh "0 1 2"   (-DTSR_PM_ABL values)
# The ablated object and library go to lib/abl/ (results of such a build are wrong by design); the shipped
# lib/libtsr_hip.so and lib/obj/ are never touched, the timing tool is pointed at the ablated library through TSR_HIP_LIB.
# Compiler output is shown and a failed compile stops the run.
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..} && mkdir -p gpurun_out
O=template-speech-recognition_amd/lib/obj; A_DIR=template-speech-recognition_amd/lib/abl
mkdir -p $A_DIR
python template-speech-recognition_amd/tsr_hip/build.py > /dev/null
for A in ${1:-0}; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fvisibility=hidden -DTSR_PM_ABL=$A -Iinclude -Itemplate-speech-recognition_amd/csrc -c template-speech-recognition_amd/csrc/parts_mfma.hip -o $A_DIR/parts_mfma.hip.o
  objs=$(for o in $O/*.o; do b=$(basename $o); if [ -f $A_DIR/$b ]; then echo $A_DIR/$b; else echo $o; fi; done)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $A_DIR/libtsr_abl.so $objs -ldl
  echo "ABL=$A: $(TSR_HIP_LIB=$PWD/$A_DIR/libtsr_abl.so bash tools/kstat3.sh 1000 bits 3 | grep code_parts_mfma | cut -d, -f2-4 | sed 's/.*PmArgs)",//')"
done
