# Development ablations of code_parts_mfma_kernel on config 3 (GPU box): bash tools/abl_coder.sh   (edit the list of -DTSR_PM_ABL values below)
# NB: this REBUILDS the library with ablation flags -- results are wrong by design; meant for the throw-away copy of the repo a
# gpurun call works on.  Anywhere else, run `python template-speech-recognition_amd/tsr_hip/build.py --force` afterwards.
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for A in 0; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fvisibility=hidden -DTSR_PM_ABL=$A -Iinclude -Itemplate-speech-recognition_amd/csrc -c template-speech-recognition_amd/csrc/parts_mfma.hip -o template-speech-recognition_amd/lib/obj/parts_mfma.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o template-speech-recognition_amd/lib/libtsr_hip.so template-speech-recognition_amd/lib/obj/*.o -ldl
  echo "ABL=$A: $(bash tools/kstat3.sh 1000 bits 3 | grep code_parts_mfma | cut -d, -f2-4 | sed 's/.*PmArgs)",//')"
done
