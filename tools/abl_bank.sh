# NB: this REBUILDS the library with ablation flags -- results are wrong by design; meant for the throw-away copy of the repo a
# gpurun call works on.  Anywhere else, run `python template-speech-recognition_amd/tsr_hip/build.py --force` afterwards.
# Development ablations of fast_bank_kernel on config 3's D = 2050 bank (GPU box): bash tools/abl_bank.sh "0 1 2 4 8"
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
C=template-speech-recognition_amd/csrc; O=template-speech-recognition_amd/lib/obj
for A in ${1:-0 1 2 4 8}; do
  for f in detect_mfma_i0 detect_mfma_i1 detect_mfma_i2 detect_mfma_i3 detect_mfma_i4 detect_mfma_i5; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fvisibility=hidden -DTSR_ABL=$A -Iinclude -I$C -c $C/$f.hip -o $O/$f.hip.o 2>/dev/null &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o template-speech-recognition_amd/lib/libtsr_hip.so $O/*.o -ldl
  echo "ABL=$A: $(python tools/time_config3.py 1000 bits 3 2>&1 | grep layout=bits | sed 's/.*D=2050 bank \([0-9.]*\) ms.*/bank \1 ms/')"
done
