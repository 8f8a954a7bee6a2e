# usage (on the GPU box): bash tools/stamp_run.sh [tap2|old]   (library from tools/build_stamp.sh)
mkdir -p gpurun_out
export TSR_HIP_LIB=$PWD/template-speech-recognition_amd/lib/libtsr_stamp.so
K=${1:-tap2}
if [ "$K" = tap2 ]; then export TSR_FASTK=tap2; R=alt; else R=old; fi
TSR_STAMP_OUT=gpurun_out/stamp_$K.txt timeout -k 10 120 python tools/time_fast.py 8 1 > gpurun_out/stamp_$K.log 2>&1 && python tools/stamp_report.py gpurun_out/stamp_$K.txt $R
