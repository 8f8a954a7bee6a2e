mkdir -p gpurun_out
export TSR_HIP_LIB=$PWD/template-speech-recognition_amd/lib/libtsr_stamp.so
TSR_STAMP_OUT=gpurun_out/stamp_tap2.txt timeout -k 10 120 python tools/time_fast.py 8 1 > gpurun_out/stamp_tap2.log 2>&1 && python tools/stamp_report.py gpurun_out/stamp_tap2.txt alt
