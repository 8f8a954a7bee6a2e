#!/bin/bash
# Development build with s_memtime stamps (-DTSR_STAMP) -> template-speech-recognition_amd/lib/libtsr_stamp.so
# (selected at run time by TSR_HIP_LIB; never the shipped library).
set -e
cd "$(dirname "$0")/../template-speech-recognition_amd"
mkdir -p lib/obj_stamp
pids=()
for f in csrc/*.hip; do
    o=lib/obj_stamp/$(basename "$f").o
    case "$f" in
    *detect_mfma*|*detect_mx*) /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fvisibility=hidden -DTSR_STAMP -c "$f" -o "$o" & pids+=($!) ;;
    *) [ "$o" -nt "$f" ] || { /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fvisibility=hidden -c "$f" -o "$o" & pids+=($!); } ;;
    esac
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libtsr_stamp.so lib/obj_stamp/*.o -ldl
echo lib/libtsr_stamp.so
