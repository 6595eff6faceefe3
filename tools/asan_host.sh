#!/bin/bash
# CPU test suite against a build of libsd_hip.so whose HOST code is compiled with AddressSanitizer (device code as
# usual; GPU sanitizers are not available on this pool).  Covers what runs without a GPU: FASTA index, chunk tables,
# layout plans, seam merge, TSV text, file output (write_parts), the final / _alt post-processing with the host
# identities.  usage: bash tools/asan_host.sh   (a few minutes; needs the normal build in csrc/_obj for the kernel units)
set -e
C=stringdecomposer_amd/csrc
T=$(mktemp -d)
for f in sd_engine sd_stream sd_run_files sd_range_asm sd_host_api sd_post sd_convert sd_nw sd_generic sd_filter sd_ident; do
  hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -c $C/$f.hip -o $T/$f.o &
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address -shared-libasan -o $T/libsd_hip_asan.so $T/*.o \
  $(ls $C/_obj/*.o | grep -v "sd_engine.o\|sd_stream.o\|sd_run_files.o\|sd_range_asm.o\|sd_host_api.o\|sd_post.o\|sd_convert.o\|sd_nw.o\|sd_generic.o\|sd_filter.o\|sd_ident.o") -lpthread
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$RT SD_HIP_LIB=$T/libsd_hip_asan.so python -m pytest tests -q -m "not gpu" -k "not gloo"
rm -rf $T
