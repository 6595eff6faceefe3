#!/bin/bash
# AddressSanitizer + UBSan run of the host identity code (csrc/sd_post.hip is plain C++: block traceback, Hirschberg's
# split, byte alphabets) on CPU: long pairs, empty sequences, all 255 byte values.  usage: bash tools/asan_post.sh
set -e
T=$(mktemp -d)
g++ -x c++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -c stringdecomposer_amd/csrc/sd_post.hip -o $T/sd_post.o
g++ -std=c++17 -O1 -g -fsanitize=address,undefined tools/scratch/asan_post.cpp $T/sd_post.o -o $T/asan_post -lpthread
$T/asan_post
rm -rf $T
