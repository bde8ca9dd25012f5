#!/bin/bash
# usage: bash tools/asan_book_host.sh   (CPU only)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
printf '#include <string>\n#include <cstdio>\nint adr_set_error(int status, const std::string& msg) { fprintf(stderr, "err %%d %%s\\n", status, msg.c_str()); return status; }\n' > $T/stub.cpp
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -fPIC -shared -pthread -I$ROOT/include $ROOT/adrates_amd/csrc/book_host.cpp $T/stub.cpp -o $T/libbook_asan.so
LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 ADR_ASAN_LIB=$T/libbook_asan.so python3 $ROOT/tools/asan_book_host.py
rm -rf $T
