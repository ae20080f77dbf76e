#!/bin/bash
# builds the stamp variant of the library (CAFE_PROF) IN PLACE of cafe_amd/libcafe_hip.so; `make -C cafe_amd/csrc clean all` restores the normal one
set -e
cd "$(dirname "$0")/../cafe_amd/csrc"
make clean >/dev/null
make -j8 EXTRA=-DCAFE_PROF 2>&1 | grep -E "error|warning: unused" || true
ls -la ../libcafe_hip.so
