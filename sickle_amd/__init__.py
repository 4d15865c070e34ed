"""sickle_amd -- MI355X-native drop-in for the sliding-window quality scan of
pentalpha/sickle (reference src/trim.cpp:3-116) behind its `sickle se` / `sickle pe` CLI.

The product is native: sickle_amd/csrc/ holds the HIP kernels, the C-ABI library
(include/sickle_amd.h -> libsickle_amd.so) and the C++ host pipeline (`sickle` binary).
This Python package is only the ctypes view of that C ABI used by tests and bench.py."""
