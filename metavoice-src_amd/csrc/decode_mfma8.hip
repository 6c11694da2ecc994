// decode_mfma_kernel<8> as its own translation unit: the Makefile compiles this file with -mllvm -amdgpu-sched-strategy=max-memory-clause (see the
// note at the end of decode_mfma.hip); everything is in decode_mfma.hip.
#define MF_INSTANCE 8
#include "decode_mfma.hip"
