// stands in for <rccl/rccl.h> in the host emulation build: a one-rank "communicator" whose all-gather is a copy
#pragma once
#include <cstring>
typedef int ncclResult_t;
enum { ncclSuccess = 0 };
typedef void* ncclComm_t;
enum ncclDataType_t { ncclDouble = 8 };
inline const char* ncclGetErrorString(ncclResult_t) { return "wave_emu: rccl stub"; }
inline ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t, void*) { memmove(recv, send, count * sizeof(double)); return ncclSuccess; }
