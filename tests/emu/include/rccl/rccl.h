// TEST INFRASTRUCTURE -- the declarations of RCCL that galah_amd/csrc/comm.cpp names (it reaches the library through
// dlopen/dlsym only, so the emulated build needs the types and prototypes, not the library).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
extern "C" {
typedef struct ncclComm *ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5,
               ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId *id);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommAbort(ncclComm_t comm);
ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t *async_error);
const char *ncclGetErrorString(ncclResult_t r);
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclSend(const void *send, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void *recv, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
}
