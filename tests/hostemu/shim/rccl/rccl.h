// TEST INFRASTRUCTURE -- the handful of RCCL calls comm_rccl.hip makes, between "ranks" that are host threads of ONE
// process (each with its own blomgpu context).  ncclSend copies its message into a mailbox keyed by (source,
// destination); ncclRecv takes the oldest message of its pair -- RCCL's matching rule for point-to-point calls.
// This lets the CPU suite run the library's real pack / exchange / unpack code with several ranks.
#pragma once
#include <hip/hip_runtime.h>
typedef struct hostemu_comm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclInternalError = 3 };
typedef enum { ncclDouble = 8 } ncclDataType_t;
const char *ncclGetErrorString(ncclResult_t);
ncclResult_t ncclGetUniqueId(ncclUniqueId *);
ncclResult_t ncclCommInitRank(ncclComm_t *, int nranks, ncclUniqueId id, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t, int peer, ncclComm_t, hipStream_t);
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t, int peer, ncclComm_t, hipStream_t);
