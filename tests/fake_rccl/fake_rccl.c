/* fake_rccl.c -- TEST DOUBLE for librccl (tests only; never shipped, never on the product path).
 *
 * The GPU boxes of this build have ONE GPU and RCCL refuses two ranks on one device, so the multi-rank logic of
 * libcpm_hip.so (cpm_comm_*, cpm_allreduce_grid_sparse: union masks, slot tables, capacity policy, overflow fall-back,
 * root reduce) could only ever run with a communicator of size 1.  This library implements the eight RCCL entry points
 * libcpm_hip binds (csrc/cpm_comm.hip, load_rccl; ncclSend / ncclRecv since round 5) over POSIX shared memory between processes that share one GPU:
 * a collective synchronises the caller's stream, stages the operands through host memory, meets the other ranks at a
 * process-shared barrier, reduces, and copies the result back.  Stream order is kept (the call blocks the host instead of
 * enqueueing, which a caller cannot tell from a very slow RCCL).  Selected with CPM_RCCL_LIBRARY=<path to this .so>.
 *
 * Built by tests/fake_rccl/build.py: hipcc -x c -shared -fPIC fake_rccl.c -lamdhip64 -lpthread -lrt */
#define __HIP_PLATFORM_AMD__ 1
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;

#define FAKE_MAX_RANKS 8
#define FAKE_SLOT_BYTES ((size_t)40 << 20) /* per rank: the biggest operand a test hands over (config 2's 8 MiB grid and then some) */
#define FAKE_CHANNEL_BYTES ((size_t)24 << 20) /* per ordered pair of ranks: one ncclSend in flight (pages are only touched when used) */

/* point-to-point: one single-message channel per ordered pair (src -> dst).  ncclSend copies the operand in and returns (it waits
 * only while the pair's previous message has not been taken); ncclRecv waits for the message and copies it out.  Only the two ranks
 * of a pair ever meet -- unlike the collectives below there is no barrier over the communicator. */
typedef struct {
    volatile unsigned long sent, taken; /* messages put in / taken out */
    volatile size_t bytes;
    char pad[40];
} fake_channel;

typedef struct {
    volatile int ready;
    int nranks;
    pthread_barrier_t barrier;
} fake_header;

typedef struct fake_comm {
    int rank, nranks;
    char name[64];
    fake_header* hdr;
    unsigned char* slots;  /* nranks x FAKE_SLOT_BYTES */
    unsigned char* channels; /* nranks x nranks x (64-byte control block + FAKE_CHANNEL_BYTES), [src][dst] */
    size_t map_bytes;
    unsigned char* host;   /* staging for the result */
} fake_comm;
typedef fake_comm* ncclComm_t;

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake rccl: HIP error";
        case ncclSystemError: return "fake rccl: system error";
        case ncclInvalidArgument: return "fake rccl: invalid argument";
        default: return "fake rccl: internal error";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof id->internal, "/cpm_fake_rccl_%d_%ld_%ld", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (!out || nranks < 1 || nranks > FAKE_MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    /* (tests of the launcher's wall-clock budget: a communicator set-up that never returns, as a wedged xGMI link would make it) */
    if (getenv("FAKE_RCCL_BLOCK_INIT") && nranks > 1) for (;;) sleep(1);
    fake_comm* c = (fake_comm*)calloc(1, sizeof *c);
    if (!c) return ncclSystemError;
    c->rank = rank; c->nranks = nranks;
    memcpy(c->name, id.internal, sizeof c->name - 1);
    c->map_bytes = 4096 + (size_t)nranks * FAKE_SLOT_BYTES + (size_t)nranks * nranks * (64 + FAKE_CHANNEL_BYTES);
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { free(c); return ncclSystemError; }
    } else {
        for (int tries = 0; tries < 20000 && fd < 0; ++tries) {  /* up to ~20 s for rank 0 */
            fd = shm_open(c->name, O_RDWR, 0600);
            if (fd < 0) usleep(1000);
        }
        if (fd < 0) { free(c); return ncclSystemError; }
        struct stat st;
        for (int tries = 0; tries < 20000; ++tries) { if (fstat(fd, &st) == 0 && (size_t)st.st_size >= c->map_bytes) break; usleep(1000); }
    }
    void* m = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { free(c); return ncclSystemError; }
    c->hdr = (fake_header*)m;
    c->slots = (unsigned char*)m + 4096;
    c->channels = c->slots + (size_t)nranks * FAKE_SLOT_BYTES;   /* (a fresh shm object reads as zeros: sent == taken == 0) */
    if (rank == 0) {
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&c->hdr->barrier, &a, (unsigned)nranks);
        c->hdr->nranks = nranks;
        __sync_synchronize();
        c->hdr->ready = 1;
    } else {
        for (int tries = 0; tries < 20000 && !c->hdr->ready; ++tries) usleep(1000);
        if (!c->hdr->ready) { munmap(m, c->map_bytes); free(c); return ncclSystemError; }
    }
    c->host = (unsigned char*)malloc(FAKE_SLOT_BYTES);
    if (!c->host) return ncclSystemError;
    pthread_barrier_wait(&c->hdr->barrier);  /* (ncclCommInitRank is collective) */
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    (void)devlist;
    if (ndev != 1) return ncclInvalidArgument;  /* one process driving several devices: not what this double is for */
    ncclUniqueId id;
    ncclGetUniqueId(&id);
    return ncclCommInitRank(&comms[0], 1, id, 0);
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (c->hdr) munmap((void*)c->hdr, c->map_bytes);
    if (c->rank == 0) shm_unlink(c->name);
    free(c->host);
    free(c);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { return ncclSuccess; }

static size_t elem_size(ncclDataType_t t) { return t == ncclUint8 || t == ncclInt8 ? 1 : (t == ncclFloat32 || t == ncclInt32 || t == ncclUint32 ? 4 : 0); }

static ncclResult_t collective(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, int root, ncclComm_t c, hipStream_t s) {
    const size_t es = elem_size(dt), bytes = count * es;
    if (!c || es == 0 || bytes > FAKE_SLOT_BYTES) return ncclInvalidArgument;
    if (!(dt == ncclFloat32 && op == ncclSum) && !(dt == ncclUint8 && op == ncclMax)) return ncclInvalidArgument;
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;      /* everything enqueued before the collective */
    unsigned char* mine = c->slots + (size_t)c->rank * FAKE_SLOT_BYTES;
    if (hipMemcpy(mine, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    pthread_barrier_wait(&c->hdr->barrier);
    if (root < 0 || root == c->rank) {
        if (dt == ncclFloat32) {   /* rank order 0, 1, ...: every rank adds in the same order -> the same bits everywhere */
            float* o = (float*)c->host;
            memcpy(o, c->slots, bytes);
            for (int r = 1; r < c->nranks; ++r) {
                const float* a = (const float*)(c->slots + (size_t)r * FAKE_SLOT_BYTES);
                for (size_t i = 0; i < count; ++i) o[i] += a[i];
            }
        } else {
            unsigned char* o = c->host;
            memcpy(o, c->slots, bytes);
            for (int r = 1; r < c->nranks; ++r) {
                const unsigned char* a = c->slots + (size_t)r * FAKE_SLOT_BYTES;
                for (size_t i = 0; i < count; ++i) if (a[i] > o[i]) o[i] = a[i];
            }
        }
    }
    pthread_barrier_wait(&c->hdr->barrier);  /* the slots may be overwritten by the next collective only now */
    if (root < 0 || root == c->rank)
        if (hipMemcpy(recv, c->host, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

static unsigned char* channel_of(ncclComm_t c, int src, int dst) { return c->channels + ((size_t)src * c->nranks + dst) * (64 + FAKE_CHANNEL_BYTES); }

static int wait_until(volatile unsigned long* word, unsigned long at_least, int seconds) {
    for (long spins = 0; *word < at_least; ++spins) {
        if (spins > (long)seconds * 20000) return 0;
        usleep(50);
    }
    __sync_synchronize();
    return 1;
}

ncclResult_t ncclSend(const void* send, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) {
    const size_t bytes = count * elem_size(dt);
    if (!c || elem_size(dt) == 0 || peer < 0 || peer >= c->nranks || peer == c->rank || bytes > FAKE_CHANNEL_BYTES) return ncclInvalidArgument;
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    unsigned char* ch = channel_of(c, c->rank, peer);
    fake_channel* k = (fake_channel*)ch;
    if (!wait_until(&k->taken, k->sent, 60)) return ncclSystemError;   /* the pair's previous message is still in the channel */
    if (hipMemcpy(ch + 64, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    k->bytes = bytes;
    __sync_synchronize();
    k->sent = k->sent + 1;
    return ncclSuccess;
}

ncclResult_t ncclRecv(void* recv, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t s) {
    const size_t bytes = count * elem_size(dt);
    if (!c || elem_size(dt) == 0 || peer < 0 || peer >= c->nranks || peer == c->rank || bytes > FAKE_CHANNEL_BYTES) return ncclInvalidArgument;
    if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
    unsigned char* ch = channel_of(c, peer, c->rank);
    fake_channel* k = (fake_channel*)ch;
    if (!wait_until(&k->sent, k->taken + 1, 60)) return ncclSystemError;
    if (k->bytes != bytes) return ncclInvalidArgument;   /* (RCCL would hang or corrupt: sizes of a send and its receive must agree) */
    if (hipMemcpy(recv, ch + 64, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    __sync_synchronize();
    k->taken = k->taken + 1;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
    return collective(send, recv, count, dt, op, -1, c, s);
}
ncclResult_t ncclReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, int root, ncclComm_t c, hipStream_t s) {
    return collective(send, recv, count, dt, op, root, c, s);
}
