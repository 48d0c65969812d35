/*
 * amt_rccl_collectives.hpp -- AmtGpuCollectives (amt_gpu.h) over an RCCL communicator, for a C++ host that runs one process
 * (or thread) per GPU: the sharded drivers amtgpu_scanlogo_sharded / amtgpu_logoframe_allgather_results call back into these
 * two functions, which stage the (small) host buffers through device memory and run ncclAllGather / ncclAllReduce over xGMI.
 *
 *     ncclComm_t comm = ...;                        // ncclCommInitRank, one rank per GPU, rank order == stream order of the shards
 *     amtgpu::RcclCollectives coll(comm, rank, world, device);
 *     amtgpu_scanlogo_sharded(ctx, coll.get(), dY, dU, dV, ..., nframes_local, ...);
 *
 * Volumes are tiny (per-rank valid counts; 3 int64 per rectangle sample = 1.2 MB for a 256x128 logo, three times per ScanLogo;
 * 8 bytes per frame per logo for the all-frames scan), so the exchange is latency-bound and a blocking host-staged call is
 * the right tool; int64 sums make the reduced accumulators -- and hence the .lgd -- identical at any world size.
 * Header-only; needs <hip/hip_runtime_api.h> and <rccl/rccl.h> from ROCm.
 */
#ifndef AMT_RCCL_COLLECTIVES_HPP
#define AMT_RCCL_COLLECTIVES_HPP

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>

#include "amt_gpu.h"

namespace amtgpu {

class RcclCollectives {
    ncclComm_t comm_;
    int device_;
    hipStream_t stream_ = nullptr;
    void* dbuf_ = nullptr;
    size_t dbytes_ = 0;
    AmtGpuCollectives c_{};
    std::string err_;

    bool reserve(size_t bytes)
    {
        if (bytes <= dbytes_) return true;
        if (dbuf_) (void)hipFree(dbuf_);
        dbuf_ = nullptr; dbytes_ = 0;
        if (hipMalloc(&dbuf_, bytes) != hipSuccess) return false;
        dbytes_ = bytes;
        return true;
    }
    bool fail(const char* what) { err_ = what; return false; }

    bool allgather(const void* send, void* recv, int64_t bytes)
    {
        if (hipSetDevice(device_) != hipSuccess) return fail("hipSetDevice");
        const size_t n = (size_t)bytes, total = n * (size_t)c_.world;
        if (!reserve(n + total)) return fail("hipMalloc");
        char* dsend = static_cast<char*>(dbuf_);
        char* drecv = dsend + n;
        if (hipMemcpyAsync(dsend, send, n, hipMemcpyHostToDevice, stream_) != hipSuccess) return fail("hipMemcpyAsync");
        if (ncclAllGather(dsend, drecv, n, ncclUint8, comm_, stream_) != ncclSuccess) return fail("ncclAllGather");
        if (hipMemcpyAsync(recv, drecv, total, hipMemcpyDeviceToHost, stream_) != hipSuccess) return fail("hipMemcpyAsync");
        return hipStreamSynchronize(stream_) == hipSuccess || fail("hipStreamSynchronize");
    }
    bool allreduce(int64_t* buf, int64_t count)
    {
        if (hipSetDevice(device_) != hipSuccess) return fail("hipSetDevice");
        const size_t n = (size_t)count * sizeof(int64_t);
        if (!reserve(n)) return fail("hipMalloc");
        if (hipMemcpyAsync(dbuf_, buf, n, hipMemcpyHostToDevice, stream_) != hipSuccess) return fail("hipMemcpyAsync");
        if (ncclAllReduce(dbuf_, dbuf_, (size_t)count, ncclInt64, ncclSum, comm_, stream_) != ncclSuccess) return fail("ncclAllReduce");   /* exact */
        if (hipMemcpyAsync(buf, dbuf_, n, hipMemcpyDeviceToHost, stream_) != hipSuccess) return fail("hipMemcpyAsync");
        return hipStreamSynchronize(stream_) == hipSuccess || fail("hipStreamSynchronize");
    }
    static int s_allgather(void* user, const void* send, void* recv, int64_t bytes)
    {
        return static_cast<RcclCollectives*>(user)->allgather(send, recv, bytes) ? 1 : 0;
    }
    static int s_allreduce(void* user, int64_t* buf, int64_t count) { return static_cast<RcclCollectives*>(user)->allreduce(buf, count) ? 1 : 0; }

public:
    RcclCollectives(ncclComm_t comm, int rank, int world, int device) : comm_(comm), device_(device)
    {
        if (hipSetDevice(device_) != hipSuccess || hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) != hipSuccess)
            throw std::runtime_error("RcclCollectives: no HIP device / stream");
        c_.rank = rank; c_.world = world;
        c_.allgather = &RcclCollectives::s_allgather;
        c_.allreduce_sum_i64 = &RcclCollectives::s_allreduce;
        c_.user = this;
    }
    ~RcclCollectives()
    {
        (void)hipSetDevice(device_);
        if (dbuf_) (void)hipFree(dbuf_);
        if (stream_) (void)hipStreamDestroy(stream_);
    }
    RcclCollectives(const RcclCollectives&) = delete;
    RcclCollectives& operator=(const RcclCollectives&) = delete;
    const AmtGpuCollectives* get() const { return &c_; }
    const std::string& last_error() const { return err_; }
};

} /* namespace amtgpu */
#endif /* AMT_RCCL_COLLECTIVES_HPP */
