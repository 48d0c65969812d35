// compile check of include/amt_rccl_collectives.hpp (a 2-GPU box is needed to RUN RCCL; the sharded drivers themselves are
// exercised with torch.distributed collectives in tests/test_gpu_sharded.py)
#include "amt_rccl_collectives.hpp"

int sharded_scanlogo_with_rccl(AmtGpuContext* ctx, ncclComm_t comm, int rank, int world, int device, const void* dY, const void* dU,
                               const void* dV, int64_t strideY, int64_t strideUV, int pitchY, int pitchUV, int imgw, int imgh,
                               int nframes_local, const char* dst)
{
    amtgpu::RcclCollectives coll(comm, rank, world, device);
    return amtgpu_scanlogo_sharded(ctx, coll.get(), dY, dU, dV, strideY, strideUV, pitchY, pitchUV, imgw, imgh, nframes_local, 1041, dst, 1120,
                                   64, 256, 128, 12, 20000, nullptr);
}
