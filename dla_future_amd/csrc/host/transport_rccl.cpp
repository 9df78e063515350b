// transport_rccl.cpp -- RCCL (xGMI) transport: one world communicator plus a row and a column
// sub-communicator per process, broadcasts enqueued on the executor's communication stream.
//
// Replaces CommunicatorGrid's three MPI_Comm_split communicators and the MPI_Ibcast kernels
// (src/communication/communicator_grid.cpp:20-68, communication/kernels/internal/broadcast.h:36-119).
// Stream order on s_comm plays the role of the reference's CommunicatorPipeline token
// (communication/communicator_pipeline.h:41-148): every member of a communicator enqueues its
// collectives in the same program order.
#include <rccl/rccl.h>

#include <cstring>

#include "runtime.hpp"

namespace dlaf_mi355x {

#define DLAF_NCCL_CHECK(expr)                                                                          \
  do {                                                                                                 \
    ncclResult_t r_ = (expr);                                                                          \
    if (r_ != ncclSuccess)                                                                             \
      ::dlaf_mi355x::fatal("[dlaf_mi355x] RCCL error %s at %s:%d: %s\n", ncclGetErrorString(r_), __FILE__, \
                           __LINE__, #expr);                                                           \
  } while (0)

namespace {
class RcclTransport final : public Transport {
public:
  RcclTransport(const void* unique_id, int nranks, int rank, int myrow, int mycol) {
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    DLAF_NCCL_CHECK(ncclCommInitRank(&world_, nranks, id, rank));
    // row communicator: same process row, ranked by process column (and vice versa)
    DLAF_NCCL_CHECK(ncclCommSplit(world_, myrow, mycol, &row_, nullptr));
    DLAF_NCCL_CHECK(ncclCommSplit(world_, mycol, myrow, &col_, nullptr));
    DLAF_HIP_CHECK(hipMalloc(&token_, sizeof(int)));
    DLAF_HIP_CHECK(zero_device_now(token_, sizeof(int)));  // (hipMemset returns before its fill has run: device_api.hpp)
    DLAF_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&scal_dev_), sizeof(double) * kScalars));
    DLAF_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&scal_host_), sizeof(double) * kScalars, hipHostMallocDefault));
    DLAF_HIP_CHECK(hipStreamCreateWithFlags(&scal_stream_, hipStreamNonBlocking));
  }
  ~RcclTransport() override {
    (void) hipFree(token_);
    (void) hipFree(scal_dev_);
    (void) hipHostFree(scal_host_);
    (void) hipStreamDestroy(scal_stream_);
    if (row_)
      (void) ncclCommDestroy(row_);
    if (col_)
      (void) ncclCommDestroy(col_);
    if (world_)
      (void) ncclCommDestroy(world_);
  }
  bool device_side() const override { return true; }
  void bcast(CommAxis axis, int root, int /*my_index*/, const void* send, void* recv, size_t bytes,
             hipStream_t stream) override {
    if (bytes == 0)
      return;
    ncclComm_t c = (axis == CommAxis::Row) ? row_ : col_;
    DLAF_NCCL_CHECK(ncclBroadcast(send ? send : recv, recv, bytes, ncclChar, root, c, stream));
  }
  void group_begin() override { DLAF_NCCL_CHECK(ncclGroupStart()); }
  void group_end() override { DLAF_NCCL_CHECK(ncclGroupEnd()); }
  void barrier(hipStream_t stream) override {
    DLAF_NCCL_CHECK(ncclAllReduce(token_, token_, 1, ncclInt, ncclSum, world_, stream));
    DLAF_HIP_CHECK(hipStreamSynchronize(stream));
  }
  void allreduce_max(double* host_vals, int n, int, int, int, int) override {
    // a handful of scalars (norms, info): a persistent device word buffer + pinned mirror on a stream of
    // their own -- no allocation, no null-stream synchronisation (every factorization ends with one)
    if (n > kScalars)
      fatal("[dlaf_mi355x] allreduce_max of %d values (max %d)\n", n, kScalars);
    std::memcpy(scal_host_, host_vals, sizeof(double) * (size_t) n);
    DLAF_HIP_CHECK(hipMemcpyAsync(scal_dev_, scal_host_, sizeof(double) * (size_t) n, hipMemcpyHostToDevice, scal_stream_));
    DLAF_NCCL_CHECK(ncclAllReduce(scal_dev_, scal_dev_, (size_t) n, ncclDouble, ncclMax, world_, scal_stream_));
    DLAF_HIP_CHECK(hipMemcpyAsync(scal_host_, scal_dev_, sizeof(double) * (size_t) n, hipMemcpyDeviceToHost, scal_stream_));
    DLAF_HIP_CHECK(hipStreamSynchronize(scal_stream_));
    std::memcpy(host_vals, scal_host_, sizeof(double) * (size_t) n);
  }

  void allreduce_sum(void* dev, size_t count, char type, char scope, hipStream_t stream) override {
    if (count == 0)
      return;
    const bool dbl = (type == 'd' || type == 'z');
    const size_t nreal = count * ((type == 'c' || type == 'z') ? 2 : 1);
    ncclComm_t c = scope == 'R' ? row_ : scope == 'C' ? col_ : world_;
    DLAF_NCCL_CHECK(ncclAllReduce(dev, dev, nreal, dbl ? ncclDouble : ncclFloat, ncclSum, c, stream));
  }

private:
  static constexpr int kScalars = 16;
  ncclComm_t world_ = nullptr, row_ = nullptr, col_ = nullptr;
  void* token_ = nullptr;
  double* scal_dev_ = nullptr;
  double* scal_host_ = nullptr;
  hipStream_t scal_stream_ = nullptr;
};
}  // namespace

std::unique_ptr<Transport> make_rccl_transport(const void* unique_id, int nranks, int rank, int /*nprow*/,
                                               int /*npcol*/, int myrow, int mycol) {
  return std::unique_ptr<Transport>(new RcclTransport(unique_id, nranks, rank, myrow, mycol));
}

void rccl_get_unique_id(void* out128) {
  ncclUniqueId id;
  DLAF_NCCL_CHECK(ncclGetUniqueId(&id));
  std::memcpy(out128, &id, sizeof(id));
}

}  // namespace dlaf_mi355x
