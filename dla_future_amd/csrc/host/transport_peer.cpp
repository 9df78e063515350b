// transport_peer.cpp -- host-driven peer-copy transport for the GPUs of one node: a broadcast is one device-to-device
// copy per receiver out of the root's memory (hipIpc mapping), ordered by interprocess events; no kernel of one process
// ever waits for a flag another process sets.
//
// Replaces, like transport_rccl.cpp, the reference's MPI_Ibcast on row / column communicators
// (communication/kernels/internal/broadcast.h:36-119, communication/broadcast_panel.h:125-210).  On a fully connected
// xGMI node an M-way broadcast is M - 1 concurrent point-to-point copies on the DMA engines -- no compute units, no ring
//  -- where ncclBroadcast runs a kernel per rank.
//
// Protocol of one broadcast on a communicator of M members (every member calls bcast() in the same program order, as
// with every transport here; "control" = a small host-buffer broadcast over the grid's host callback, the channel the
// host-staged transport moves the DATA through):
//   root      copies the payload into one of its STAGING buffers (a local copy at HBM speed), records interprocess
//             event READY on its stream, control-broadcasts {hipIpc memory handle of the staging buffer, READY's handle};
//   receiver  opens the handles (the mapping is cached), makes its stream wait for READY, enqueues
//             hipMemcpyAsync(recv, mapped root allocation + offset), records its interprocess event DONE;
//   then every receiver in turn control-broadcasts DONE's handle and the root makes its stream wait for it, and a
//             last control message from the root releases everybody: when bcast() returns, the root's send buffer is reusable in stream order and the receivers' data is there in
//             stream order -- NCCL's semantics, which the executor's event graph is written against.
// A control message is the happens-before edge that puts each hipEventRecord before the hipStreamWaitEvent on its
// opened twin in the other process.  Why staging buffers: receivers cache the mappings they open, and on this stack the
// hipIpc handle of a fresh allocation can equal the handle of a freed one at the same address (measured: a receiver
// found a 68 KiB mapping behind the handle of a 128 KiB buffer) -- exporting the caller's allocations would hand
// receivers stale mappings.  The staging buffers belong to the transport and are never freed before it goes away.  Reductions and barriers (a handful of scalars, the eigensolver's panel sums) stay on
// the host-staged transport underneath.  Selected with DLAF_MI355X_TRANSPORT=peer on grids that have a host callback.
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "runtime.hpp"

namespace dlaf_mi355x {

namespace {

struct ControlMsg {
  hipIpcMemHandle_t mem;
  hipIpcEventHandle_t ready;
  unsigned long long offset;
  unsigned long long bytes;
};

class PeerTransport final : public Transport {
public:
  PeerTransport(dlaf_host_bcast_fn b, dlaf_host_barrier_fn bar, void* user)
      : inner_(make_host_transport(b, bar, user)), bcast_(b), user_(user) {
    for (auto& st : staging_)
      DLAF_HIP_CHECK(hipEventCreateWithFlags(&st.free_ev, hipEventDisableTiming));
  }
  ~PeerTransport() override {
    (void) hipDeviceSynchronize();
    for (const Twin& t : twins_) {
      (void) hipEventDestroy(t.twin);
      (void) hipEventDestroy(t.marker);
    }
    for (hipEvent_t e : markers_)
      (void) hipEventDestroy(e);
    for (auto& kv : mappings_)
      (void) hipIpcCloseMemHandle(kv.second);
    for (hipEvent_t e : mine_)
      (void) hipEventDestroy(e);
    for (auto& st : staging_) {
      (void) hipEventDestroy(st.free_ev);
      if (st.buf)
        (void) hipFree(st.buf);
    }
    for (void* p : retired_)
      (void) hipFree(p);
  }
  bool device_side() const override { return false; }  // DMA engines: no workgroup slots to keep free

  void bcast(CommAxis axis, int root, int my_index, const void* send, void* recv, size_t bytes,
             hipStream_t stream) override {
    if (bytes == 0)
      return;
    const int members = axis == CommAxis::Row ? npcol : nprow;
    ControlMsg msg;
    std::memset(&msg, 0, sizeof(msg));
    hipIpcEventHandle_t my_done;
    std::memset(&my_done, 0, sizeof(my_done));
    Staging* st = nullptr;
    if (my_index == root) {
      const void* src = send ? send : recv;
      st = &staging_[stage_next_++ % kStaging];
      if (st->bytes < bytes) {
        // (the outgrown buffer stays allocated: its handle must not come back for other memory)
        if (st->buf)
          retired_.push_back(st->buf);
        st->bytes = std::max<size_t>(bytes + bytes / 2, 1 << 20);
        DLAF_HIP_CHECK(hipMalloc(&st->buf, st->bytes));
        DLAF_HIP_CHECK(hipIpcGetMemHandle(&st->handle, st->buf));
      }
      DLAF_HIP_CHECK(hipStreamWaitEvent(stream, st->free_ev, 0));  // its previous readers are done
      DLAF_HIP_CHECK(hipMemcpyAsync(st->buf, src, bytes, hipMemcpyDeviceToDevice, stream));
      msg.mem = st->handle;
      msg.offset = 0;
      msg.bytes = bytes;
      DLAF_HIP_CHECK(hipEventRecord(fresh(&msg.ready), stream));
    }
    control(axis, root, &msg, sizeof(msg));
    if (my_index != root) {
      if (msg.bytes != bytes)
        fatal("[dlaf_mi355x] peer transport: root sends %llu bytes, receiver expects %zu\n", msg.bytes, bytes);
      const char* peer = static_cast<const char*>(mapping(msg.mem));
      wait_opened(msg.ready, stream);
      const hipError_t ce = hipMemcpyAsync(recv, peer + msg.offset, bytes, hipMemcpyDeviceToDevice, stream);
      if (ce != hipSuccess) {
        hipDeviceptr_t b0 = nullptr, b1 = nullptr;
        size_t s0 = 0, s1 = 0;
        const hipError_t e0 = hipMemGetAddressRange(&b0, &s0, const_cast<char*>(peer));
        const hipError_t e1 = hipMemGetAddressRange(&b1, &s1, recv);
        fatal("[dlaf_mi355x] peer transport: copy of %zu bytes from mapping %p + %llu (range %p + %zu: %s) to %p (range %p + %zu: %s) "
              "failed: %s\n", bytes, (const void*) peer, msg.offset, b0, s0, hipGetErrorString(e0), recv, b1, s1,
              hipGetErrorString(e1), hipGetErrorString(ce));
      }
      DLAF_HIP_CHECK(hipEventRecord(fresh(&my_done), stream));
    }
    else if (send != nullptr && recv != nullptr && send != recv) {
      DLAF_HIP_CHECK(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, stream));
    }
    // every receiver tells the root (everybody listens: a control message is a broadcast) that its copy is enqueued
    for (int r = 0; r < members; ++r) {
      if (r == root)
        continue;
      hipIpcEventHandle_t h;
      if (r == my_index)
        h = my_done;
      control(axis, r, &h, sizeof(h));
      if (my_index == root)
        wait_opened(h, stream);
    }
    if (st)
      DLAF_HIP_CHECK(hipEventRecord(st->free_ev, stream));
    // nobody leaves before the root has enqueued its waits (the owners destroy their events a fixed number of
    // broadcasts later)
    char ack = 1;
    control(axis, root, &ack, 1);
  }
  void barrier(hipStream_t stream) override { inner_->barrier(stream); }
  void allreduce_max(double* v, int n, int pr, int pc, int r, int c) override { inner_->allreduce_max(v, n, pr, pc, r, c); }
  void allreduce_sum(void* dev, size_t count, char type, char scope, hipStream_t stream) override {
    share_grid();
    inner_->allreduce_sum(dev, count, type, scope, stream);
  }

private:
  void share_grid() {
    inner_->nprow = nprow;
    inner_->npcol = npcol;
    inner_->myrow = myrow;
    inner_->mycol = mycol;
  }
  void control(CommAxis axis, int root, void* buf, size_t bytes) {
    if (bcast_(user_, (int) axis, root, buf, bytes) != 0)
      fatal("[dlaf_mi355x] peer transport: control broadcast failed\n");
  }
  void* mapping(const hipIpcMemHandle_t& h) {
    const std::string key(reinterpret_cast<const char*>(&h), sizeof(h));
    auto it = mappings_.find(key);
    if (it != mappings_.end())
      return it->second;
    void* p = nullptr;
    DLAF_HIP_CHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    mappings_[key] = p;
    return p;
  }
  // One interprocess event per record: on this stack another process can wait for an interprocess event's FIRST record
  // only -- a wait on the opened twin fails with "invalid argument" once the owner has recorded the event a second time
  // (measured with rings of 64 and of 4 events: the first reuse of a slot).  An interprocess event costs 32 kernel-driver
  // signals (a process owns 4096), so events are retired -- but by GPU progress, not by count: the host thread runs many
  // steps ahead of the device, and the closing control message of a broadcast only says that the peers' waits have been
  // ENQUEUED.  An own event goes when it is at least 16 broadcasts old AND its record has executed (a wait on a signalled
  // event has nothing left to wait for); past kMaxOwn outstanding events the oldest is waited for (bound: kMaxOwn own +
  // kMaxTwins opened events = 112 x 32 = 3584 signals per process over both communicators).
  static constexpr size_t kMinAge = 16, kMaxOwn = 48, kMaxTwins = 64;
  hipEvent_t fresh(hipIpcEventHandle_t* h) {
    hipEvent_t e = nullptr;
    DLAF_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventInterprocess));
    DLAF_HIP_CHECK(hipIpcGetEventHandle(h, e));
    mine_.push_back(e);
    while (mine_.size() > kMinAge) {
      hipError_t q = hipEventQuery(mine_.front());
      if (q == hipErrorNotReady && mine_.size() > kMaxOwn)
        q = hipEventSynchronize(mine_.front());
      if (q != hipSuccess) {
        (void) hipGetLastError();
        break;
      }
      (void) hipEventDestroy(mine_.front());
      mine_.erase(mine_.begin());
    }
    return e;
  }
  // A handle is opened anew for every wait: on this stack a wait on an opened event fails with "invalid argument"
  // once the owner has recorded the event again after the handle was opened (measured: the first reuse of a ring
  // slot).  An opened twin is dropped when a marker recorded on the waiting stream right after the wait has completed,
  // i.e. when the wait itself has executed (see wait_opened).
  struct Twin {
    hipEvent_t twin, marker;
  };
  void wait_opened(const hipIpcEventHandle_t& h, hipStream_t stream) {
    Twin t{nullptr, nullptr};
    DLAF_HIP_CHECK(hipIpcOpenEventHandle(&t.twin, h));
    DLAF_HIP_CHECK(hipStreamWaitEvent(stream, t.twin, 0));
    if (!markers_.empty()) {
      t.marker = markers_.back();
      markers_.pop_back();
    }
    else {
      DLAF_HIP_CHECK(hipEventCreateWithFlags(&t.marker, hipEventDisableTiming));
    }
    DLAF_HIP_CHECK(hipEventRecord(t.marker, stream));
    twins_.push_back(t);
    while (!twins_.empty()) {
      hipError_t q = hipEventQuery(twins_.front().marker);
      if (q == hipErrorNotReady && twins_.size() > kMaxTwins)
        q = hipEventSynchronize(twins_.front().marker);
      if (q != hipSuccess) {
        (void) hipGetLastError();
        break;
      }
      (void) hipEventDestroy(twins_.front().twin);
      markers_.push_back(twins_.front().marker);
      twins_.erase(twins_.begin());
    }
  }

  struct Staging {
    void* buf = nullptr;
    size_t bytes = 0;
    hipIpcMemHandle_t handle;
    hipEvent_t free_ev = nullptr;  // recorded when every receiver of the buffer's last broadcast has copied
  };
  static constexpr int kStaging = 4;
  Staging staging_[kStaging];
  std::vector<void*> retired_;
  unsigned long long stage_next_ = 0;
  std::unique_ptr<Transport> inner_;
  dlaf_host_bcast_fn bcast_;
  void* user_;
  std::vector<hipEvent_t> mine_;
  std::map<std::string, void*> mappings_;
  std::vector<Twin> twins_;
  std::vector<hipEvent_t> markers_;
};

}  // namespace

std::unique_ptr<Transport> make_peer_transport(dlaf_host_bcast_fn b, dlaf_host_barrier_fn bar, void* user) {
  return std::unique_ptr<Transport>(new PeerTransport(b, bar, user));
}

}  // namespace dlaf_mi355x
