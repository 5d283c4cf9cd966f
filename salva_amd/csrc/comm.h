// comm.h — neighbour exchange between the x-slabs of a multi-GPU run (one process / one world per GPU).
//
// The reference has no distributed code at all; this layer exists only because the path shards
// spatially: interactions have range h = one grid cell (contacts.rs:164-165), so a slab needs a one-cell-plane ghost
// layer from each x-neighbour and nothing else.  Traffic per exchange is the particles of one or two cell planes
// (10^4-10^5 particles, 4-64 bytes each), point to point between adjacent ranks over xGMI, plus one tiny all-reduce
// per convergence test.  No large collective is ever needed.
//
// Three transports: RCCL (ncclSend/ncclRecv groups + ncclAllReduce on the world's stream), the default for multi-GPU runs;
// xGMI peer-direct (comm_peer.hip: flagged stores into IPC-mapped windows of the other ranks' memory, for the ranks of one
// node); and an in-process loopback (one thread per world, device-to-device copies through a shared mailbox) that exercises
// the identical World code on a single GPU for the tests.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <memory>

namespace salva {

class Transport {
  public:
    virtual ~Transport() = default;
    virtual int rank() const = 0;
    virtual int size() const = 0;
    // the HIP device this transport's buffers live on (-1: whichever device the calling thread has current — the loopback)
    virtual int device() const { return -1; }
    bool has_lo() const { return rank() > 0; }
    bool has_hi() const { return rank() + 1 < size(); }

    // Send `n_lo` bytes at `send_lo` to rank-1 and `n_hi` bytes at `send_hi` to rank+1 (device pointers); receive
    // exactly `m_lo` bytes from rank-1 into `recv_lo` and `m_hi` from rank+1 into `recv_hi`.  Sizes must match the
    // peers'.  Enqueued on / ordered with `s`; the data is usable by work enqueued on `s` afterwards.
    virtual void sendrecv(const void* send_lo, size_t n_lo, const void* send_hi, size_t n_hi, void* recv_lo, size_t m_lo,
                          void* recv_hi, size_t m_hi, hipStream_t s) = 0;
    // Host-visible exchange of two counters with the neighbours (blocks until complete): what each will send me.
    virtual void exchange_counts(const uint64_t to_lo[2], const uint64_t to_hi[2], uint64_t from_lo[2], uint64_t from_hi[2],
                                 hipStream_t s) = 0;
    // In-place sum over all ranks of `n` floats / `n` uint64 at device pointer `buf`, ordered with `s`.
    virtual void allreduce_sum_f32(float* buf, int n, hipStream_t s) = 0;
    virtual void allreduce_sum_u64(unsigned long long* buf, int n, hipStream_t s) = 0;
    // All-gather of `n_each` uint64 per rank: all[r * n_each + k] = rank r's mine[k], on every rank, ordered with `s`.  (What the
    // collective DynamicContactSampling assembles its table with: through the sum all-reduce of a zeroed table — the fallback below,
    // right for any transport — a 256-value pass of the peer transport carried 256 values in all; gathered, it carries 256 per
    // RANK.  ADVICE r04 / VERDICT r05.)
    virtual void allgather_u64(const unsigned long long* mine, unsigned long long* all, int n_each, hipStream_t s) {
        if (n_each <= 0) return;
        (void)hipMemsetAsync(all, 0, (size_t)size() * n_each * sizeof(unsigned long long), s);
        (void)hipMemcpyAsync(all + (size_t)rank() * n_each, mine, (size_t)n_each * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s);
        const size_t words = (size_t)size() * n_each;
        for (size_t at = 0; at < words;) {  // (an int count per call)
            const size_t len = words - at < ((size_t)1 << 28) ? words - at : ((size_t)1 << 28);
            allreduce_sum_u64(all + at, (int)len, s);
            at += len;
        }
    }
};

// In-process loopback: `size` transports sharing one mailbox; each must be driven from its own host thread.
struct LoopbackShared;
std::shared_ptr<LoopbackShared> loopback_create(int size);
Transport* loopback_transport(const std::shared_ptr<LoopbackShared>& group, int rank);

// RCCL: unique id created by rank 0 (128 bytes) and distributed by the caller (e.g. torch.distributed / MPI / a file).
constexpr size_t RCCL_ID_BYTES = 128;
void rccl_unique_id(unsigned char out[RCCL_ID_BYTES]);
Transport* rccl_transport(int rank, int size, const unsigned char id[RCCL_ID_BYTES], int device);

// Collective self-test of a transport (patterned messages of `rounds` different lengths up to `max_bytes`, both all-reduces, the
// count exchange); throws HipError describing the first mismatch.
void transport_selftest(Transport& t, size_t max_bytes, int rounds, hipStream_t s);

// Collective: average host-clock microseconds of one exchange of `bytes` bytes each way with both neighbours, and of one
// all-reduce of four floats, over `iters` back-to-back calls each (what one solver iteration of a decomposed run adds).
void transport_time(Transport& t, size_t bytes, int iters, float* us_sendrecv, float* us_allreduce, hipStream_t s);

// xGMI peer-direct, the ranks of one node (one process per rank; two ranks may share a GPU, which is how the single-GPU boxes
// test it).  Two phases, because the windows' IPC handles have to travel between the processes: peer_begin allocates this
// rank's window and returns its handle; the caller gathers all ranks' handles (rank order) and hands them to peer_transport,
// which takes ownership of the setup object.  `slot_bytes`: capacity of one receive slot (longer messages go in rounds).
constexpr size_t PEER_HANDLE_BYTES = 64;
struct PeerSetup;
PeerSetup* peer_begin(int rank, int size, int device, size_t slot_bytes, unsigned char handle_out[PEER_HANDLE_BYTES]);
void peer_abort(PeerSetup* setup);
Transport* peer_transport(PeerSetup* setup, const unsigned char* handles);

}  // namespace salva
